// Backward of one generator grow block (add_generator_block, trainer/models.py:358-399; the
// tf.GradientTape sweep of models.py:144-146) in three steps:
//   1. gblock_dgrad_kernel: the data-gradient chain dY -> dX of all 26 linear layers in registers
//      (same transposed, register-chained MFMA scheme as the forward kernel, with the weight
//      fragments packed "input row x output channel"); every intermediate gradient a weight
//      gradient needs is written to the tile-major gradient buffer (row map: gblock_rows.h).
//   2. gblock_wgrad_kernel<TA,TB>: dW[ci][co] = sum over positions of act[ci][p] * grad[co][p]
//      for a table of (activation rows, gradient rows) jobs.  Both buffers hold 32 positions per
//      row, so a lane reads 16 bytes = 4 contraction steps of its channel: operands go straight
//      from L2/HBM to the MFMA, no LDS.  A wave keeps TA x TB accumulator tiles for its share of
//      the position tiles and writes one slab; bias gradients are per-lane sums of the gradient
//      operand.
//   3. gblock_unpack_kernel: one thread per weight / bias element adds the slabs in split order
//      (bitwise reproducible) into the Keras-layout gradient tensors.
#include "bf16.h"
#include "gblock_frag.h"
#include "gblock_rows.h"
#include "gblock_jobs.h"
#include "../../include/mgx.h"

#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------
// transposed weight fragments (units of 64 floats)
// ------------------------------------------------------------------------------------------
constexpr int PT_W43 = 0;                 // 16 slots: [0] W4, [1..4] W3 rows 0-31, [8..11] W3 rows 32-63, rest 0
constexpr int PT_W2 = 16;                 // [c 8][it 2][s 16]
constexpr int PT_W1 = PT_W2 + 256;        // [c 8][ot 4][s 16]
constexpr int PT_BR = PT_W1 + 512;        // 6 branches x PBR
constexpr int PBR = 272;
constexpr int PB_D2 = 0;                  // [ot 2][it 4][s 16]
constexpr int PB_K2 = 128;                // [b 2][it 2][s 16]
constexpr int PB_D1 = 192;                // [s 16], rows >= 16 zero
constexpr int PB_K1 = 208;                // [a 2][ot 4][s 8]
constexpr int PT_K0 = PT_BR + 6 * PBR;    // [it 4][s 16], rows >= 2 zero, the four taps summed
constexpr int PT_FRAGS = PT_K0 + 64;
constexpr int PACKT_FLOATS = PT_FRAGS * FR;

// A-operand fragment of step s for lane (i, h): W[input row i][output channel chan(s, h)]
__global__ __launch_bounds__(256) void gblock_packT_kernel(mgx_gblock_weights w, float* __restrict__ packT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACKT_FLOATS) return;
    const int fr = idx / FR, lane = idx % FR;
    const int i = lane & 31, h = lane >> 5;
    float v = 0.f;
    if (fr < PT_W2) {
        if (fr == 0) v = i < 8 ? w.w4[i * 2 + h] : 0.f;                       // W4 (8,2)
        else if (fr <= 4 || (fr >= 8 && fr < 12)) {                           // W3 (64,8)
            const int ot = fr >= 8, s = ot ? fr - 8 : fr - 1;
            v = w.w3[(32 * ot + i) * 8 + s + 4 * h];
        }
    } else if (fr < PT_W1) {                                                  // W2 (256,64)
        const int g = fr - PT_W2;
        const int c = g / 32, it = (g % 32) / 16, s = g % 16;
        v = w.w2[(32 * c + i) * 64 + 32 * it + chan(s, h)];
    } else if (fr < PT_BR) {                                                  // W1 (128,256)
        const int g = fr - PT_W1;
        const int c = g / 64, ot = (g % 64) / 16, s = g % 16;
        v = w.w1[(32 * ot + i) * 256 + 32 * c + chan(s, h)];
    } else if (fr < PT_K0) {
        const int k = (fr - PT_BR) / PBR, f = (fr - PT_BR) % PBR;
        if (f < PB_K2) {                                                      // D2 (64,128)
            const int ot = f / 64, it = (f % 64) / 16, s = f % 16;
            v = w.d2[k][(32 * ot + i) * 128 + 32 * it + chan(s, h)];
        } else if (f < PB_D1) {                                               // K2 (1,2,64,32): [b][co][ci]
            const int g = f - PB_K2;
            const int b = g / 32, it = (g % 32) / 16, s = g % 16;
            v = w.k2[k][(b * 64 + 32 * it + chan(s, h)) * 32 + i];
        } else if (f < PB_K1) {                                               // D1 (16,32)
            const int s = f - PB_D1;
            v = i < 16 ? w.d1[k][i * 32 + chan(s, h)] : 0.f;
        } else {                                                              // K1 (2,1,16,128): [a][co][ci]
            const int g = f - PB_K1;
            const int a = g / 32, ot = (g % 32) / 8, s = g % 8;
            v = w.k1[k][(a * 16 + chan(s, h)) * 128 + 32 * ot + i];
        }
    } else {                                                                  // K0 (2,2,2,128): [tap][c][n]
        const int g = fr - PT_K0;
        const int it = g / 16, s = g % 16;
        if (i < 2)
            for (int tap = 0; tap < 4; ++tap) v += w.k0[(tap * 2 + i) * 128 + 32 * it + chan(s, h)];
    }
    packT[frag_slot(fr, lane)] = v;
}

__device__ __forceinline__ void zero(f32x16& t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.f;
}

// ------------------------------------------------------------------------------------------
// data gradient.  A wave owns 32 low-resolution positions; sub-row a is the (rolled) outer loop.
// Per a: the tail chain of both sub-columns (50 fragment groups each), then per branch 23 groups
// (D2 and K2 transposed for both sub-columns, D1, K1).  Buffer parities are compile-time: the
// branch body is instantiated for both entry parities and the branch loop runs two per trip.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void gblock_dgrad_kernel(const float* __restrict__ packT,
                                                              const float* __restrict__ dy,
                                                              float* __restrict__ delta, float* __restrict__ dx,
                                                              long pos0, long npix, int hl, int wl) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long tile = (long)blockIdx.x * 4 + wv;     // tile of the side buffer, counted from pos0
    const long p = pos0 + tile * 32 + r;
    const bool pvalid = p < npix;
    const long pc = pvalid ? p : 0;
    const long row = pc / wl;                 // n*hl + i
    const int j = (int)(pc - row * wl);
    const float* dy0 = dy + ((row * 2) * (2L * wl) + 2L * j) * 2 + h;     // output row 2i, column 2j, channel h
    float* dt = delta + tile * (GB_ROWS * 32L);
    WBuf B;
    B.touch[0] = B.touch[1] = 0.f;
    f32x16 dF[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) zero(dF[it]);
    pf<0, 16>(B, fptr(packT, PT_W43), lane);

    for (int a = 0; a < 2; ++a) {
        f32x16 dS[2][4];
        // ---- tail: dY -> dT3 -> dT2 -> {dT1_c -> dS} for both sub-columns ----
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ab = a * 2 + b;
            const float o = pvalid ? dy0[(a * (2L * wl) + b) * 2] : 0.f;
            f32x16 dT3, dT2[2];
            zero(dT3); zero(dT2[0]); zero(dT2[1]);
            pf<1, 4>(B, fptr(packT, PT_W43 + 8), lane);
            dT3 = mfma32(B.v[0][0], o, dT3);
#pragma unroll
            for (int s = 0; s < 4; ++s) dT2[0] = mfma32(B.v[0][1 + s], dT3[s], dT2[0]);
            MGX_FENCE();
            pf<0, 16>(B, fptr(packT, PT_W2), lane);
#pragma unroll
            for (int s = 0; s < 4; ++s) dT2[1] = mfma32(B.v[1][s], dT3[s], dT2[1]);
            dt[(GB_X + ab * 2 + h) * 32 + r] = o;
            save_rows<4>(dt, GB_T3 + ab * 8, dT3, r, h);
            MGX_FENCE();
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) zero(dS[b][ot]);
            for (int c = 0; c < 8; ++c) {
                f32x16 dT1;
                zero(dT1);
                const int w2 = PT_W2 + c * 32, w1 = PT_W1 + c * 64;
                pf<1, 16>(B, fptr(packT, w2 + 16), lane); mma<0, 0, 16>(dT1, dT2[0], B); MGX_FENCE();
                pf<0, 16>(B, fptr(packT, w1), lane);      mma<1, 0, 16>(dT1, dT2[1], B); MGX_FENCE();
                pf<1, 16>(B, fptr(packT, w1 + 16), lane); mma<0, 0, 16>(dS[b][0], dT1, B); MGX_FENCE();
                pf<0, 16>(B, fptr(packT, w1 + 32), lane); mma<1, 0, 16>(dS[b][1], dT1, B); MGX_FENCE();
                pf<1, 16>(B, fptr(packT, w1 + 48), lane); mma<0, 0, 16>(dS[b][2], dT1, B); MGX_FENCE();
                const float* nxt = c < 7 ? fptr(packT, w2 + 32) : (b == 0 ? fptr(packT, PT_W43) : fptr(packT, PT_BR));
                pf<0, 16>(B, nxt, lane);
                mma<1, 0, 16>(dS[b][3], dT1, B);
                save_rows<16>(dt, GB_T1 + ab * 256 + c * 32, dT1, r, h);
                MGX_FENCE();
            }
            save_rows<16>(dt, GB_T2 + ab * 64, dT2[0], r, h);
            save_rows<16>(dt, GB_T2 + ab * 64 + 32, dT2[1], r, h);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) save_rows<16>(dt, GB_S + ab * 128 + ot * 32, dS[b][ot], r, h);
        }
        // ---- branches ----
        auto branch = [&](auto qc, int k) {
            constexpr int Q = decltype(qc)::value, R = Q ^ 1;
            const int bf = PT_BR + k * PBR;
            float* dbr = dt + (GB_BR + k * GB_BR_ROWS) * 32;
            f32x16 do2;
            zero(do2);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x16 do3[2];
                zero(do3[0]); zero(do3[1]);
                // do3 = D2 dS: 8 groups (ot, it); entry: buffer Q holds group 0
#pragma unroll
                for (int g = 0; g < 8; g += 2) {
                    pf<R, 16>(B, fptr(packT, bf + PB_D2 + (g + 1) * 16), lane);
                    mma<Q, 0, 16>(do3[g / 4], dS[b][g % 4], B); MGX_FENCE();
                    pf<Q, 16>(B, fptr(packT, g < 6 ? bf + PB_D2 + (g + 2) * 16 : bf + PB_K2 + b * 32), lane);
                    mma<R, 0, 16>(do3[g / 4], dS[b][g % 4 + 1], B); MGX_FENCE();
                }
                // do2 += K2[b]^T do3
                pf<R, 16>(B, fptr(packT, bf + PB_K2 + b * 32 + 16), lane); mma<Q, 0, 16>(do2, do3[0], B); MGX_FENCE();
                pf<Q, 16>(B, fptr(packT, b == 0 ? bf + PB_D2 : bf + PB_D1), lane);
                mma<R, 0, 16>(do2, do3[1], B);
                save_rows<16>(dbr, GB_O3 + (a * 2 + b) * 64, do3[0], r, h);
                save_rows<16>(dbr, GB_O3 + (a * 2 + b) * 64 + 32, do3[1], r, h);
                MGX_FENCE();
            }
            // do1 = D1 do2 (rows < 16), dF += K1[a]^T do1 (8 steps per output tile)
            f32x16 do1;
            zero(do1);
            pf<R, 16>(B, fptr(packT, bf + PB_K1 + a * 32), lane);
            mma<Q, 0, 16>(do1, do2, B);
            save_rows<16>(dbr, GB_O2 + a * 32, do2, r, h);
            MGX_FENCE();
            pf<Q, 16>(B, fptr(packT, bf + PB_K1 + a * 32 + 16), lane);
            mma<R, 0, 8>(dF[0], do1, B);
#pragma unroll
            for (int s = 0; s < 8; ++s) dF[1] = mfma32(B.v[R][8 + s], do1[s], dF[1]);
            save_rows<8>(dbr, GB_O1 + a * 16, do1, r, h);
            MGX_FENCE();
            const float* nxt = k < 5 ? fptr(packT, bf + PBR) : (a == 0 ? fptr(packT, PT_W43) : fptr(packT, PT_K0));
            pf<R, 16>(B, nxt, lane);
            mma<Q, 0, 8>(dF[2], do1, B);
#pragma unroll
            for (int s = 0; s < 8; ++s) dF[3] = mfma32(B.v[Q][8 + s], do1[s], dF[3]);
            MGX_FENCE();
        };
        for (int k = 0; k < 6; k += 2) {
            branch(std::integral_constant<int, 0>{}, k);
            branch(std::integral_constant<int, 1>{}, k + 1);
        }
    }
    // ---- dX = (sum of the four taps of K0)^T dF; the up-sampling backward is that tap sum ----
#pragma unroll
    for (int it = 0; it < 4; ++it) save_rows<16>(dt, GB_F + it * 32, dF[it], r, h);
    f32x16 dxa;
    zero(dxa);
    pf<1, 16>(B, fptr(packT, PT_K0 + 16), lane); mma<0, 0, 16>(dxa, dF[0], B); MGX_FENCE();
    pf<0, 16>(B, fptr(packT, PT_K0 + 32), lane); mma<1, 0, 16>(dxa, dF[1], B); MGX_FENCE();
    pf<1, 16>(B, fptr(packT, PT_K0 + 48), lane); mma<0, 0, 16>(dxa, dF[2], B); MGX_FENCE();
    mma<1, 0, 16>(dxa, dF[3], B);
    if (pvalid && h == 0) *(float2*)(dx + pc * 2) = make_float2(dxa[0], dxa[1]);
}

// ------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------
template <int TA, int TB>
struct Frags {
    f32x4 a[TA][4], b[TB][4];
};

template <int TA, int TB>
__device__ __forceinline__ void load_frags(Frags<TA, TB>& f, const float* __restrict__ act,
                                           const float* __restrict__ delta, const GJob& jb, long pt, int sub, int i,
                                           int h) {
    const float* pa = act + (pt * GB_ROWS + jb.rowA + sub * jb.strideA + i) * 32L + 16 * h;      // float32: always the full map
    const float* pb = delta + (pt * GB_ROWS + jb.rowB + sub * jb.strideB + i) * 32L + 16 * h;
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) f.a[t][q] = *(const f32x4*)(pa + t * 1024 + q * 4);
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) f.b[u][q] = *(const f32x4*)(pb + u * 1024 + q * 4);
}

template <int TA, int TB>
__global__ __launch_bounds__(256, 1) void gblock_wgrad_kernel(const float* __restrict__ act,
                                                              const float* __restrict__ delta, GJobs jobs, int job0,
                                                              float* __restrict__ ws, long ntiles) {
    // Jobs of one launch share operand rows (the same activation tile meets several gradient tiles
    // and vice versa) and walk the position tiles at the same pace.  Workgroups are handed to the 8
    // XCDs round-robin by linear id, so when the grid allows it the ids are re-mapped to put all
    // jobs of one slab index on one XCD: the shared rows are then read from HBM once per XCD-local
    // L2 instead of once per job.  Slabs are addressed by (job, split): the sums do not change.
    int xb = blockIdx.x, jobi = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        const unsigned q = lin >> 3;
        jobi = (int)(q % gridDim.y);
        xb = (int)((q / gridDim.y) * 8 + (lin & 7));
    }
    const GJob jb = jobs.j[job0 + jobi];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long split = (long)xb * 4 + wv, nw = (long)gridDim.x * 4;
    f32x16 acc[TA][TB];
    float bs[TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u) zero(acc[t][u]);
#pragma unroll
    for (int u = 0; u < TB; ++u) bs[u] = 0.f;

    long pt = split;
    int sub = 0;
    Frags<TA, TB> cur, nxt;
    if (pt < ntiles) load_frags<TA, TB>(cur, act, delta, jb, pt, 0, i, h);
    while (pt < ntiles) {
        long ptn = pt;
        int subn = sub + 1;
        if (subn == jb.nsub) { subn = 0; ptn = pt + nw; }
        const bool more = ptn < ntiles;
        if (more) load_frags<TA, TB>(nxt, act, delta, jb, ptn, subn, i, h);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int u = 0; u < TB; ++u) acc[t][u] = mfma32(cur.a[t][q][e], cur.b[u][q][e], acc[t][u]);
        if (jb.bias) {
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) bs[u] += (cur.b[u][q][0] + cur.b[u][q][1]) + (cur.b[u][q][2] + cur.b[u][q][3]);
        }
        MGX_FENCE();
        if (more) cur = nxt;
        pt = ptn;
        sub = subn;
    }
    float* slab = ws + jb.ws_off + split * (TA * TB * 1024L + TB * 64L);
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) slab[((t * TB + u) * 16 + reg) * 64 + lane] = acc[t][u][reg];
#pragma unroll
    for (int u = 0; u < TB; ++u) slab[TA * TB * 1024 + u * 64 + lane] = bs[u];
}

// The same weight-gradient jobs from bf16 side buffers on v_mfma_f32_32x32x16_bf16: a row of a side buffer
// holds 32 positions = 64 bytes, so lane (channel i, half h) reads 16 bytes = the 8 contraction steps
// 8h .. 8h+7 of k-step s at byte 32 s + 16 h of its row: operands still go straight from L2 / HBM to the
// MFMA.  Accumulator tiles, slabs and the bias sums have the layout of the float32 kernel, so
// gblock_unpack_kernel serves both.
template <int TA, int TB>
struct FragsB {
    u32x4 a[TA][2], b[TB][2];
};

template <int TA, int TB>
__device__ __forceinline__ void load_frags_b(FragsB<TA, TB>& f, const bf16_t* __restrict__ act,
                                             const bf16_t* __restrict__ delta, const GJob& jb, long rows, long pt, int sub,
                                             int i, int h) {
    const bf16_t* pa = act + (pt * rows + jb.rowA + sub * jb.strideA + i) * 32L + 8 * h;
    const bf16_t* pb = delta + (pt * rows + jb.rowB + sub * jb.strideB + i) * 32L + 8 * h;
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.a[t][ks] = *(const u32x4*)(pa + t * 1024 + ks * 16);
#pragma unroll
    for (int u = 0; u < TB; ++u)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) f.b[u][ks] = *(const u32x4*)(pb + u * 1024 + ks * 16);
}

template <int TA, int TB>
__global__ __launch_bounds__(256, 1) void gblock_wgrad_bf16_kernel(const bf16_t* __restrict__ act,
                                                                   const bf16_t* __restrict__ delta, GJobs jobs, int job0,
                                                                   float* __restrict__ ws, long ntiles) {
    int xb = blockIdx.x, jobi = blockIdx.y;
    if ((gridDim.x & 7) == 0) {           // all jobs of one slab index on one XCD (see gblock_wgrad_kernel)
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        const unsigned q = lin >> 3;
        jobi = (int)(q % gridDim.y);
        xb = (int)((q / gridDim.y) * 8 + (lin & 7));
    }
    const GJob jb = jobs.j[job0 + jobi];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const long split = (long)xb * 4 + wv, nw = (long)gridDim.x * 4;
    f32x16 acc[TA][TB];
    float bs[TB];
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u) zero(acc[t][u]);
#pragma unroll
    for (int u = 0; u < TB; ++u) bs[u] = 0.f;

    long pt = split;
    int sub = 0;
    FragsB<TA, TB> cur, nxt;
    const long rows = jobs.rows;
    if (pt < ntiles) load_frags_b<TA, TB>(cur, act, delta, jb, rows, pt, 0, i, h);
    while (pt < ntiles) {
        long ptn = pt;
        int subn = sub + 1;
        if (subn == jb.nsub) { subn = 0; ptn = pt + nw; }
        const bool more = ptn < ntiles;
        if (more) load_frags_b<TA, TB>(nxt, act, delta, jb, rows, ptn, subn, i, h);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < TA; ++t)
#pragma unroll
                for (int u = 0; u < TB; ++u) acc[t][u] = mfma_bf16(cur.a[t][ks], cur.b[u][ks], acc[t][u]);
        if (jb.bias) {
#pragma unroll
            for (int u = 0; u < TB; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bs[u] += bf_lo(cur.b[u][ks][q]) + bf_hi(cur.b[u][ks][q]);
        }
        MGX_FENCE();
        if (more) cur = nxt;
        pt = ptn;
        sub = subn;
    }
    float* slab = ws + jb.ws_off + split * (TA * TB * 1024L + TB * 64L);
#pragma unroll
    for (int t = 0; t < TA; ++t)
#pragma unroll
        for (int u = 0; u < TB; ++u)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) slab[((t * TB + u) * 16 + reg) * 64 + lane] = acc[t][u][reg];
#pragma unroll
    for (int u = 0; u < TB; ++u) slab[TA * TB * 1024 + u * 64 + lane] = bs[u];
}

// sum over the job's slabs of accumulator element (tile, ci, co)
// Partial sums are taken by 8 cooperating threads per element (thread `sub` adds slabs sub, sub+8,
// ...); gblock_unpack_kernel then adds the 8 partials in a fixed order.
__device__ __forceinline__ float slab_sum(const float* __restrict__ ws, const GJob& jb, int j, int tile, int ci, int co,
                                          int sub) {
    const JobShape sh = job_shape(j);
    const long stride = sh.ta * sh.tb * 1024L + sh.tb * 64L;
    const int lane = co + 32 * ((ci >> 2) & 1), reg = (ci & 3) + 4 * (ci >> 3);
    const float* p = ws + jb.ws_off + (tile * 16 + reg) * 64 + lane;
    float s = 0.f;
    for (int sp = sub; sp < jb.nsplit; sp += 8) s += p[sp * stride];
    return s;
}

__device__ __forceinline__ float bias_sum(const float* __restrict__ ws, const GJob& jb, int j, int u, int col, int sub) {
    const JobShape sh = job_shape(j);
    const long stride = sh.ta * sh.tb * 1024L + sh.tb * 64L;
    const float* p = ws + jb.ws_off + sh.ta * sh.tb * 1024L + u * 64 + col;
    float s = 0.f;
    for (int sp = sub; sp < jb.nsplit; sp += 8) s += p[sp * stride] + p[sp * stride + 32];
    return s;
}

constexpr int BR_PARAMS = 4096 + 16 + 512 + 32 + 4096 + 64 + 8192 + 128;   // 17136
constexpr int N_PARAMS = 1024 + 128 + 6 * BR_PARAMS + 32768 + 256 + 16384 + 64 + 512 + 8 + 16 + 2;   // 153978

__global__ __launch_bounds__(256) void gblock_unpack_kernel(mgx_gblock_grads g, GJobs jobs,
                                                            const float* __restrict__ ws) {
    const int gt = blockIdx.x * 256 + threadIdx.x;
    int e = gt >> 3;
    const int sub = gt & 7;
    const bool live = e < N_PARAMS;
    if (!live) e = N_PARAMS - 1;          // keep the 8-lane groups whole for the shuffles below
    float* dst;
    float v;
    if (e < 1024) {                                   // K0 [tap][c][n]
        const int c = (e >> 7) & 1, n = e & 127;
        dst = g.k0 + e;
        v = slab_sum(ws, jobs.j[38], 38, n >> 5, c, n & 31, sub);
    } else if ((e -= 1024) < 128) {                   // b0
        dst = g.b0 + e;
        v = bias_sum(ws, jobs.j[38], 38, e >> 5, e & 31, sub);
    } else if ((e -= 128) < 6 * BR_PARAMS) {
        const int k = e / BR_PARAMS;
        e -= k * BR_PARAMS;
        if (e < 4096) {                               // K1 [a][co][ci]
            const int aco = e >> 7, ci = e & 127;
            dst = g.k1[k] + e;
            v = slab_sum(ws, jobs.j[32 + k], 32 + k, ci >> 5, ci & 31, aco, sub);
        } else if ((e -= 4096) < 16) {                // b1
            dst = g.b1[k] + e;
            v = bias_sum(ws, jobs.j[32 + k], 32 + k, 0, e, sub) + bias_sum(ws, jobs.j[32 + k], 32 + k, 0, 16 + e, sub);
        } else if ((e -= 16) < 512) {                 // D1 [in 16][out 32]
            dst = g.d1[k] + e;
            v = slab_sum(ws, jobs.j[14 + k], 14 + k, 0, e >> 5, e & 31, sub);
        } else if ((e -= 512) < 32) {                 // c1
            dst = g.c1[k] + e;
            v = bias_sum(ws, jobs.j[14 + k], 14 + k, 0, e, sub);
        } else if ((e -= 32) < 4096) {                // K2 [b][co][ci]
            const int b = e >> 11, co = (e >> 5) & 63, ci = e & 31;
            const int j = 20 + k * 2 + b;
            dst = g.k2[k] + e;
            v = slab_sum(ws, jobs.j[j], j, co >> 5, ci, co & 31, sub);
        } else if ((e -= 4096) < 64) {                // b2
            const int j = 20 + k * 2;
            dst = g.b2[k] + e;
            v = bias_sum(ws, jobs.j[j], j, e >> 5, e & 31, sub) + bias_sum(ws, jobs.j[j + 1], j + 1, e >> 5, e & 31, sub);
        } else if ((e -= 64) < 8192) {                // D2 [in 64][out 128]
            const int in = e >> 7, out = e & 127;
            dst = g.d2[k] + e;
            v = slab_sum(ws, jobs.j[4 + k], 4 + k, (in >> 5) * 4 + (out >> 5), in & 31, out & 31, sub);
        } else {                                      // c2: the Add hands every branch the same gradient
            e -= 8192;
            dst = g.c2[k] + e;
            v = bias_sum(ws, jobs.j[4], 4, e >> 5, e & 31, sub);
        }
    } else if ((e -= 6 * BR_PARAMS) < 32768) {        // W1 [in 128][out 256]
        const int in = e >> 8, out = e & 255;
        const int j = (in >> 6) * 2 + (out >> 7);
        dst = g.w1 + e;
        v = slab_sum(ws, jobs.j[j], j, ((in & 63) >> 5) * 4 + ((out & 127) >> 5), in & 31, out & 31, sub);
    } else if ((e -= 32768) < 256) {                  // e1
        const int j = e >> 7;
        dst = g.e1 + e;
        v = bias_sum(ws, jobs.j[j], j, (e & 127) >> 5, e & 31, sub);
    } else if ((e -= 256) < 16384) {                  // W2 [in 256][out 64]
        const int in = e >> 6, out = e & 63;
        const int j = 10 + (in >> 7);
        dst = g.w2 + e;
        v = slab_sum(ws, jobs.j[j], j, ((in & 127) >> 5) * 2 + (out >> 5), in & 31, out & 31, sub);
    } else if ((e -= 16384) < 64) {                   // e2
        dst = g.e2 + e;
        v = bias_sum(ws, jobs.j[10], 10, e >> 5, e & 31, sub);
    } else if ((e -= 64) < 512) {                     // W3 [in 64][out 8]
        const int in = e >> 3, out = e & 7;
        dst = g.w3 + e;
        v = slab_sum(ws, jobs.j[12], 12, in >> 5, in & 31, out, sub);
    } else if ((e -= 512) < 8) {                      // e3
        dst = g.e3 + e;
        v = bias_sum(ws, jobs.j[12], 12, 0, e, sub);
    } else if ((e -= 8) < 16) {                       // W4 [in 8][out 2]
        dst = g.w4 + e;
        v = slab_sum(ws, jobs.j[13], 13, 0, e >> 1, e & 1, sub);
    } else {                                          // e4
        e -= 16;
        dst = g.e4 + e;
        v = bias_sum(ws, jobs.j[13], 13, 0, e, sub);
    }
    float total = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) total += __shfl(v, (threadIdx.x & ~7) + k, 64);
    if (live && sub == 0) *dst += total;
}

template <int TA, int TB>
void launch_wgrad(const ShapeRun& run, const float* act, const float* delta, const GJobs& J, float* ws, long ntiles,
                  hipStream_t st) {
    dim3 grid(grid_x(ntiles, run), run.njobs);
    hipLaunchKernelGGL((gblock_wgrad_kernel<TA, TB>), grid, dim3(256), 0, st, act, delta, J, run.job0, ws, ntiles);
}

template <int TA, int TB>
void launch_wgrad_b(const ShapeRun& run, const bf16_t* act, const bf16_t* delta, const GJobs& J, float* ws, long ntiles,
                    hipStream_t st) {
    dim3 grid(grid_x(ntiles, run), run.njobs);
    hipLaunchKernelGGL((gblock_wgrad_bf16_kernel<TA, TB>), grid, dim3(256), 0, st, act, delta, J, run.job0, ws, ntiles);
}

bool all_set(const void* strukt, size_t bytes) {
    const void* const* ptrs = (const void* const*)strukt;
    for (size_t i = 0; i < bytes / sizeof(void*); ++i)
        if (!ptrs[i]) return false;
    return true;
}

}  // namespace

extern "C" int64_t mgx_gblock_packt_floats(void) { return PACKT_FLOATS + FRAG_SLACK; }

extern "C" int64_t mgx_gblock_wgrad_workspace_n(int64_t npos) {
    if (npos <= 0) return 0;
    GJobs J;
    const long ntiles = ((npos + 127) / 128) * 4;
    const long full = build_jobs(J, ntiles), lean = build_jobs(J, ntiles, true);     // either entry point may follow
    return (full > lean ? full : lean) * (int64_t)sizeof(float);
}

extern "C" int64_t mgx_gblock_wgrad_workspace(int64_t nimg, int hl, int wl) {
    if (nimg <= 0 || hl <= 0 || wl <= 0) return 0;
    return mgx_gblock_wgrad_workspace_n((long)nimg * hl * wl);
}

extern "C" int mgx_gblock_dgrad_f32(const mgx_gblock_weights* wp, float* packt, const float* dy, float* delta,
                                    float* dx, int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos,
                                    void* stream) {
    if (!wp || !packt || !dy || !delta || !dx || nimg <= 0 || hl <= 0 || wl <= 0) return MGX_ERR_ARG;
    const long npix = (long)nimg * hl * wl;
    if (pos0 < 0 || pos0 >= npix || (pos0 & 127)) return MGX_ERR_ARG;
    const long end = (npos <= 0 || pos0 + npos > npix) ? npix : pos0 + npos;
    const mgx_gblock_weights w = *wp;
    if (!all_set(&w, sizeof(w))) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (pos0 == 0)      // the packed fragments of a call sequence over ranges are filled by its first call
        hipLaunchKernelGGL(gblock_packT_kernel, dim3((PACKT_FLOATS + 255) / 256), dim3(256), 0, st, w, packt);
    const long blocks = (end - pos0 + 127) / 128;
    if (blocks > 0x7fffffffL) return MGX_ERR_ARG;
    hipLaunchKernelGGL(gblock_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packt, dy, delta, dx, (long)pos0,
                       end, hl, wl);
    return mgx_launch_status();
}

extern "C" int mgx_gblock_wgrad_f32(const float* act, const float* delta, const mgx_gblock_grads* gp, void* workspace,
                                    int64_t workspace_bytes, int64_t npos, void* stream) {
    if (!act || !delta || !gp || !workspace || npos <= 0) return MGX_ERR_ARG;
    const mgx_gblock_grads g = *gp;
    if (!all_set(&g, sizeof(g))) return MGX_ERR_ARG;
    const long ntiles = ((npos + 127) / 128) * 4;
    GJobs J;
    const long need = build_jobs(J, ntiles);
    if (workspace_bytes < need * (int64_t)sizeof(float)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    launch_wgrad<2, 4>(kRuns[0], act, delta, J, ws, ntiles, st);
    launch_wgrad<4, 2>(kRuns[1], act, delta, J, ws, ntiles, st);
    launch_wgrad<2, 1>(kRuns[2], act, delta, J, ws, ntiles, st);
    launch_wgrad<1, 1>(kRuns[3], act, delta, J, ws, ntiles, st);
    launch_wgrad<1, 2>(kRuns[4], act, delta, J, ws, ntiles, st);
    launch_wgrad<4, 1>(kRuns[5], act, delta, J, ws, ntiles, st);
    launch_wgrad<1, 4>(kRuns[6], act, delta, J, ws, ntiles, st);
    hipLaunchKernelGGL(gblock_unpack_kernel, dim3((N_PARAMS * 8 + 255) / 256), dim3(256), 0, st, g, J, ws);
    return mgx_launch_status();
}

extern "C" int mgx_gblock_wgrad_bf16(const void* act, const void* delta, const mgx_gblock_grads* gp, void* workspace,
                                     int64_t workspace_bytes, int64_t npos, void* stream) {
    if (!act || !delta || !gp || !workspace || npos <= 0) return MGX_ERR_ARG;
    const mgx_gblock_grads g = *gp;
    if (!all_set(&g, sizeof(g))) return MGX_ERR_ARG;
    const long ntiles = ((npos + 127) / 128) * 4;
    GJobs J;
    const long need = build_jobs(J, ntiles);
    if (workspace_bytes < need * (int64_t)sizeof(float)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const bf16_t* a = (const bf16_t*)act;
    const bf16_t* d = (const bf16_t*)delta;
    launch_wgrad_b<2, 4>(kRuns[0], a, d, J, ws, ntiles, st);
    launch_wgrad_b<4, 2>(kRuns[1], a, d, J, ws, ntiles, st);
    launch_wgrad_b<2, 1>(kRuns[2], a, d, J, ws, ntiles, st);
    launch_wgrad_b<1, 1>(kRuns[3], a, d, J, ws, ntiles, st);
    launch_wgrad_b<1, 2>(kRuns[4], a, d, J, ws, ntiles, st);
    launch_wgrad_b<4, 1>(kRuns[5], a, d, J, ws, ntiles, st);
    launch_wgrad_b<1, 4>(kRuns[6], a, d, J, ws, ntiles, st);
    hipLaunchKernelGGL(gblock_unpack_kernel, dim3((N_PARAMS * 8 + 255) / 256), dim3(256), 0, st, g, J, ws);
    return mgx_launch_status();
}

// gblock_lean.hip
int mgx_gblock_lean_launch(const mgx_gblock_weights& w, const void* act, const void* delta, const GJobs& J, float* ws,
                           long ntiles, hipStream_t st);

extern "C" int mgx_gblock_wgrad_bf16_lean(const mgx_gblock_weights* wp, const void* act, const void* delta,
                                          const mgx_gblock_grads* gp, void* workspace, int64_t workspace_bytes,
                                          int64_t npos, void* stream) {
    if (!wp || !act || !delta || !gp || !workspace || npos <= 0) return MGX_ERR_ARG;
    const mgx_gblock_grads g = *gp;
    const mgx_gblock_weights w = *wp;
    if (!all_set(&g, sizeof(g)) || !all_set(&w, sizeof(w))) return MGX_ERR_ARG;
    const long ntiles = ((npos + 127) / 128) * 4;
    GJobs J;
    const long need = build_jobs(J, ntiles, true);
    if (workspace_bytes < need * (int64_t)sizeof(float)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const bf16_t* a = (const bf16_t*)act;
    const bf16_t* d = (const bf16_t*)delta;
    if (mgx_gblock_lean_launch(w, act, delta, J, ws, ntiles, st) != MGX_OK) return MGX_ERR_LAUNCH;
    launch_wgrad_b<2, 1>(kRuns[2], a, d, J, ws, ntiles, st);
    launch_wgrad_b<1, 1>(kRuns[3], a, d, J, ws, ntiles, st);
    launch_wgrad_b<4, 1>(kRuns[5], a, d, J, ws, ntiles, st);
    launch_wgrad_b<1, 4>(kRuns[6], a, d, J, ws, ntiles, st);
    hipLaunchKernelGGL(gblock_unpack_kernel, dim3((N_PARAMS * 8 + 255) / 256), dim3(256), 0, st, g, J, ws);
    return mgx_launch_status();
}
