// Fused forward of one generator grow block (add_generator_block, trainer/models.py:358-399):
//   UpSampling2D -> Conv2D(128,(2,2),s2) -> 6 x {ConvT(16,(2,1)) -> Dense(32) -> ConvT(64,(1,2)) ->
//   Dense(128)} -> Add -> Dense(256) -> Dense(64) -> Dense(8) -> Dense(2)
// as ONE kernel on v_mfma_f32_32x32x2_f32.  A wave owns 32 low-resolution positions (= 128 output
// positions); every layer is computed transposed, D[channel][position] = W^T * X, so that the
// 32x32 accumulator tile of one layer (channel rows in the 16 registers x 2 lane halves, position
// on the lane) is, register by register, the B operand of the next layer's MFMAs: activations
// never leave the register file, no LDS, no intermediate tensor in HBM.  The contraction order
// inside a 32-channel tile follows the accumulator layout (step s <-> channel (s&3)+8(s>>2)+4h);
// a pack kernel rewrites the Keras-layout weights into that per-step, lane-ordered "fragment"
// order once per call (615 KB, L2 resident).  Sub-row a of the 2x2 output block is the outer
// loop (both sub-columns b share the branch prefix); the 128-wide featured tensor is recomputed
// per branch (16 MFMAs) instead of being held in registers.
#include "common.h"
#include "gblock_frag.h"
#include "gblock_rows.h"
#include "../../include/mgx.h"

#include <type_traits>

namespace {

// fragment offsets (units of 64 floats)
constexpr int OFF_F = 0;                 // featured conv: [ot 4][tap 4]
constexpr int OFF_BR = 16;               // 6 branches x 264
constexpr int BR_FR = 264;
constexpr int BR_K1 = 0;                 // [it 4][s 16]         rows (a,co)
constexpr int BR_D1 = 64;                // [s' 8]
constexpr int BR_K2 = 72;                // [b 2][ot 2][s 16]
constexpr int BR_D2 = 136;               // [ot 4][it 2][s 16]
constexpr int OFF_W1 = OFF_BR + 6 * BR_FR;   // 1600: [c 8][it 4][s 16]
constexpr int OFF_W2 = OFF_W1 + 512;         // 2112: [c 8][ot 2][s 16]
constexpr int OFF_W3 = OFF_W2 + 256;         // 2368: [it 2][s 16]
constexpr int OFF_W4 = OFF_W3 + 32;          // 2400: [s 4]
constexpr int N_FR = OFF_W4 + 4;             // 2404
constexpr int OFF_BSUM = N_FR * FR;          // 128 floats: sum of the six Dense(128) biases
constexpr int OFF_BBR = OFF_BSUM + 128;      // per branch: b1 (16) | c1 (32) | b2 (64)
constexpr int BBR = 112;
constexpr int OFF_B0 = OFF_BBR + 6 * BBR;    // featured bias (128)
constexpr int OFF_E1 = OFF_B0 + 128;         // tail biases 256 | 64 | 8 | 2
constexpr int OFF_E2 = OFF_E1 + 256;
constexpr int OFF_E3 = OFF_E2 + 64;
constexpr int OFF_E4 = OFF_E3 + 8;
constexpr int PACK_FLOATS = OFF_E4 + 8;

// ------------------------------------------------------------------------------------------
// pack: one thread per packed float
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gblock_pack_kernel(mgx_gblock_weights w, float* __restrict__ pack) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACK_FLOATS) return;
    if (idx >= OFF_BSUM) {
        float s = 0.f;
        if (idx < OFF_BBR) {
            for (int k = 0; k < 6; ++k) s += w.c2[k][idx - OFF_BSUM];
        } else if (idx < OFF_B0) {
            const int k = (idx - OFF_BBR) / BBR, e = (idx - OFF_BBR) % BBR;
            s = e < 16 ? w.b1[k][e] : (e < 48 ? w.c1[k][e - 16] : w.b2[k][e - 48]);
        } else if (idx < OFF_E1) s = w.b0[idx - OFF_B0];
        else if (idx < OFF_E2) s = w.e1[idx - OFF_E1];
        else if (idx < OFF_E3) s = w.e2[idx - OFF_E2];
        else if (idx < OFF_E4) s = w.e3[idx - OFF_E3];
        else s = (idx - OFF_E4) < 2 ? w.e4[idx - OFF_E4] : 0.f;
        pack[idx] = s;
        return;
    }
    const int fr = idx / FR, lane = idx % FR;
    const int i = lane & 31, h = lane >> 5;
    float v = 0.f;
    if (fr < OFF_BR) {                                   // featured: K0[tap][c][n], (2,2,2,128)
        const int ot = fr / 4, tap = fr % 4;
        v = w.k0[(tap * 2 + h) * 128 + ot * 32 + i];
    } else if (fr < OFF_W1) {
        const int k = (fr - OFF_BR) / BR_FR, f = (fr - OFF_BR) % BR_FR;
        if (f < BR_D1) {                                 // K1 (2,1,16,128): [a][co][ci]
            const int it = f / 16, s = f % 16;
            v = w.k1[k][((i >> 4) * 16 + (i & 15)) * 128 + it * 32 + chan(s, h)];
        } else if (f < BR_K2) {                          // D1 (16,32): [in][out]
            const int s = f - BR_D1;
            v = w.d1[k][chan(s, h) * 32 + i];
        } else if (f < BR_D2) {                          // K2 (1,2,64,32): [b][co][ci]
            const int g = f - BR_K2;
            const int b = g / 32, ot = (g % 32) / 16, s = g % 16;
            v = w.k2[k][(b * 64 + ot * 32 + i) * 32 + chan(s, h)];
        } else {                                         // D2 (64,128): [in][out]
            const int g = f - BR_D2;
            const int ot = g / 32, it = (g % 32) / 16, s = g % 16;
            v = w.d2[k][(it * 32 + chan(s, h)) * 128 + ot * 32 + i];
        }
    } else if (fr < OFF_W2) {                            // W1 (128,256)
        const int g = fr - OFF_W1;
        const int c = g / 64, it = (g % 64) / 16, s = g % 16;
        v = w.w1[(it * 32 + chan(s, h)) * 256 + c * 32 + i];
    } else if (fr < OFF_W3) {                            // W2 (256,64)
        const int g = fr - OFF_W2;
        const int c = g / 32, ot = (g % 32) / 16, s = g % 16;
        v = w.w2[(c * 32 + chan(s, h)) * 64 + ot * 32 + i];
    } else if (fr < OFF_W4) {                            // W3 (64,8): rows >= 8 are zero
        const int g = fr - OFF_W3;
        const int it = g / 16, s = g % 16;
        v = i < 8 ? w.w3[(it * 32 + chan(s, h)) * 8 + i] : 0.f;
    } else {                                             // W4 (8,2): rows >= 2 are zero
        const int s = fr - OFF_W4;
        v = i < 2 ? w.w4[chan(s, h) * 2 + i] : 0.f;
    }
    pack[frag_slot(fr, lane)] = v;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// Weight fragments stream through two register buffers of 16: while the MFMAs of one group run
// from buffer P, the fragments of the next group are loaded into buffer P^1 (their L2 latency is
// covered by the 16 x 64 cycles of the running group).  Buffer parities are compile-time: every
// rolled loop body holds an even number of groups.
// accumulator tile initialised with bias rows c(reg,h): registers 4g..4g+3 are the 4 consecutive
// rows 8g+4h.. (one 16-byte load each); groups g >= ngroups start at zero
__device__ __forceinline__ void init_bias(f32x16& acc, const float* __restrict__ bias, int h, int mask, int ngroups) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g < ngroups) v = *(const f32x4*)(bias + ((8 * g + 4 * h) & mask));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
    }
}

// One sub-row A of the 2x2 output block, both sub-columns.  On entry buffer PS holds the 16
// featured-conv fragments; on exit buffer PS^1 holds the 16 fragments at `after` (if LAST is false).
// With SAVE every intermediate tensor a weight gradient needs is written to the wave's tile of the
// activation buffer `act` (row map: gblock_rows.h); stores are placed behind the MFMAs of the
// following group so that they do not wait on the matrix pipeline.
template <int A, int PS, bool LAST, bool SAVE>
__device__ __forceinline__ void half_block(const float* __restrict__ pack, float x0, int lane, int h, bool pvalid,
                                           float* __restrict__ yrow, WBuf& B, const float* __restrict__ after,
                                           float* __restrict__ act) {
    const int r = lane & 31;
    constexpr int P0 = PS, P1 = PS ^ 1;
    f32x16 S[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) init_bias(S[b][ot], pack + OFF_BSUM + ot * 32, h, 127, 4);
    // featured = sum over the 4 taps of the 2x2/s2 conv on the (nearest) up-sampled input: the four taps
    // see the same low-resolution value, so each tap is one K=2 MFMA on (x[c=0], x[c=1])
    f32x16 F[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) init_bias(F[it], pack + OFF_B0 + it * 32, h, 127, 4);
    pf<P1, 16>(B, fptr(pack, OFF_BR + BR_K1), lane);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) F[it] = mfma32(B.v[P0][it * 4 + tap], x0, F[it]);
    MGX_FENCE();
    if (SAVE && A == 0) {
        act[(GB_X + h) * 32 + r] = x0;
#pragma unroll
        for (int it = 0; it < 4; ++it) save_rows<16>(act, GB_F + it * 32, F[it], r, h);
    }

    // six branches (26 groups each); every branch adds D2_k * o3_k into S, its bias is in bsum
    for (int k = 0; k < 6; ++k) {
        const int bf = OFF_BR + k * BR_FR;
        const float* bb = pack + OFF_BBR + k * BBR;
        f32x16 o1, o2;
        init_bias(o1, bb, h, 15, 4);
        pf<P0, 16>(B, fptr(pack, bf + BR_K1 + 16), lane); mma<P1, 0, 16>(o1, F[0], B); MGX_FENCE();
        pf<P1, 16>(B, fptr(pack, bf + BR_K1 + 32), lane); mma<P0, 0, 16>(o1, F[1], B); MGX_FENCE();
        pf<P0, 16>(B, fptr(pack, bf + BR_K1 + 48), lane); mma<P1, 0, 16>(o1, F[2], B); MGX_FENCE();
        pf<P1, 4>(B, fptr(pack, bf + BR_D1), lane);       mma<P0, 0, 16>(o1, F[3], B); MGX_FENCE();
        // Dense(32) on the 16 channels of sub-row A: registers 8A..8A+7 of o1
        init_bias(o2, bb + 16, h, 31, 4);
        float* abr = act + (GB_BR + k * GB_BR_ROWS) * 32;
        pf<P0, 4>(B, fptr(pack, bf + BR_D1 + 4), lane);   mma<P1, 8 * A, 4>(o2, o1, B);
        if (SAVE && A == 0) save_rows<16>(abr, GB_O1, o1, r, h);
        MGX_FENCE();
        pf<P1, 16>(B, fptr(pack, bf + BR_K2), lane);      mma<P0, 8 * A + 4, 4>(o2, o1, B); MGX_FENCE();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x16 o3[2];
            init_bias(o3[0], bb + 48, h, 63, 4);
            init_bias(o3[1], bb + 80, h, 63, 4);
            pf<P0, 16>(B, fptr(pack, bf + BR_K2 + (b * 2 + 1) * 16), lane); mma<P1, 0, 16>(o3[0], o2, B);
            if (SAVE && b == 0) save_rows<16>(abr, GB_O2 + A * 32, o2, r, h);
            MGX_FENCE();
            pf<P1, 16>(B, fptr(pack, bf + BR_D2), lane);                    mma<P0, 0, 16>(o3[1], o2, B); MGX_FENCE();
#pragma unroll
            for (int g = 0; g < 8; g += 2) {       // D2 groups g = ot*2 + it
                pf<P0, 16>(B, fptr(pack, bf + BR_D2 + (g + 1) * 16), lane);
                mma<P1, 0, 16>(S[b][g / 2], o3[0], B);
                if (SAVE && g == 0) save_rows<16>(abr, GB_O3 + (A * 2 + b) * 64, o3[0], r, h);
                if (SAVE && g == 2) save_rows<16>(abr, GB_O3 + (A * 2 + b) * 64 + 32, o3[1], r, h);
                MGX_FENCE();
                const float* nxt;
                if (g < 6) nxt = fptr(pack, bf + BR_D2 + (g + 2) * 16);
                else if (b == 0) nxt = fptr(pack, bf + BR_K2 + 32);
                else nxt = k < 5 ? fptr(pack, bf + BR_FR + BR_K1) : fptr(pack, OFF_W1);
                pf<P1, 16>(B, nxt, lane);
                mma<P0, 0, 16>(S[b][g / 2], o3[1], B); MGX_FENCE();
            }
        }
    }
    if (SAVE) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) save_rows<16>(act, GB_S + (A * 2 + b) * 128 + ot * 32, S[b][ot], r, h);
    }
    // tail (51 groups per sub-column): Dense(256) in 8 chunks of 32 rows, each chunk immediately consumed by
    // Dense(64); then Dense(8), Dense(2).  Entry: buffer Q holds W1 chunk 0 / input tile 0.
    auto tail = [&](auto qc, auto bc) {
        constexpr int Q = decltype(qc)::value, R = Q ^ 1, b = decltype(bc)::value;
        f32x16 T2[2];
        init_bias(T2[0], pack + OFF_E2, h, 63, 4);
        init_bias(T2[1], pack + OFF_E2 + 32, h, 63, 4);
        constexpr int ab = A * 2 + b;
        for (int c = 0; c < 8; ++c) {
            f32x16 T1;
            init_bias(T1, pack + OFF_E1 + c * 32, h, 255, 4);
            const int w1 = OFF_W1 + c * 64, w2 = OFF_W2 + c * 32;
            pf<R, 16>(B, fptr(pack, w1 + 16), lane); mma<Q, 0, 16>(T1, S[b][0], B); MGX_FENCE();
            pf<Q, 16>(B, fptr(pack, w1 + 32), lane); mma<R, 0, 16>(T1, S[b][1], B); MGX_FENCE();
            pf<R, 16>(B, fptr(pack, w1 + 48), lane); mma<Q, 0, 16>(T1, S[b][2], B); MGX_FENCE();
            pf<Q, 16>(B, fptr(pack, w2), lane);      mma<R, 0, 16>(T1, S[b][3], B); MGX_FENCE();
            pf<R, 16>(B, fptr(pack, w2 + 16), lane); mma<Q, 0, 16>(T2[0], T1, B); MGX_FENCE();
            pf<Q, 16>(B, c < 7 ? fptr(pack, w1 + 64) : fptr(pack, OFF_W3), lane);
            mma<R, 0, 16>(T2[1], T1, B);
            if (SAVE) save_rows<16>(act, GB_T1 + ab * 256 + c * 32, T1, r, h);
            MGX_FENCE();
        }
        f32x16 T3, O;
        init_bias(T3, pack + OFF_E3, h, 7, 1);
        init_bias(O, pack + OFF_E4, h, 7, 1);
        pf<R, 16>(B, fptr(pack, OFF_W3 + 16), lane); mma<Q, 0, 16>(T3, T2[0], B); MGX_FENCE();
        pf<Q, 4>(B, fptr(pack, OFF_W4), lane);       mma<R, 0, 16>(T3, T2[1], B);
        if (SAVE) {
            save_rows<16>(act, GB_T2 + ab * 64, T2[0], r, h);
            save_rows<16>(act, GB_T2 + ab * 64 + 32, T2[1], r, h);
        }
        MGX_FENCE();
        if (b == 0) pf<R, 16>(B, fptr(pack, OFF_W1), lane);
        else if (!LAST) pf<R, 16>(B, after, lane);
        mma<Q, 0, 4>(O, T3, B);
        if (SAVE) save_rows<4>(act, GB_T3 + ab * 8, T3, r, h);
        MGX_FENCE();
        // rows 0,1 (the two output channels) live in registers 0,1 of lane half 0
        if (pvalid && h == 0 && yrow) *(float2*)(yrow + b * 2) = make_float2(O[0], O[1]);
    };
    tail(std::integral_constant<int, P1>{}, std::integral_constant<int, 0>{});
    tail(std::integral_constant<int, P0>{}, std::integral_constant<int, 1>{});
}

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void gblock_fwd_kernel(const float* __restrict__ pack,
                                                            const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ act, long pos0, long npix, int hl,
                                                            int wl, int store_y) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long p = pos0 + ((long)blockIdx.x * 4 + wv) * 32 + r;    // side-buffer tiles are counted from pos0
    const bool pvalid = p < npix;
    const long pc = pvalid ? p : 0;
    // B operand of the featured conv: lane (position r, half h) supplies channel h of its position
    // positions past the end contribute activations of a zero input; their gradients are zero
    const float x0 = pvalid ? x[pc * 2 + h] : 0.f;
    float* atile = SAVE ? act + ((long)blockIdx.x * 4 + wv) * (GB_ROWS * 32L) : nullptr;
    const long row = pc / wl;                 // n*hl + i
    const int j = (int)(pc - row * wl);
    float* yrow0 = store_y ? y + ((row * 2) * (2L * wl) + 2L * j) * 2 : nullptr;      // output row 2i, column 2j
    float* yrow1 = store_y ? yrow0 + (2L * wl) * 2 : nullptr;                         // output row 2i+1
    WBuf B;
    B.touch[0] = B.touch[1] = 0.f;
    pf<0, 16>(B, fptr(pack, OFF_F), lane);
    half_block<0, 0, false, SAVE>(pack, x0, lane, h, pvalid, yrow0, B, fptr(pack, OFF_F), atile);
    half_block<1, 1, true, SAVE>(pack, x0, lane, h, pvalid, yrow1, B, nullptr, atile);
}

}  // namespace

extern "C" int64_t mgx_gblock_pack_floats(void) { return PACK_FLOATS + FRAG_SLACK; }

extern "C" int64_t mgx_gblock_side_floats_n(int64_t npos) {
    return npos <= 0 ? 0 : ((npos + 127) / 128) * 4 * (GB_ROWS * 32L);
}

extern "C" int64_t mgx_gblock_side_floats(int64_t nimg, int hl, int wl) {
    if (nimg <= 0 || hl <= 0 || wl <= 0) return 0;
    return mgx_gblock_side_floats_n((long)nimg * hl * wl);
}

extern "C" int mgx_gblock_fwd_f32(const mgx_gblock_weights* wp, float* pack, const float* x, float* y, float* act,
                                  int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream) {
    if (!wp || !pack || !x || (!y && !act) || nimg <= 0 || hl <= 0 || wl <= 0) return MGX_ERR_ARG;
    const long npix = (long)nimg * hl * wl;
    if (pos0 < 0 || pos0 >= npix || (pos0 & 127)) return MGX_ERR_ARG;
    const long end = (npos <= 0 || pos0 + npos > npix) ? npix : pos0 + npos;
    const mgx_gblock_weights w = *wp;
    const void* const* ptrs = (const void* const*)&w;
    for (size_t i = 0; i < sizeof(w) / sizeof(void*); ++i)
        if (!ptrs[i]) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (pos0 == 0)      // the packed fragments of a call sequence over ranges are filled by its first call
        hipLaunchKernelGGL(gblock_pack_kernel, dim3((PACK_FLOATS + 255) / 256), dim3(256), 0, st, w, pack);
    const long blocks = (end - pos0 + 127) / 128;
    if (blocks > 0x7fffffffL) return MGX_ERR_ARG;
    if (act)
        hipLaunchKernelGGL(gblock_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, pack, x, y, act,
                           (long)pos0, end, hl, wl, y ? 1 : 0);
    else
        hipLaunchKernelGGL(gblock_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, pack, x, y, act,
                           (long)pos0, end, hl, wl, 1);
    return mgx_launch_status();
}
