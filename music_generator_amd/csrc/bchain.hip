// A chain of two or three consecutive convolution + activation layers of a discriminator block
// (trainer/models.py:259-276: Conv2D(32, 3x3 | 2x2, 'same') + SoftRectifier, 32 -> 32 channels each) in ONE
// launch, for forward passes that keep no tape, bf16 NHWC in and out: {3x3, 3x3, 2x2}, {3x3, 3x3}, {3x3, 2x2}.
// The outputs of the inner layers never reach HBM.
//
// Row streaming instead of 2-D tiles: a WAVE owns a strip of CW = 29 / 30 / 31 output columns of one image and
// walks down its rows.  At step s it computes row s of layer 1, row s-1 of layer 2 and row s-2 of layer 3, each
// as ONE 32-position MFMA tile D[output channel][position] (18 or 8 v_mfma_f32_32x32x16_bf16), from rings of the
// last three (two) rows of the layer below that live in the wave's own LDS: no row is computed twice (a 2-D
// tile of a fused pair re-computes its halo rows), only the 34 - CW halo columns are, and since nothing is
// shared between waves there is no barrier in the kernel.  The layers' weights stay in registers for the whole
// launch as MFMA A fragments (72 + 72 + 32 registers), so the LDS pipe carries only the B operand (one 16-byte
// read per MFMA; both operands from LDS would need the LDS's whole bandwidth at the full MFMA rate).
// Input rows are requested one step ahead (global -> registers -> LDS ring).  Zero padding of every layer is
// the reference's: ring rows outside the image and positions outside the image's columns hold zeros, not
// values computed from padded input.
// A strip may be cut into pieces of rows, one wave each, so that the workgroups of a launch fill the chip's 256
// CUs evenly (a workgroup occupies a CU alone; 828 workgroups of equal length take as long as 1024).  A piece
// starts one row above its first output row: the rows of the inner layers it computes before their own inputs
// are complete feed only output rows above the piece, which it does not store.
// LDS position layout, operand fragments, accumulation order (bias first, taps row-major, two 16-channel steps
// a tap), activation and the lane exchange before the stores are those of bconv.hip: results are bit-identical
// to the layer-by-layer launches of mgx_conv_gemm_bf16.
#include <cstdlib>
#include <type_traits>
#include "bf16.h"
#include "../../include/mgx.h"

namespace {

constexpr int CH_ROWP = 34;                // positions of a ring row (input: all used; layer rows: 32)
constexpr int CH_ROWB = CH_ROWP * 64;      // bytes
constexpr int CH_BIAS_B = 3 * 32 * 4;

template <int NL>
constexpr int chain_wave_lds() { return (3 + 3 + (NL == 3 ? 2 : 0)) * CH_ROWB + CH_BIAS_B; }
// valid output columns of a strip: the 34 input columns minus what the kernels consume
template <int NL, int KLAST>
constexpr int chain_cw() { return 34 - 2 * (NL - 1) - (KLAST - 1); }
// A fragments that do not fit the accumulation registers are read from a copy in LDS the workgroup shares: with
// two waves on a SIMD a wave has 256 registers, which the compiler splits into 128 vector + 128 accumulation
// registers once inline asm names the latter.  128 accumulation registers = 32 fragments: all 18 of the first
// layer and the first 14 of a 3x3 second layer; its last 4 and the third layer's 8 come from LDS.
template <int NL, int KLAST>
constexpr int chain_l1_acc() { return (NL == 3 || KLAST == 3) ? 14 : KLAST * KLAST * 2; }     // second layer: in registers
template <int NL, int KLAST>
constexpr int chain_shared_frags() {
    return ((NL == 3 || KLAST == 3) ? 18 - 14 : 0) + (NL == 3 ? KLAST * KLAST * 2 : 0);
}
template <int NL, int KLAST, int WAVES>
constexpr int chain_lds() { return chain_wave_lds<NL>() * WAVES + chain_shared_frags<NL, KLAST>() * 1024; }

struct ChainDesc {
    int nimg, h, w, strips;                // strips per image row
    int nseg, seg_rows;                    // a strip is walked in nseg pieces of seg_rows output rows, one wave each
    float alpha[3];
};

// acc += A * B with the A fragment (a layer's weights, constant for the whole launch) in the accumulation
// registers and the accumulator in vector registers: 176 registers of weights beside the B fragments, row
// buffers and epilogue values do not fit the 256 vector registers, and left to itself the compiler keeps the
// weights there and serialises every LDS read with its MFMA to save registers.
__device__ __forceinline__ void mfma_wacc(f32x16& acc, const u32x4& wa, const u32x4& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}

__device__ __forceinline__ int cxoff(int p, int q) { return p * 64 + ((q ^ ((p >> 2) & 3)) << 4); }

// Keras kernel (k, k, 32, 32) float32 -> the lane's bf16 A fragments [tap][16-channel step], placed in the
// accumulation registers: a value the compiler computes lives in vector registers and would be copied into an
// accumulation register in front of every MFMA that names one, so the fragments take the one path that DEFINES
// them there: through the wave's own LDS (8 fragments at a time) and back with ds_read_b128 into a[..].
template <int NF>
__device__ __forceinline__ void chain_weights(const float* __restrict__ w, int r, int h, int lane, unsigned char* stage,
                                              u32x4 (&f)[NF]) {
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(stage + lane * 16);
#pragma unroll
    for (int c0 = 0; c0 < NF; c0 += 8) {
#pragma unroll
        for (int i = c0; i < c0 + 8 && i < NF; ++i) {
            const int tap = i >> 1, ks = i & 1;
            const float* p = w + ((tap * 32 + 16 * ks + 8 * h) * 32 + r);
            u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bf_pack(p[(2 * e) * 32], p[(2 * e + 1) * 32]);
            *(u32x4*)(stage + (i - c0) * 1024 + lane * 16) = v;
        }
#pragma unroll
        for (int i = c0; i < c0 + 8 && i < NF; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(f[i]) : "v"(la), "n"((i - c0) * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// TAPE: the outputs of the inner layers are also written to t0 (layer 1) and t1 (layer 2 of three), for forward
// passes whose backward reads every layer's output: each wave stores the columns and rows it owns (the halo
// columns and the rows above a piece are the neighbours').
template <int NL, int KLAST, int WAVES, bool TAPE>
__global__ __launch_bounds__(64 * WAVES) void bchain_kernel(ChainDesc d, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ w0, const float* __restrict__ b0,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           bf16_t* __restrict__ y, bf16_t* __restrict__ t0,
                                                           bf16_t* __restrict__ t1) {
    static_assert(NL == 2 || KLAST == 2, "three layers end in the 2x2 convolution");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CW = chain_cw<NL, KLAST>();
    constexpr int K1 = NL == 3 ? 3 : KLAST;          // kernel of the second layer
    constexpr int PADS = 1 + (K1 - 1) / 2 + (NL == 3 ? (KLAST - 1) / 2 : 0);     // sum of the leading pads
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = d.h, W = d.w;
    // ---- shared A fragments: [second layer's fragments NA1 ..][third layer's] ----
    constexpr int K1_ = NL == 3 ? 3 : KLAST;
    constexpr int NA1 = chain_l1_acc<NL, KLAST>(), NF1 = K1_ * K1_ * 2;
    unsigned char* const wl1 = lds + WAVES * chain_wave_lds<NL>() - NA1 * 1024;     // fragment f of layer 2 at wl1 + f KiB
    unsigned char* const wl2 = wl1 + NF1 * 1024;
    if constexpr (chain_shared_frags<NL, KLAST>() > 0) {
        auto share = [&](const float* __restrict__ w, int f0, int f1, unsigned char* base) {
            // fragment f = tap * 2 + (k >> 4) holds channels k = 16 (f & 1) .. + 15 of tap f >> 1
            for (int i = threadIdx.x; i < (f1 - f0) * 512; i += 64 * WAVES) {
                const int n = i & 31, kk = (i >> 5) & 15, f = f0 + (i >> 9);
                const int k = 16 * (f & 1) + kk, tap = f >> 1;
                const int ln = ((k >> 3) & 1) * 32 + n;
                *(bf16_t*)(base + ((f * 64 + ln) * 8 + (k & 7)) * 2) = (bf16_t)w[(tap * 32 + k) * 32 + n];
            }
        };
        if (NA1 < NF1) share(w1, NA1, NF1, wl1);
        if constexpr (NL == 3) share(w2, 0, KLAST * KLAST * 2, wl2);
        __syncthreads();
    }
    const long gw = (long)blockIdx.x * WAVES + wv;
    if (gw >= (long)d.nimg * d.strips * d.nseg) return;       // no barrier anywhere below
    const int per_img = d.strips * d.nseg;
    const int img = (int)(gw / per_img);
    const int seg = (int)(gw - (long)img * per_img) / d.strips;
    const int c0 = ((int)(gw - (long)img * per_img) - seg * d.strips) * CW;
    const int r0 = seg * d.seg_rows, r1 = min(d.h, r0 + d.seg_rows);          // output rows of this wave
    const int s_start = max(0, r0 - 1), s_end = min(d.h + NL - 2, r1 + NL - 2);
    const int o_in = c0 - PADS, o1 = o_in + 1, o2 = o1 + (K1 - 1) / 2;

    unsigned char* const my = lds + wv * chain_wave_lds<NL>();
    float* const bl = (float*)(my + chain_wave_lds<NL>() - CH_BIAS_B);
    if (lane < 32) {
        bl[lane] = b0[lane];
        bl[32 + lane] = b1[lane];
        if (NL == 3) bl[64 + lane] = b2[lane];
    }
    // ring slots (byte addresses inside the wave's LDS), rotated every step
    unsigned char *i0 = my, *i1 = my + CH_ROWB, *i2 = my + 2 * CH_ROWB;                  // input rows s-1, s, s+1
    unsigned char *p0 = my + 3 * CH_ROWB, *p1 = my + 4 * CH_ROWB, *p2 = my + 5 * CH_ROWB; // layer-1 rows s-2, s-1, s
    unsigned char *q0 = my + 6 * CH_ROWB, *q1 = my + 7 * CH_ROWB;                        // layer-2 rows s-2, s-1

    u32x4 wf0[18], wf1[NA1];
    chain_weights<18>(w0, r, h, lane, my, wf0);
    chain_weights<NA1>(w1, r, h, lane, my, wf1);

    // ---- input rows: 34 positions x 4 quads = 136 16-byte pieces, piece lane + 64 i ----
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (long)img * H * W * 32), 0, (unsigned)((long)H * W * 64), 0x00020000);
    unsigned pcol[3];
    int pdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pc = lane + 64 * i, pos = pc >> 2, q = pc & 3;
        const int col = o_in + pos;
        pcol[i] = (pos < CH_ROWP && (unsigned)col < (unsigned)W) ? (unsigned)(col * 64 + q * 16) : 0xfffffff0u;
        pdst[i] = cxoff(pos, q);
    }
    auto row_load = [&](int row, u32x4 (&sv)[3]) {
        const bool ok = (unsigned)row < (unsigned)H;
        const unsigned rb = ok ? (unsigned)row * (unsigned)W * 64u : 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = (ok && pcol[i] != 0xfffffff0u) ? rb + pcol[i] : 0xfffffff0u;
            sv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto row_write = [&](unsigned char* slot, const u32x4 (&sv)[3]) {
        *(u32x4*)(slot + pdst[0]) = sv[0];
        *(u32x4*)(slot + pdst[1]) = sv[1];
        if (lane < 8) *(u32x4*)(slot + pdst[2]) = sv[2];
    };
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int wq0 = cxoff(r, h), wq1 = cxoff(r, 2 + h);          // where the lane's two 16-byte results of a row go
    auto ring_write = [&](unsigned char* slot, u32x4 s0, u32x4 s1) {
        *(u32x4*)(slot + wq0) = s0;
        *(u32x4*)(slot + wq1) = s1;
    };
    // B-operand offsets of position r + b, 16-channel step ks
    int bo[3][2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bo[b][ks] = cxoff(r + b, 2 * ks + h);

    // registers 4g .. 4g+3 of an accumulator tile are output channels 8g + 4h .. 8g + 4h + 3
    auto acc_init = [&](f32x16& acc, const float* b32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(b32 + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
        }
    };
    // Inner layers: activation, then straight into the ring as four 8-byte pieces (channels 8g + 4h .. + 3 of the
    // lane's position are bytes [16g + 8h, 16g + 8h + 8) of its 64): no exchange between the lane halves.
    const int wg8 = cxoff(r, 0) + 8 * h, wsw = (r >> 2) & 3;
    // tp / trow / tcol: where the row goes on the tape (TAPE only; tcol is out of range for lanes that own no column of it)
    auto finish_ring = [&](f32x16& a, float ia, float k1, unsigned char* slot, bool inside, bf16_t* tp, int trow,
                           unsigned tcol) {
        mfma_result_wait(a);
        const bool taped = TAPE && trow >= r0 && trow < r1;
        __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0u, 0x00020000);
        if (taped)
            rt = __builtin_amdgcn_make_buffer_rsrc((void*)(tp + ((long)img * H + trow) * W * 32), 0, (unsigned)W * 64u,
                                                   0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float y0, y1, y2, y3, dd;
            act_bf16(a[4 * g], ia, k1, y0, dd);
            act_bf16(a[4 * g + 1], ia, k1, y1, dd);
            act_bf16(a[4 * g + 2], ia, k1, y2, dd);
            act_bf16(a[4 * g + 3], ia, k1, y3, dd);
            u32x2 o = {bf_pack(y0, y1), bf_pack(y2, y3)};
            if (!inside) o = (u32x2){0u, 0u};
            *(u32x2*)(slot + wg8 + ((g ^ wsw) << 4) - ((0 ^ wsw) << 4)) = o;
            // channels 8g + 4h .. + 3 of the lane's position: bytes [16g + 8h, + 8) of its 64
            if (taped) __builtin_amdgcn_raw_buffer_store_b64(o, rt, tcol + 16 * g, 0, 0);
        }
    };
    auto finish = [&](f32x16& a, float ia, float k1, u32x4& s0, u32x4& s1) {
        mfma_result_wait(a);
        acc_rows_interleaved(a);
        unsigned o[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            float y0, y1, d0, d1;
            act_bf16(a[i], ia, k1, y0, d0);
            act_bf16(a[i + 1], ia, k1, y1, d1);
            o[i >> 1] = bf_pack(y0, y1);
        }
        s0 = (u32x4){o[0], o[1], o[2], o[3]};
        s1 = (u32x4){o[4], o[5], o[6], o[7]};
    };
    // One layer row: acc += sum over taps.  The B fragments of a tap row (KK taps x two 16-channel steps) are read
    // as one batch, and the next tap row's batch is requested before the MFMAs of the current one are issued: a read
    // placed right in front of its MFMA costs the whole LDS latency every time.  Fragments NACC and up of the layer's
    // weights are read from the workgroup's LDS copy (at wl + fragment KiB) with the batch they belong to.
    auto conv_row = [&](auto kk_tag, auto nacc_tag, const auto& wf, const unsigned char* wl,
                        const unsigned char* const (&rows)[3], f32x16& acc) {
        constexpr int KK = decltype(kk_tag)::value, NACC = decltype(nacc_tag)::value, NB = KK * 2;
        const unsigned char* wlane = wl + lane * 16;
        u32x4 cur[NB], nxt[NB], wa[NB], wn[NB];
        auto batch = [&](int a, u32x4 (&bb)[NB], u32x4 (&ww)[NB]) {
#pragma unroll
            for (int b = 0; b < KK; ++b)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bb[b * 2 + ks] = *(const u32x4*)(rows[a] + bo[b][ks]);
                    if (a * NB + b * 2 + ks >= NACC) ww[b * 2 + ks] = *(const u32x4*)(wlane + (a * NB + b * 2 + ks) * 1024);
                }
        };
        batch(0, cur, wa);
        // the wave in its MFMA phase issues first, the activation phase of the SIMD's other wave fills in behind it
        // (measured on the three-layer chain at 64 x 32 x 6000: 1.05 -> 0.90 ms)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < KK; ++a) {
            if (a + 1 < KK) batch(a + 1, nxt, wn);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (a * NB + i < NACC) mfma_wacc(acc, wf[a * NB + i < NACC ? a * NB + i : 0], cur[i]);
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa[i]), "v"(cur[i]));
            }
            if (a + 1 < KK) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    cur[i] = nxt[i];
                    if ((a + 1) * NB + i >= NACC) wa[i] = wn[i];
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    const float ia0 = 1.0f / d.alpha[0], k10 = 1.0f + ia0;
    const float ia1 = 1.0f / d.alpha[1], k11 = 1.0f + ia1;
    const float ia2 = NL == 3 ? 1.0f / d.alpha[2] : 0.f, k12 = 1.0f + ia2;
    const bool in1 = (unsigned)(o1 + r) < (unsigned)W;          // the lane's position of a layer-1 / layer-2 row
    const bool in2 = (unsigned)(o2 + r) < (unsigned)W;          // lies inside the image
    const unsigned ocol = (r < CW && c0 + r < W) ? (unsigned)((c0 + r) * 64 + 16 * h) : 0x80000000u;
    // the lane's position of a layer-1 / layer-2 row is column o1 + r / o2 + r; the strip owns columns c0 .. c0 + CW - 1
    const unsigned tcol1 = (o1 + r >= c0 && o1 + r < c0 + CW && o1 + r < W) ? (unsigned)((o1 + r) * 64 + 8 * h) : 0x80000000u;
    const unsigned tcol2 = (o2 + r >= c0 && o2 + r < c0 + CW && o2 + r < W) ? (unsigned)((o2 + r) * 64 + 8 * h) : 0x80000000u;
    bf16_t* const yimg = y + (long)img * H * W * 32;
    auto row_store = [&](int row, u32x4 s0, u32x4 s1) {
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void*)(yimg + (long)row * W * 32), 0, (unsigned)W * 64u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(s0, ry, ocol, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(s1, ry, ocol + 32, 0, 0);
    };

    // ---- prologue: input rows s_start - 1 and s_start staged (a row outside the image loads as zeros), the next one
    // in flight; layer-1 row s_start - 1 is zero padding in the first piece and feeds nothing that is stored in the others
    u32x4 svb[3];
    row_load(s_start - 1, svb);
    row_write(i0, svb);
    row_load(s_start, svb);
    ring_write(p1, zero4, zero4);
    row_write(i1, svb);
    row_load(s_start + 1, svb);
    // from here: i0 = row s-1, i1 = row s, svb = row s+1 (in flight: requested one step ahead; with two waves on
    // a SIMD the other wave's step covers what is left of the latency)

    for (int s = s_start; s <= s_end; ++s) {
        row_write(i2, svb);
        row_load(s + 2, svb);
        // ---- layer 1, row s ----
        if (s < H) {
            f32x16 acc;
            acc_init(acc, bl);
            const unsigned char* const rows[3] = {i0, i1, i2};
            conv_row(std::integral_constant<int, 3>{}, std::integral_constant<int, 18>{}, wf0, wl1, rows, acc);
            finish_ring(acc, ia0, k10, p2, in1, t0, s, tcol1);
        } else {
            ring_write(p2, zero4, zero4);
        }
        // ---- layer 2, row s-1 ----
        if (s > s_start && s - 1 < H) {
            f32x16 acc;
            acc_init(acc, bl + 32);
            const unsigned char* const rows[3] = {K1 == 3 ? p0 : p1, K1 == 3 ? p1 : p2, p2};
            conv_row(std::integral_constant<int, K1>{}, std::integral_constant<int, NA1>{}, wf1, wl1, rows, acc);
            if constexpr (NL == 2) {
                u32x4 s0, s1;
                finish(acc, ia1, k11, s0, s1);
                if (s - 1 >= r0) row_store(s - 1, s0, s1);
            } else {
                finish_ring(acc, ia1, k11, q1, in2, t1, s - 1, tcol2);
            }
        } else if (NL == 3 && s > s_start) {
            ring_write(q1, zero4, zero4);
        }
        // ---- layer 3 (2x2), row s-2 ----
        if constexpr (NL == 3) {
            if (s >= s_start + 2) {
                f32x16 acc;
                acc_init(acc, bl + 64);
                const unsigned char* const rows[3] = {q0, q1, q1};
                conv_row(std::integral_constant<int, KLAST>{}, std::integral_constant<int, 0>{}, wf0, wl2, rows, acc);
                u32x4 s0, s1;
                finish(acc, ia2, k12, s0, s1);
                if (s - 2 >= r0) row_store(s - 2, s0, s1);
            }
        }
        // rotate the rings
        unsigned char* t = i0; i0 = i1; i1 = i2; i2 = t;
        t = p0; p0 = p1; p1 = p2; p2 = t;
        t = q0; q0 = q1; q1 = t;
    }
}


// ------------------------------------------------------------------------------------------
// The same chain as a software pipeline ACROSS layers, one wave per SIMD (4 waves a workgroup, every weight
// fragment in accumulation registers).  A wave that issues the six instructions of an activation value one after
// the other behind an MFMA pays for the MFMA in full (60 cycles for MFMA + value); issued as a rotating pipeline
// (every MFMA slot one stage of six different values) the pair takes 46.  For that the MFMAs of a layer row must
// have another row's epilogue to sit between, so the layers run further apart than in bchain_kernel:
// at step s   phase A: MFMAs of layer 1, row s      | epilogue of the LAST layer's row of the step before (stores)
//                                                     | and the second half of layer 2's ring epilogue (three layers)
//             phase B: MFMAs of layer 2, row s - 2  | ring epilogue of layer 1, row s
//             phase C: MFMAs of layer 3, row s - 4  | first half of the ring epilogue of layer 2, row s - 2
// with four ring rows of layer 1 and three of layer 2 (a row is read one step after it is complete).  The LDS reads
// of a tap row are issued one per MFMA slot a tap row ahead, the first tap row of a phase during the last of the
// phase before.  Arithmetic, fragments and accumulation order are bchain_kernel's: bit-identical results.
// What it buys is little: with one B-operand read from LDS per MFMA (12-15 cycles of issue each) and 8.5 vector
// instructions per MFMA slot the instruction stream of a step is ~1.7x its MFMA time however it is ordered, and
// one wave per SIMD (176 accumulation + 124-136 vector registers) has no second wave to fill what is left.
// ------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void cstatic_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cstatic_for<I + 1, N>(f);
    }
}

template <int NL>
constexpr int pchain_wave_lds() { return (3 + 4 + (NL == 3 ? 3 : 0)) * CH_ROWB + CH_BIAS_B; }

template <int NL, int KLAST, bool TAPE>
__global__ __launch_bounds__(256) void bchainp_kernel(ChainDesc d, const bf16_t* __restrict__ x,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      bf16_t* __restrict__ y, bf16_t* __restrict__ t0,
                                                      bf16_t* __restrict__ t1) {
    static_assert(NL == 2 || KLAST == 2, "three layers end in the 2x2 convolution");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int CW = chain_cw<NL, KLAST>();
    constexpr int K1 = NL == 3 ? 3 : KLAST;          // kernel of the second layer
    constexpr int PADS = 1 + (K1 - 1) / 2 + (NL == 3 ? (KLAST - 1) / 2 : 0);
    constexpr int NF1 = K1 * K1 * 2, NF2 = NL == 3 ? KLAST * KLAST * 2 : 2;
    constexpr int LAG = NL == 3 ? 4 : 2;             // the last layer's row of step s is row s - LAG
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = d.h, W = d.w;
    const long gw = (long)blockIdx.x * 4 + wv;
    if (gw >= (long)d.nimg * d.strips * d.nseg) return;       // no barrier anywhere
    const int per_img = d.strips * d.nseg;
    const int img = (int)(gw / per_img);
    const int seg = (int)(gw - (long)img * per_img) / d.strips;
    const int c0 = ((int)(gw - (long)img * per_img) - seg * d.strips) * CW;
    const int r0 = seg * d.seg_rows, r1 = min(H, r0 + d.seg_rows);            // output rows of this wave
    const int s_start = max(0, r0 - 1), s_last = r1 - 1 + LAG;
    const int o_in = c0 - PADS, o1 = o_in + 1, o2 = o1 + (K1 - 1) / 2;

    unsigned char* const my = lds + wv * pchain_wave_lds<NL>();
    float* const bl = (float*)(my + pchain_wave_lds<NL>() - CH_BIAS_B);
    if (lane < 32) {
        bl[lane] = b0[lane];
        bl[32 + lane] = b1[lane];
        if (NL == 3) bl[64 + lane] = b2[lane];
    }
    unsigned char *i0 = my, *i1 = my + CH_ROWB, *i2 = my + 2 * CH_ROWB;                  // input rows s-1, s, s+1
    unsigned char *p0 = my + 3 * CH_ROWB, *p1 = my + 4 * CH_ROWB, *p2 = my + 5 * CH_ROWB, *p3 = my + 6 * CH_ROWB;
    unsigned char *q0 = my + 7 * CH_ROWB, *q1 = my + 8 * CH_ROWB, *q2 = my + 9 * CH_ROWB;    // layer-2 rows s-4, s-3, s-2

    u32x4 wf0[18], wf1[NF1], wf2[NF2];
    chain_weights<18>(w0, r, h, lane, my, wf0);
    chain_weights<NF1>(w1, r, h, lane, my, wf1);
    if constexpr (NL == 3) chain_weights<NF2>(w2, r, h, lane, my, wf2);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (long)img * H * W * 32), 0, (unsigned)((long)H * W * 64), 0x00020000);
    unsigned pcol[3];
    int pdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pc = lane + 64 * i, pos = pc >> 2, q = pc & 3;
        const int col = o_in + pos;
        pcol[i] = (pos < CH_ROWP && (unsigned)col < (unsigned)W) ? (unsigned)(col * 64 + q * 16) : 0xfffffff0u;
        pdst[i] = cxoff(pos, q);
    }
    auto row_load = [&](int row, u32x4 (&sv)[3]) {
        const bool ok = (unsigned)row < (unsigned)H;
        const unsigned rb = ok ? (unsigned)row * (unsigned)W * 64u : 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = (ok && pcol[i] != 0xfffffff0u) ? rb + pcol[i] : 0xfffffff0u;
            sv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto row_write = [&](unsigned char* slot, const u32x4 (&sv)[3]) {
        *(u32x4*)(slot + pdst[0]) = sv[0];
        *(u32x4*)(slot + pdst[1]) = sv[1];
        if (lane < 8) *(u32x4*)(slot + pdst[2]) = sv[2];
    };
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    int bo[3][2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bo[b][ks] = cxoff(r + b, 2 * ks + h);
    auto acc_init = [&](f32x16& acc, const float* b32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(b32 + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
        }
    };
    const float ia0 = 1.0f / d.alpha[0], ia1 = 1.0f / d.alpha[1], ia2 = NL == 3 ? 1.0f / d.alpha[2] : 0.f;
    const bool in1 = (unsigned)(o1 + r) < (unsigned)W, in2 = (unsigned)(o2 + r) < (unsigned)W;
    const unsigned ocol = (r < CW && c0 + r < W) ? (unsigned)((c0 + r) * 64 + 16 * h) : 0x80000000u;
    const unsigned tcol1 = (o1 + r >= c0 && o1 + r < c0 + CW && o1 + r < W) ? (unsigned)((o1 + r) * 64 + 8 * h) : 0x80000000u;
    const unsigned tcol2 = (o2 + r >= c0 && o2 + r < c0 + CW && o2 + r < W) ? (unsigned)((o2 + r) * 64 + 8 * h) : 0x80000000u;
    const int wg8 = cxoff(r, 0) + 8 * h, wsw = (r >> 2) & 3;
    bf16_t* const yimg = y + (long)img * H * W * 32;

    // ---- one stage of the activation tanh(v) + v / alpha of act_bf16, value by value ----
    //   stage 0: m = c v   1: e = exp2(m)   2: s = 1 + e   3: q = rcp(s)   4: t = 1 - 2 q   5: y = v / alpha + t
    // (stages 0-4 return the new intermediate, stage 5 the activation's value)
    auto act_stage = [&](int stage, float v, float tmp, float ia) -> float {
        if (stage == 0) return 2.885390081777927f * v;
        if (stage == 1) return __builtin_amdgcn_exp2f(tmp);
        if (stage == 2) return 1.0f + tmp;
        if (stage == 3) return __builtin_amdgcn_rcpf(tmp);
        if (stage == 4) return __builtin_fmaf(-2.0f, tmp, 1.0f);
        return __builtin_fmaf(v, ia, tmp);
    };
    // Piece P (0 .. 20) of the epilogue of an inner layer's row: activation as a rotating pipeline (stage k of value
    // P - k), each group of four values (channels 8g + 4h .. + 3 of the lane's position) rounded and written to the
    // ring row (and to the tape) as soon as it is complete.
    constexpr int NP_RING = 21, NP_LAST = 23;
    auto ring_piece = [&](auto P_, f32x16& a, float (&tmp)[16], float ia, unsigned char* slot, bool inside, bool row_ok,
                          bf16_t* tp, int trow, unsigned tcol) {
        constexpr int P = decltype(P_)::value;
        if constexpr (P == 0) mfma_result_wait(a);
        cstatic_for<0, 6>([&](auto k_) {
            constexpr int k = decltype(k_)::value, v = P - k;
            if constexpr (v >= 0 && v < 16) {
                const float nv = act_stage(k, a[v], tmp[v], ia);
                if constexpr (k < 5) tmp[v] = nv;
                else a[v] = nv;
            }
        });
        if constexpr (P >= 8 && (P & 3) == 0) {
            constexpr int g = (P - 8) >> 2;
            u32x2 o = {bf_pack(a[4 * g], a[4 * g + 1]), bf_pack(a[4 * g + 2], a[4 * g + 3])};
            if (!(inside && row_ok)) o = (u32x2){0u, 0u};
            *(u32x2*)(slot + wg8 + ((g ^ wsw) << 4) - ((0 ^ wsw) << 4)) = o;
            if constexpr (TAPE) {
                // no branch around the store (the compiler would sink the whole activation into it): a row that is
                // not this wave's is dropped through an out-of-range offset
                const bool mine = trow >= r0 && trow < r1;
                const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(tp + ((long)img * H + (mine ? trow : 0)) * W * 32), 0, (unsigned)W * 64u, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b64(o, rt, mine ? tcol + 16 * g : 0x80000000u, 0, 0);
            }
        }
    };
    // Piece P (0 .. 22) of the epilogue of the last layer's row: the exchange between the lane halves one swap a
    // piece (acc_rows_interleaved), the activation pipeline behind it in the order the swaps release the values, the
    // two 16-byte stores at the end.
    auto last_piece = [&](auto P_, f32x16& a, float (&tmp)[16], float ia, int row) {
        constexpr int P = decltype(P_)::value;
        if constexpr (P == 0) mfma_result_wait(a);
        if constexpr (P < 8) {
            constexpr int g = P >> 2, e = P & 3;
            float lo = a[8 * g + e], hi = a[8 * g + 4 + e];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
            a[8 * g + e] = lo;
            a[8 * g + 4 + e] = hi;
        }
        cstatic_for<0, 6>([&](auto k_) {
            constexpr int k = decltype(k_)::value, n = P - 1 - k;
            if constexpr (n >= 0 && n < 16) {
                // value n in release order: swap n / 2 frees registers 8g + e and 8g + 4 + e (g = n / 8, e = (n / 2) & 3)
                constexpr int v = 8 * (n >> 3) + ((n >> 1) & 3) + 4 * (n & 1);
                const float nv = act_stage(k, a[v], tmp[v], ia);
                if constexpr (k < 5) tmp[v] = nv;
                else a[v] = nv;
            }
        });
        if constexpr (P == NP_LAST - 1) {
            // no branch around the stores either
            const bool mine = row >= r0 && row < r1;
            unsigned o[8];
#pragma unroll
            for (int i = 0; i < 16; i += 2) o[i >> 1] = bf_pack(a[i], a[i + 1]);
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(yimg + (long)(mine ? row : 0) * W * 32), 0, (unsigned)W * 64u, 0x00020000);
            const unsigned oc = mine ? ocol : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[0], o[1], o[2], o[3]}, ry, oc, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){o[4], o[5], o[6], o[7]}, ry, oc + 32, 0, 0);
        }
    };
    // One phase: the KK x KK x 2 MFMAs of a layer row, `bat` holding the B fragments of its first tap row on entry and
    // those of the next phase's first tap row (row `nrow`, KKN taps) on exit; `pieces(P)` for P in [0, NP) is spread
    // evenly over the MFMA slots.  The scheduling barriers pin the order.
    auto phase = [&](auto kk_, auto kkn_, auto np_, const auto& wf, const unsigned char* ra, const unsigned char* rb,
                     const unsigned char* rc, const unsigned char* nrow, u32x4 (&bat)[6], f32x16& acc, auto&& pieces) {
        constexpr int KK = decltype(kk_)::value, KKN = decltype(kkn_)::value, NP = decltype(np_)::value;
        constexpr int NB = KK * 2, NBN = KKN * 2, M = KK * NB;
        u32x4 cur[6], nxt[6];
#pragma unroll
        for (int i = 0; i < NB; ++i) cur[i] = bat[i];
        cstatic_for<0, M>([&](auto m_) {
            constexpr int m = decltype(m_)::value, a = m / NB, i = m - a * NB;
            if constexpr (a + 1 < KK) {
                const unsigned char* row = (a + 1 == 1) ? rb : rc;
                nxt[i] = *(const u32x4*)(row + bo[i >> 1][i & 1]);
            } else {
                constexpr int l0 = (i * NBN + NB - 1) / NB, l1 = ((i + 1) * NBN + NB - 1) / NB;
                cstatic_for<l0, l1>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    bat[j] = *(const u32x4*)(nrow + bo[j >> 1][j & 1]);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_wacc(acc, wf[m], cur[i]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int q0_ = (m * NP) / M, q1_ = ((m + 1) * NP) / M;
            cstatic_for<q0_, q1_>(pieces);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i == NB - 1 && a + 1 < KK) {
#pragma unroll
                for (int j = 0; j < NB; ++j) cur[j] = nxt[j];
            }
        });
        (void)ra;
    };

    // ---- prologue ----
    u32x4 svb[3];
    row_load(s_start - 1, svb);
    row_write(i0, svb);
    row_load(s_start, svb);
    {
        // layer-1 row s_start - 1: zero padding in the first piece, feeds nothing that is stored in the others
        *(u32x4*)(p2 + cxoff(r, h)) = zero4;
        *(u32x4*)(p2 + cxoff(r, 2 + h)) = zero4;
    }
    row_write(i1, svb);
    row_load(s_start + 1, svb);
    f32x16 acc1, acc2, acc3;
    float tm1[16], tm2[16], tm3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; tm1[i] = 0.f; tm2[i] = 0.f; tm3[i] = 0.f; }
    u32x4 bat[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) bat[i] = *(const u32x4*)(i0 + bo[i >> 1][i & 1]);
    int rowl = -(1 << 20);           // row of the last layer's accumulator waiting for its epilogue (none yet)
    int row2 = -(1 << 20);           // three layers: row of layer 2 whose ring epilogue is half done
    using C3 = std::integral_constant<int, 3>;
    using CK1 = std::integral_constant<int, K1>;
    using CKL = std::integral_constant<int, KLAST>;

    for (int s = s_start; s <= s_last; ++s) {
        row_write(i2, svb);
        row_load(s + 2, svb);
        const bool ok1 = s < H;
        // ---- phase A: layer 1, row s ----
        acc_init(acc1, bl);
        if constexpr (NL == 3) {
            const bool ok2 = (unsigned)row2 < (unsigned)H;
            phase(C3{}, CK1{}, std::integral_constant<int, NP_LAST + NP_RING - 10>{}, wf0, i0, i1, i2, p0, bat, acc1,
                  [&](auto P_) {
                      constexpr int P = decltype(P_)::value;
                      // the two epilogues alternate piece by piece while both last
                      if constexpr (P < 2 * (NP_RING - 10)) {
                          if constexpr (P & 1) ring_piece(std::integral_constant<int, 10 + (P >> 1)>{}, acc2, tm2, ia1, q1, in2, ok2, t1, row2, tcol2);
                          else last_piece(std::integral_constant<int, (P >> 1)>{}, acc3, tm3, ia2, rowl);
                      } else {
                          last_piece(std::integral_constant<int, P - (NP_RING - 10)>{}, acc3, tm3, ia2, rowl);
                      }
                  });
        } else {
            phase(C3{}, CK1{}, std::integral_constant<int, NP_LAST>{}, wf0, i0, i1, i2, K1 == 3 ? p0 : p1, bat, acc1,
                  [&](auto P_) { last_piece(P_, acc2, tm2, ia1, rowl); });
        }
        // ---- phase B: layer 2, row s - 2 ----
        acc_init(acc2, bl + 32);
        if constexpr (NL == 3) {
            phase(CK1{}, CKL{}, std::integral_constant<int, NP_RING>{}, wf1, p0, p1, p2, q0, bat, acc2,
                  [&](auto P_) { ring_piece(P_, acc1, tm1, ia0, p3, in1, ok1, t0, s, tcol1); });
            row2 = s - 2;
            // ---- phase C: layer 3, row s - 4 ----
            acc_init(acc3, bl + 64);
            const bool ok2 = (unsigned)row2 < (unsigned)H;
            phase(CKL{}, C3{}, std::integral_constant<int, 10>{}, wf2, q0, q1, q1, i1, bat, acc3,
                  [&](auto P_) { ring_piece(P_, acc2, tm2, ia1, q2, in2, ok2, t1, row2, tcol2); });
            rowl = s - 4;
        } else {
            if constexpr (K1 == 3)
                phase(CK1{}, C3{}, std::integral_constant<int, NP_RING>{}, wf1, p0, p1, p2, i1, bat, acc2,
                      [&](auto P_) { ring_piece(P_, acc1, tm1, ia0, p3, in1, ok1, t0, s, tcol1); });
            else
                phase(CK1{}, C3{}, std::integral_constant<int, NP_RING>{}, wf1, p1, p2, p2, i1, bat, acc2,
                      [&](auto P_) { ring_piece(P_, acc1, tm1, ia0, p3, in1, ok1, t0, s, tcol1); });
            rowl = s - 2;
        }
        // rotate the rings
        unsigned char* t = i0; i0 = i1; i1 = i2; i2 = t;
        t = p0; p0 = p1; p1 = p2; p2 = p3; p3 = t;
        t = q0; q0 = q1; q1 = q2; q2 = t;
    }
    // the epilogue of the last row
    if constexpr (NL == 3) cstatic_for<0, NP_LAST>([&](auto P_) { last_piece(P_, acc3, tm3, ia2, rowl); });
    else cstatic_for<0, NP_LAST>([&](auto P_) { last_piece(P_, acc2, tm2, ia1, rowl); });
}

template <int NL, int KLAST, int WAVES, bool TAPE>
int launch_chain(const ChainDesc& d, const bf16_t* x, const float* const* w, const float* const* b, bf16_t* y,
                 bf16_t* t0, bf16_t* t1, hipStream_t st) {
    const long waves = (long)d.nimg * d.strips * d.nseg;
    const long blocks = (waves + WAVES - 1) / WAVES;
    if (blocks <= 0 || blocks >= (1L << 31)) return MGX_ERR_ARG;
    constexpr int lds = chain_lds<NL, KLAST, WAVES>();
    static_assert(lds <= 160 * 1024, "the chain kernel's LDS image exceeds a CU");
    if (hipFuncSetAttribute((const void*)bchain_kernel<NL, KLAST, WAVES, TAPE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            lds) != hipSuccess)
        return MGX_ERR_LAUNCH;
    hipLaunchKernelGGL((bchain_kernel<NL, KLAST, WAVES, TAPE>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, d, x,
                       w[0], b[0], w[1], b[1], NL == 3 ? w[2] : nullptr, NL == 3 ? b[2] : nullptr, y, t0, t1);
    return mgx_launch_status();
}

template <int NL, int KLAST, bool TAPE>
int launch_chainp(const ChainDesc& d, const bf16_t* x, const float* const* w, const float* const* b, bf16_t* y,
                  bf16_t* t0, bf16_t* t1, hipStream_t st) {
    const long waves = (long)d.nimg * d.strips * d.nseg;
    const long blocks = (waves + 3) / 4;
    if (blocks <= 0 || blocks >= (1L << 31)) return MGX_ERR_ARG;
    constexpr int lds = 4 * pchain_wave_lds<NL>();
    static_assert(lds <= 160 * 1024, "the chain kernel's LDS image exceeds a CU");
    if (hipFuncSetAttribute((const void*)bchainp_kernel<NL, KLAST, TAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
        return MGX_ERR_LAUNCH;
    hipLaunchKernelGGL((bchainp_kernel<NL, KLAST, TAPE>), dim3((unsigned)blocks), dim3(256), lds, st, d, x, w[0], b[0], w[1],
                       b[1], NL == 3 ? w[2] : nullptr, NL == 3 ? b[2] : nullptr, y, t0, t1);
    return mgx_launch_status();
}

}  // namespace

// Chains the kernel serves: kernel sizes {3,3,2}, {3,3} or {3,2}; byte offsets inside one image within 32 bits.
extern "C" int mgx_conv_chain_bf16_supported(int64_t nimg, int h, int w, int nlayers, const int* ksizes) {
    if (nimg <= 0 || h < 2 || w < 8 || !ksizes) return 0;
    if ((long)h * w * 64 >= (1L << 32) - 64) return 0;
    if (nlayers == 3) return ksizes[0] == 3 && ksizes[1] == 3 && ksizes[2] == 2;
    if (nlayers == 2) return ksizes[0] == 3 && (ksizes[1] == 3 || ksizes[1] == 2);
    return 0;
}

// Pieces per strip: the count (1 .. 8, pieces of 16 rows or more) for which the launch's workgroups, one per CU at a
// time on the 256 CUs of an MI355X, take the fewest row steps: rounds of workgroups x (rows of a piece + the rows a
// piece computes in front of its first output row and behind its last: `extra_rows` steps).
static void chain_pieces(ChainDesc& d, int extra_rows, int waves_per_group) {
    long best = -1;
    for (int n = 1; n <= 8; ++n) {
        const int rows = (d.h + n - 1) / n;
        if (n > 1 && rows < 16) break;
        const int used = (d.h + rows - 1) / rows;            // pieces that hold any row
        const long groups = ((long)d.nimg * d.strips * used + waves_per_group - 1) / waves_per_group;
        const long cost = ((groups + 255) / 256) * (rows + extra_rows);
        if (best < 0 || cost < best) {
            best = cost;
            d.nseg = used;
            d.seg_rows = rows;
        }
    }
}

// ys[i] = output of layer i (bf16, shape of x); with `taped` every one is written, else only the last
static int chain_run(int64_t nimg, int h, int w, int nlayers, const int* ksizes, const void* x,
                     const float* const* weights, const float* const* biases, const float* alphas, void* const* ys,
                     bool taped, void* stream) {
    if (!x || !ys || !weights || !biases || !alphas || !mgx_conv_chain_bf16_supported(nimg, h, w, nlayers, ksizes))
        return MGX_ERR_ARG;
    ChainDesc d;
    d.nimg = (int)nimg; d.h = h; d.w = w;
    for (int i = 0; i < nlayers; ++i) {
        if (!weights[i] || !biases[i] || !(alphas[i] > 0.f) || ((taped || i == nlayers - 1) && !ys[i])) return MGX_ERR_ARG;
        d.alpha[i] = alphas[i];
    }
    if (nlayers == 2) d.alpha[2] = 1.f;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xb = (const bf16_t*)x;
    bf16_t* yb = (bf16_t*)ys[nlayers - 1];
    bf16_t* t0 = taped ? (bf16_t*)ys[0] : nullptr;
    bf16_t* t1 = (taped && nlayers == 3) ? (bf16_t*)ys[1] : nullptr;
    // MGX_CHAIN_PIPELINED=1: bchainp_kernel (one wave a SIMD, a layer row's MFMAs interleaved with another row's
    // epilogue) instead of bchain_kernel (two waves a SIMD, MFMAs and epilogue one after the other).  Measured at
    // 4 x 256 x 48000: {3x3, 3x3, 2x2} 3.50 against 3.37 ms, {3x3, 3x3} 2.73 against 2.62, {3x3, 2x2} 1.97 against 2.25
    // (the only variant whose registers and LDS let two workgroups share a CU): not the default.
    const char* env = getenv("MGX_CHAIN_PIPELINED");
    const bool serial = !(env && env[0] == '1');
    if (nlayers == 3) {
        d.strips = (w + chain_cw<3, 2>() - 1) / chain_cw<3, 2>();
        if (serial) {
            chain_pieces(d, 4, 8);
            return taped ? launch_chain<3, 2, 8, true>(d, xb, weights, biases, yb, t0, t1, st)
                         : launch_chain<3, 2, 8, false>(d, xb, weights, biases, yb, t0, t1, st);
        }
        chain_pieces(d, 6, 4);
        return taped ? launch_chainp<3, 2, true>(d, xb, weights, biases, yb, t0, t1, st)
                     : launch_chainp<3, 2, false>(d, xb, weights, biases, yb, t0, t1, st);
    }
    if (ksizes[1] == 3) {
        d.strips = (w + chain_cw<2, 3>() - 1) / chain_cw<2, 3>();
        if (serial) {
            chain_pieces(d, 3, 8);
            return taped ? launch_chain<2, 3, 8, true>(d, xb, weights, biases, yb, t0, t1, st)
                         : launch_chain<2, 3, 8, false>(d, xb, weights, biases, yb, t0, t1, st);
        }
        chain_pieces(d, 4, 4);
        return taped ? launch_chainp<2, 3, true>(d, xb, weights, biases, yb, t0, t1, st)
                     : launch_chainp<2, 3, false>(d, xb, weights, biases, yb, t0, t1, st);
    }
    d.strips = (w + chain_cw<2, 2>() - 1) / chain_cw<2, 2>();
    if (serial) {
        chain_pieces(d, 3, 8);
        return taped ? launch_chain<2, 2, 8, true>(d, xb, weights, biases, yb, t0, t1, st)
                     : launch_chain<2, 2, 8, false>(d, xb, weights, biases, yb, t0, t1, st);
    }
    chain_pieces(d, 4, 4);
    return taped ? launch_chainp<2, 2, true>(d, xb, weights, biases, yb, t0, t1, st)
                 : launch_chainp<2, 2, false>(d, xb, weights, biases, yb, t0, t1, st);
}

extern "C" int mgx_conv_chain_bf16(int64_t nimg, int h, int w, int nlayers, const int* ksizes, const void* x,
                                   const float* const* weights, const float* const* biases, const float* alphas,
                                   void* y, void* stream) {
    if (nlayers < 2 || nlayers > 3) return MGX_ERR_ARG;
    void* ys[3] = {nullptr, nullptr, nullptr};
    ys[nlayers - 1] = y;
    return chain_run(nimg, h, w, nlayers, ksizes, x, weights, biases, alphas, ys, false, stream);
}

extern "C" int mgx_conv_chain_bf16_taped(int64_t nimg, int h, int w, int nlayers, const int* ksizes, const void* x,
                                         const float* const* weights, const float* const* biases, const float* alphas,
                                         void* const* ys, void* stream) {
    if (nlayers < 2 || nlayers > 3) return MGX_ERR_ARG;
    return chain_run(nimg, h, w, nlayers, ksizes, x, weights, biases, alphas, ys, true, stream);
}
