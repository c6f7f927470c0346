// Fused forward of one generator grow block (add_generator_block, trainer/models.py:358-399) on
// v_mfma_f32_32x32x16_bf16: the layer chain of gblock.hip with bf16 operands and float32 accumulation.
//
// As in the float32 kernel a wave owns 32 low-resolution positions and every layer is computed
// transposed, D[channel][position] = W^T X, so the accumulator tile of one layer becomes the B operand of
// the next after one in-register conversion: registers 8s .. 8s+7 of a tile, converted pairwise to bf16,
// are k-step s of the next MFMA (contraction index of element j in lane half h = channel
// chan(8s + j, h); the weight fragments are packed in that order).  Activations never leave registers.
//
// What changes at 16 x the matrix rate is the weight stream: one 1 KiB fragment per 32-cycle MFMA and
// wave is more than the L2 can feed to four waves per CU, so the four waves of a workgroup share the
// fragments through LDS.  The block's weights (303 fragments, 303 KiB) are cut into 8 phases per output
// sub-row: six branches (33 fragments each) and two halves of the Dense tail (48 + 53); while the waves
// compute phase p from one LDS buffer, the fragments of phase p+1 arrive in the other by LDS-DMA
// (global_load_lds_dwordx4: fragments are lane-linear, 16 bytes a lane, so the DMA's lane-linear
// destination is exactly the layout ds_read_b128 wants); one wait + barrier per phase.  Workgroups are
// persistent (one per CU) and walk tiles of 128 positions, so the stream never restarts.
#include "bf16.h"
#include "gblock_frag.h"
#include "../../include/mgx.h"

namespace {

// bf16 fragment numbering (1 KiB each): [64 lanes][8 bf16]
constexpr int FB_F = 0;                  // featured conv: [it 4]; element j = (tap j>>1, channel j&1) in both lane halves
constexpr int FB_BR = 4;                 // 6 branches x 33
constexpr int BRF = 33;
constexpr int FB_K1 = 0;                 //   [it 4][ks 2]      rows (a, co)
constexpr int FB_D1 = 8;                 //   1
constexpr int FB_K2 = 9;                 //   [b 2][ot 2][ks 2]
constexpr int FB_D2 = 17;                //   [ot 4][it 2][ks 2]
constexpr int FB_T = FB_BR + 6 * BRF;    // 202: tail, per chunk c of Dense(256): W1 [it 4][ks 2] then W2 [ot 2][ks 2]
constexpr int TCH = 12;
constexpr int FB_W3 = FB_T + 8 * TCH;    // 298: [it 2][ks 2]
constexpr int FB_W4 = FB_W3 + 4;         // 302
constexpr int NFB = FB_W4 + 1;           // 303
constexpr int TAIL_LO = 4 * TCH;         // 48 fragments: chunks 0-3
constexpr int TAIL_HI = 4 * TCH + 5;     // 53 fragments: chunks 4-7, W3, W4

// float32 biases behind the fragments
constexpr int BO_SUM = 0;                // sum of the six Dense(128) biases (128)
constexpr int BO_BR = 128;               // per branch: b1 (16) | c1 (32) | b2 (64)
constexpr int BBR = 112;
constexpr int BO_B0 = BO_BR + 6 * BBR;   // featured bias (128)
constexpr int BO_E1 = BO_B0 + 128;       // tail biases 256 | 64 | 8 | 8
constexpr int BO_E2 = BO_E1 + 256;
constexpr int BO_E3 = BO_E2 + 64;
constexpr int BO_E4 = BO_E3 + 8;
constexpr int NBIAS = BO_E4 + 8;
constexpr long PACKB_BYTES = (long)NFB * 1024 + NBIAS * 4;

constexpr int BUF_BYTES = TAIL_HI * 1024;           // one phase buffer
constexpr int LDS_BYTES = 4 * 1024 + 2 * BUF_BYTES; // featured fragments + two phase buffers

__global__ __launch_bounds__(256) void gblock_pack_bf16_kernel(mgx_gblock_weights w, unsigned char* __restrict__ packb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float* bias = (float*)(packb + (long)NFB * 1024);
    if (idx < NBIAS) {
        float s = 0.f;
        if (idx < BO_BR) {
            for (int k = 0; k < 6; ++k) s += w.c2[k][idx];
        } else if (idx < BO_B0) {
            const int k = (idx - BO_BR) / BBR, e = (idx - BO_BR) % BBR;
            s = e < 16 ? w.b1[k][e] : (e < 48 ? w.c1[k][e - 16] : w.b2[k][e - 48]);
        } else if (idx < BO_E1) s = w.b0[idx - BO_B0];
        else if (idx < BO_E2) s = w.e1[idx - BO_E1];
        else if (idx < BO_E3) s = w.e2[idx - BO_E2];
        else if (idx < BO_E4) s = w.e3[idx - BO_E3];
        else s = (idx - BO_E4) < 2 ? w.e4[idx - BO_E4] : 0.f;
        bias[idx] = s;
    }
    if (idx >= NFB * 512) return;
    // one thread per (fragment, lane, element)
    const int f = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
    const int i = lane & 31, h = lane >> 5;
    float v = 0.f;
    if (f < FB_BR) {                                     // featured: K0 (2,2,2,128): [tap][c][n]
        v = w.k0[j * 128 + f * 32 + i];                  // j = tap*2 + c; both k halves (x is fed as hi + lo parts)
    } else if (f < FB_T) {
        const int k = (f - FB_BR) / BRF, g = (f - FB_BR) % BRF;
        if (g < FB_D1) {                                 // K1 (2,1,16,128): [a][co][ci], rows i = 16a + co
            const int it = g >> 1, ks = g & 1;
            v = w.k1[k][i * 128 + it * 32 + chan(8 * ks + j, h)];
        } else if (g < FB_K2) {                          // D1 (16,32): [in][out]
            v = w.d1[k][chan(j, h) * 32 + i];
        } else if (g < FB_D2) {                          // K2 (1,2,64,32): [b][co][ci]
            const int q = g - FB_K2, b = q >> 2, ot = (q >> 1) & 1, ks = q & 1;
            v = w.k2[k][(b * 64 + ot * 32 + i) * 32 + chan(8 * ks + j, h)];
        } else {                                         // D2 (64,128): [in][out]
            const int q = g - FB_D2, ot = q >> 2, it = (q >> 1) & 1, ks = q & 1;
            v = w.d2[k][(it * 32 + chan(8 * ks + j, h)) * 128 + ot * 32 + i];
        }
    } else if (f < FB_W3) {
        const int c = (f - FB_T) / TCH, g = (f - FB_T) % TCH;
        if (g < 8) {                                     // W1 (128,256)
            const int it = g >> 1, ks = g & 1;
            v = w.w1[(it * 32 + chan(8 * ks + j, h)) * 256 + c * 32 + i];
        } else {                                         // W2 (256,64)
            const int q = g - 8, ot = q >> 1, ks = q & 1;
            v = w.w2[(c * 32 + chan(8 * ks + j, h)) * 64 + ot * 32 + i];
        }
    } else if (f < FB_W4) {                              // W3 (64,8): rows >= 8 are zero
        const int q = f - FB_W3, it = q >> 1, ks = q & 1;
        v = i < 8 ? w.w3[(it * 32 + chan(8 * ks + j, h)) * 8 + i] : 0.f;
    } else {                                             // W4 (8,2): rows >= 2 and inputs >= 8 are zero
        v = (i < 2 && j < 4) ? w.w4[chan(j, h) * 2 + i] : 0.f;
    }
    *(bf16_t*)(packb + ((long)(f * 64 + lane) * 8 + j) * 2) = (bf16_t)v;
}

// accumulator tile initialised with bias rows chan(reg, h): registers 4g..4g+3 are rows 8g+4h.. (one 16-byte load)
__device__ __forceinline__ void init_bias(f32x16& acc, const float* __restrict__ bias, int h, int mask, int ngroups) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g < ngroups) v = *(const f32x4*)(bias + ((8 * g + 4 * h) & mask));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
    }
}

// B operand of the next layer: k-step s = registers 8s .. 8s+7 as bf16
struct BTile { u32x4 k[2]; };
__device__ __forceinline__ BTile to_b(const f32x16& a) {
    BTile t;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) t.k[s][q] = bf_pack(a[8 * s + 2 * q], a[8 * s + 2 * q + 1]);
    return t;
}

__device__ __forceinline__ u32x4 frag(const unsigned char* buf, int f, int lane) {
    return *(const u32x4*)(buf + f * 1024 + lane * 16);
}

// fragments [f0, f0 + n) of the packed weights -> LDS at dst, one fragment per wave-instruction
__device__ __forceinline__ void stage(const unsigned char* __restrict__ packb, int f0, int n, unsigned char* dst, int wv,
                                      int lane) {
    for (int f = wv; f < n; f += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(packb + (long)(f0 + f) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(256, 1) void gblock_fwd_bf16_kernel(const unsigned char* __restrict__ packb,
                                                                 const float* __restrict__ x, float* __restrict__ y,
                                                                 long npix, int hl, int wl, long ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const float* bias = (const float*)(packb + (long)NFB * 1024);
    unsigned char* fbuf = lds;                               // featured fragments (4)
    unsigned char* pbuf = lds + 4 * 1024;                    // two phase buffers

    // phase q of a sub-row: fragments [phase_f0(q), + phase_n(q))
    auto phase_f0 = [](int q) { return q < 6 ? FB_BR + q * BRF : (q == 6 ? FB_T : FB_T + TAIL_LO); };
    auto phase_n = [](int q) { return q < 6 ? BRF : (q == 6 ? TAIL_LO : TAIL_HI); };

    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    stage(packb, FB_F, 4, fbuf, wv, lane);
    stage(packb, phase_f0(0), phase_n(0), pbuf, wv, lane);
    __syncthreads();
    int par = 0;                                             // buffer of the current phase
    for (; tile < ntiles; tile += gridDim.x) {
        const long p = tile * 128 + wv * 32 + r;
        const bool pvalid = p < npix;
        const long pc = pvalid ? p : 0;
        const float2 xv = pvalid ? *(const float2*)(x + pc * 2) : make_float2(0.f, 0.f);
        const long row = pc / wl;
        const int jcol = (int)(pc - row * wl);
        float* yrow0 = y + ((row * 2) * (2L * wl) + 2L * jcol) * 2;
        // B operand of the featured conv: element j = x[channel j & 1] for each of the 4 taps.  Only 8 of the
        // 16 contraction slots are needed, so the block input goes in as two bf16 parts, x = hi + lo (lane
        // half 0 carries hi, half 1 the remainder): the 2-channel input keeps 16 significant bits.
        const unsigned xhi = bf_pack(xv.x, xv.y);
        const unsigned xlo = bf_pack(xv.x - bf_lo(xhi), xv.y - bf_hi(xhi));
        const unsigned xp = h == 0 ? xhi : xlo;
        const u32x4 xb = (u32x4){xp, xp, xp, xp};
        const bool more_tiles = tile + gridDim.x < ntiles;
#pragma unroll 1
        for (int A = 0; A < 2; ++A) {
            float* yrow = yrow0 + A * (2L * wl) * 2;
            BTile Fb[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x16 F;
                init_bias(F, bias + BO_B0 + it * 32, h, 127, 4);
                F = mfma_bf16(frag(fbuf, it, lane), xb, F);
                Fb[it] = to_b(F);
            }
            f32x16 S[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) init_bias(S[b][ot], bias + BO_SUM + ot * 32, h, 127, 4);
#pragma unroll 1
            for (int k = 0; k < 6; ++k) {
                stage(packb, phase_f0(k + 1), phase_n(k + 1), pbuf + (par ^ 1) * BUF_BYTES, wv, lane);
                const unsigned char* wb = pbuf + par * BUF_BYTES;
                const float* bb = bias + BO_BR + k * BBR;
                f32x16 o1;
                init_bias(o1, bb, h, 15, 4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) o1 = mfma_bf16(frag(wb, FB_K1 + it * 2 + ks, lane), Fb[it].k[ks], o1);
                const BTile o1b = to_b(o1);
                f32x16 o2;
                init_bias(o2, bb + 16, h, 31, 4);
                o2 = mfma_bf16(frag(wb, FB_D1, lane), A ? o1b.k[1] : o1b.k[0], o2);      // the 16 channels of sub-row A
                const BTile o2b = to_b(o2);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    BTile o3b[2];
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot) {
                        f32x16 o3;
                        init_bias(o3, bb + 48 + ot * 32, h, 63, 4);
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            o3 = mfma_bf16(frag(wb, FB_K2 + (b * 2 + ot) * 2 + ks, lane), o2b.k[ks], o3);
                        o3b[ot] = to_b(o3);
                    }
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                        for (int it = 0; it < 2; ++it)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                S[b][ot] = mfma_bf16(frag(wb, FB_D2 + (ot * 2 + it) * 2 + ks, lane), o3b[it].k[ks], S[b][ot]);
                }
                __syncthreads();                      // waits for this wave's DMA, then for every wave
                par ^= 1;
            }
            // ---- Dense tail: Dense(256) in 8 chunks of 32 rows, each consumed at once by Dense(64) ----
            BTile Sb[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int it = 0; it < 4; ++it) Sb[b][it] = to_b(S[b][it]);
            f32x16 T2[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) init_bias(T2[b][ot], bias + BO_E2 + ot * 32, h, 63, 4);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                // next phase: the second tail half, then branch 0 of the next sub-row / tile
                if (half == 0) stage(packb, phase_f0(7), phase_n(7), pbuf + (par ^ 1) * BUF_BYTES, wv, lane);
                else if (A == 0 || more_tiles) stage(packb, phase_f0(0), phase_n(0), pbuf + (par ^ 1) * BUF_BYTES, wv, lane);
                const unsigned char* wb = pbuf + par * BUF_BYTES;
#pragma unroll 1
                for (int cc = 0; cc < 4; ++cc) {
                    const unsigned char* wc = wb + cc * TCH * 1024;
                    const int c = half * 4 + cc;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x16 T1;
                        init_bias(T1, bias + BO_E1 + c * 32, h, 255, 4);
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) T1 = mfma_bf16(frag(wc, it * 2 + ks, lane), Sb[b][it].k[ks], T1);
                        const BTile T1b = to_b(T1);
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                T2[b][ot] = mfma_bf16(frag(wc, 8 + ot * 2 + ks, lane), T1b.k[ks], T2[b][ot]);
                    }
                }
                if (half == 1) {
                    const unsigned char* w3 = wb + 4 * TCH * 1024;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const BTile t0 = to_b(T2[b][0]), t1 = to_b(T2[b][1]);
                        f32x16 T3, O;
                        init_bias(T3, bias + BO_E3, h, 7, 1);
                        T3 = mfma_bf16(frag(w3, 0, lane), t0.k[0], T3);
                        T3 = mfma_bf16(frag(w3, 1, lane), t0.k[1], T3);
                        T3 = mfma_bf16(frag(w3, 2, lane), t1.k[0], T3);
                        T3 = mfma_bf16(frag(w3, 3, lane), t1.k[1], T3);
                        const BTile t3 = to_b(T3);
                        init_bias(O, bias + BO_E4, h, 7, 1);
                        O = mfma_bf16(frag(w3, 4, lane), t3.k[0], O);
                        // rows 0,1 (the two output channels) live in registers 0,1 of lane half 0
                        if (pvalid && h == 0) *(float2*)(yrow + b * 2) = make_float2(O[0], O[1]);
                    }
                }
                __syncthreads();
                par ^= 1;
            }
        }
    }
}

}  // namespace

extern "C" int64_t mgx_gblock_pack_bf16_bytes(void) { return PACKB_BYTES; }

extern "C" int mgx_gblock_fwd_bf16(const mgx_gblock_weights* wp, void* packb, const float* x, float* y, int64_t nimg,
                                   int hl, int wl, void* stream) {
    if (!wp || !packb || !x || !y || nimg <= 0 || hl <= 0 || wl <= 0) return MGX_ERR_ARG;
    const mgx_gblock_weights w = *wp;
    const void* const* ptrs = (const void* const*)&w;
    for (size_t i = 0; i < sizeof(w) / sizeof(void*); ++i)
        if (!ptrs[i]) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long npix = (long)nimg * hl * wl;
    const long ntiles = (npix + 127) / 128;
    hipLaunchKernelGGL(gblock_pack_bf16_kernel, dim3((NFB * 512 + 255) / 256), dim3(256), 0, st, w, (unsigned char*)packb);
    if (hipFuncSetAttribute((const void*)gblock_fwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess)
        return MGX_ERR_LAUNCH;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    hipLaunchKernelGGL(gblock_fwd_bf16_kernel, dim3(grid), dim3(256), LDS_BYTES, st, (const unsigned char*)packb, x, y,
                       npix, hl, wl, ntiles);
    return mgx_launch_status();
}
