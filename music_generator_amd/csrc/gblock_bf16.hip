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
#include "gblock_rows.h"
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

constexpr int TL_BYTES = 2048;                       // [32 positions][32 rows] bf16 transposition image; two per wave
constexpr int BUF_BYTES = TAIL_HI * 1024;           // one phase buffer
constexpr int LDS_BIAS = ((NBIAS * 4 + 1023) / 1024) * 1024;      // the float32 biases, copied once per workgroup
constexpr int LDS_BYTES = 4 * 1024 + 2 * BUF_BYTES + 8 * TL_BYTES + LDS_BIAS; // featured fragments + two phase buffers + the
                                                                // waves' transposition images of the tape stores

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

// MFMA into an accumulator tile pinned to the accumulation registers (the file is otherwise built with MFMA results
// in vector registers, where the conversions read them): for the eight tiles that sum the six branches, which are
// only ever touched by MFMAs until the sub-row's tail, so that 128 vector registers stay free for the operand
// reads to run ahead.  The compiler does not see this instruction: mfma_result_wait() before the tile is read.
__device__ __forceinline__ void mfma_bf16_pinned(f32x16& c, u32x4 a, u32x4 b) {
    // s_nop 1: the two wait states between a vector-ALU write of an operand (the conversion that produced `b`) and the
    // MFMA that reads it; the compiler pads them for MFMAs it emits itself, never inside an asm statement
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
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

// Rows chan(8s + j, h) (s < NS k-steps) of a converted tile -> 32-position bf16 rows of a tile-major side
// buffer (row map: gblock_rows.h).  Element pairs of a dword are two consecutive rows of one position.
// the two bf16 halves of a dword -> rows `row` and `row + 1` of position r
__device__ __forceinline__ void store_pair(bf16_t* __restrict__ buf, int row, int r, unsigned u) {
    unsigned short* p = (unsigned short*)buf;
    p[row * 32 + r] = (unsigned short)(u & 0xffffu);
    p[(row + 1) * 32 + r] = (unsigned short)(u >> 16);
}

// The 32 x 32 tile goes through a wave-private LDS image [position][row] (64 B a position, the eight 8-byte
// chunks of a position XOR-swizzled with bits 1-3 of the position: conflict-free for the writes and for the
// transposing reads) and comes back with ds_read_b64_tr_b16 as 8 consecutive positions of one row per lane:
// one 16-byte store per lane and 16 rows, each store instruction 1 KiB of consecutive rows.  (Stored straight
// from the B-tile registers a lane holds one position of a row: 2-byte stores, 16 instructions a tile, which
// cost more than the tile's MFMAs.)
__device__ __forceinline__ int tl_off(int pos, int chunk) { return pos * 64 + ((chunk ^ ((pos >> 1) & 7)) << 3); }

// The image is double-buffered and the way out of a tile is deferred to the next call: a call first issues the
// transposing reads of the PREVIOUS tile (written one call ago, nothing to wait for), then writes its own tile
// into the other image, then stores what it read; tape_flush() stores the last tile of a sequence.
struct TapeOut {
    unsigned char* tl;          // the wave's two images
    bf16_t* buf;                // pending tile: side buffer, first row, k-steps
    int row0, ns, par;
};

__device__ __forceinline__ void tape_read(const TapeOut& o, int lane, u32x2 (&lo)[2], u32x2 (&hi)[2]) {
    const int g16 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* img = o.tl + (o.par ^ 1) * TL_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u)                       // rows 16u .. 16u+15, positions 8 g16 .. 8 g16 + 7
        if (u < o.ns) {
            lo[u] = tr_read(img + tl_off(8 * g16 + q, 4 * u + p));
            hi[u] = tr_read(img + tl_off(8 * g16 + 4 + q, 4 * u + p));
        }
}

__device__ __forceinline__ void tape_store(const TapeOut& o, int lane, const u32x2 (&lo)[2], const u32x2 (&hi)[2]) {
    const int g16 = lane >> 4, i = lane & 15;
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (u < o.ns)
            *(u32x4*)((unsigned char*)o.buf + ((o.row0 + 16 * u + i) * 32 + 8 * g16) * 2) =
                (u32x4){lo[u][0], lo[u][1], hi[u][0], hi[u][1]};
}

template <int NS>
__device__ __forceinline__ void save_b(bf16_t* __restrict__ buf, int row0, const BTile& t, int r, int h, TapeOut& o) {
    const int lane = r + 32 * h;
    u32x2 lo[2], hi[2];
    tape_read(o, lane, lo, hi);
    // dwords 0-1 of k-step s are rows 16s + 4h + 0..3 (chunk 4s + h), dwords 2-3 rows 16s + 8 + 4h + 0..3
    unsigned char* img = o.tl + o.par * TL_BYTES;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        *(u32x2*)(img + tl_off(r, 4 * s + h)) = (u32x2){t.k[s][0], t.k[s][1]};
        *(u32x2*)(img + tl_off(r, 4 * s + 2 + h)) = (u32x2){t.k[s][2], t.k[s][3]};
    }
    tape_store(o, lane, lo, hi);
    o.buf = buf;
    o.row0 = row0;
    o.ns = NS;
    o.par ^= 1;
}

__device__ __forceinline__ void tape_flush(TapeOut& o, int r, int h) {
    const int lane = r + 32 * h;
    u32x2 lo[2], hi[2];
    tape_read(o, lane, lo, hi);
    tape_store(o, lane, lo, hi);
    o.ns = 0;
}

__device__ __forceinline__ u32x4 frag(const unsigned char* buf, int f, int lane) {
    return *(const u32x4*)(buf + f * 1024 + lane * 16);
}

// fragments [f0, f0 + n) of the packed weights -> LDS at dst, one fragment per wave-instruction
__device__ __forceinline__ void stage(const unsigned char* __restrict__ packb, int f0, int n, unsigned char* dst, int wv,
                                      int lane, int nw = 4) {
    for (int f = wv; f < n; f += nw)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(packb + (long)(f0 + f) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
}

// NW waves per workgroup share the staged weight fragments: 4 (one per SIMD), or 8 for the forward that keeps no tape
// (two per SIMD: the conversions of one wave run beside the MFMAs of the other)
template <bool SAVE, int NW, class RM>
__global__ __launch_bounds__(64 * NW, 1) void gblock_fwd_bf16_kernel(const unsigned char* __restrict__ packb,
                                                                 const float* __restrict__ x, float* __restrict__ y,
                                                                 bf16_t* __restrict__ act, long pos0, long npix, int hl,
                                                                 int wl, long ntiles) {
    GB_USE_ROWS(RM);
    constexpr bool lean = RM::lean;      // the weight gradients recompute what a lean tape leaves out (gblock_lean.hip)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    unsigned char* fbuf = lds;                               // featured fragments (4)
    unsigned char* pbuf = lds + 4 * 1024;                    // two phase buffers
    TapeOut tl = {lds + 4 * 1024 + 2 * BUF_BYTES + wv * 2 * TL_BYTES, nullptr, 0, 0, 0};   // this wave's images
    // biases in LDS: an accumulator tile starts from its bias, 26 times per sub-row; from global memory each
    // of those loads would stand in front of its MFMAs with an L2 latency (one wave per SIMD: nothing hides it)
    float* bias = (float*)(lds + 4 * 1024 + 2 * BUF_BYTES + 8 * TL_BYTES);
    {
        const float* gbias = (const float*)(packb + (long)NFB * 1024);
        for (int i = threadIdx.x; i < NBIAS; i += 64 * NW) bias[i] = gbias[i];
    }

    // phase q of a sub-row: fragments [phase_f0(q), + phase_n(q))
    auto phase_f0 = [](int q) { return q < 6 ? FB_BR + q * BRF : (q == 6 ? FB_T : FB_T + TAIL_LO); };
    auto phase_n = [](int q) { return q < 6 ? BRF : (q == 6 ? TAIL_LO : TAIL_HI); };

    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    stage(packb, FB_F, 4, fbuf, wv, lane, NW);
    stage(packb, phase_f0(0), phase_n(0), pbuf, wv, lane, NW);
    __syncthreads();
    int par = 0;                                             // buffer of the current phase
    for (; tile < ntiles; tile += gridDim.x) {
        const long p = pos0 + tile * (32 * NW) + wv * 32 + r;     // side-buffer tiles are counted from pos0
        const bool pvalid = p < npix;
        const long pc = pvalid ? p : 0;
        // positions past the end contribute activations of a zero input; their gradients are zero
        const float2 xv = pvalid ? *(const float2*)(x + pc * 2) : make_float2(0.f, 0.f);
        const long row = pc / wl;
        const int jcol = (int)(pc - row * wl);
        float* yrow0 = y ? y + ((row * 2) * (2L * wl) + 2L * jcol) * 2 : nullptr;
        bf16_t* at = SAVE ? act + (tile * NW + wv) * (GB_ROWS * 32L) : nullptr;
        // B operand of the featured conv: element j = x[channel j & 1] for each of the 4 taps.  Only 8 of the
        // 16 contraction slots are needed, so the block input goes in as two bf16 parts, x = hi + lo (lane
        // half 0 carries hi, half 1 the remainder): the 2-channel input keeps 16 significant bits.
        const unsigned xhi = bf_pack(xv.x, xv.y);
        const unsigned xlo = bf_pack(xv.x - bf_lo(xhi), xv.y - bf_hi(xhi));
        const unsigned xp = h == 0 ? xhi : xlo;
        const u32x4 xb = (u32x4){xp, xp, xp, xp};
        const bool more_tiles = tile + gridDim.x < ntiles;
        if (SAVE) {
            at[(GB_X + 0) * 32 + r] = (bf16_t)xv.x;      // both lane halves write the same two values
            at[(GB_X + 1) * 32 + r] = (bf16_t)xv.y;
        }
#pragma unroll 1
        for (int A = 0; A < 2; ++A) {
            float* yrow = yrow0 ? yrow0 + A * (2L * wl) * 2 : nullptr;
            BTile Fb[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x16 F;
                init_bias(F, bias + BO_B0 + it * 32, h, 127, 4);
                F = mfma_bf16(frag(fbuf, it, lane), xb, F);
                Fb[it] = to_b(F);
                if (SAVE && A == 0) save_b<2>(at, GB_F + it * 32, Fb[it], r, h, tl);
            }
            f32x16 S[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) init_bias(S[b][ot], bias + BO_SUM + ot * 32, h, 127, 4);
#pragma unroll 1
            for (int k = 0; k < 6; ++k) {
                stage(packb, phase_f0(k + 1), phase_n(k + 1), pbuf + (par ^ 1) * BUF_BYTES, wv, lane, NW);
                const unsigned char* wb = pbuf + par * BUF_BYTES;
                const float* bb = bias + BO_BR + k * BBR;
                f32x16 o1;
                init_bias(o1, bb, h, 15, 4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) o1 = mfma_bf16(frag(wb, FB_K1 + it * 2 + ks, lane), Fb[it].k[ks], o1);
                const BTile o1b = to_b(o1);
                bf16_t* abr = SAVE ? at + (GB_BR + k * GB_BR_ROWS) * 32 : nullptr;
                if (SAVE && A == 0) save_b<2>(abr, GB_O1, o1b, r, h, tl);
                f32x16 o2;
                init_bias(o2, bb + 16, h, 31, 4);
                o2 = mfma_bf16(frag(wb, FB_D1, lane), A ? o1b.k[1] : o1b.k[0], o2);      // the 16 channels of sub-row A
                const BTile o2b = to_b(o2);
                if (SAVE) save_b<2>(abr, GB_O2 + A * 32, o2b, r, h, tl);
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
                        if (SAVE && !lean) save_b<2>(abr, GB_O3 + (A * 2 + b) * 64 + ot * 32, o3b[ot], r, h, tl);
                    }
                    // the four output tiles innermost: an MFMA that continues the accumulator of the one before it
                    // issues ~5 % later than one on another accumulator (measured: 35.4 against 34.0 cycles); every
                    // accumulator still receives its four products in the same order
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int ot = 0; ot < 4; ++ot)
                                if (SAVE)       // the taped kernel is short of vector registers (tape stores): measured
                                    mfma_bf16_pinned(S[b][ot], frag(wb, FB_D2 + (ot * 2 + it) * 2 + ks, lane), o3b[it].k[ks]);
                                else            // 380 -> 482 TFLOP/s pinned; the plain forward loses with it (831 -> 813)
                                    S[b][ot] = mfma_bf16(frag(wb, FB_D2 + (ot * 2 + it) * 2 + ks, lane), o3b[it].k[ks], S[b][ot]);
                    // two waves per SIMD have 256 registers each: keep the two sub-columns from being interleaved
                    if (NW == 8) __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();                      // waits for this wave's DMA, then for every wave
                par ^= 1;
            }
            // ---- Dense tail: Dense(256) in 8 chunks of 32 rows, each consumed at once by Dense(64) ----
            BTile Sb[2][4];
            if (SAVE)        // tied to the eight tiles: the copies out of them must not be scheduled in front of the wait
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"
                             : "+a"(S[0][0]), "+a"(S[0][1]), "+a"(S[0][2]), "+a"(S[0][3]), "+a"(S[1][0]), "+a"(S[1][1]),
                               "+a"(S[1][2]), "+a"(S[1][3]));
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    Sb[b][it] = to_b(S[b][it]);
                    if (SAVE) save_b<2>(at, GB_S + (A * 2 + b) * 128 + it * 32, Sb[b][it], r, h, tl);
                }
            f32x16 T2[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) init_bias(T2[b][ot], bias + BO_E2 + ot * 32, h, 63, 4);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                // next phase: the second tail half, then branch 0 of the next sub-row / tile
                if (half == 0) stage(packb, phase_f0(7), phase_n(7), pbuf + (par ^ 1) * BUF_BYTES, wv, lane, NW);
                else if (A == 0 || more_tiles) stage(packb, phase_f0(0), phase_n(0), pbuf + (par ^ 1) * BUF_BYTES, wv, lane, NW);
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
                        if (SAVE && !lean) save_b<2>(at, GB_T1 + (A * 2 + b) * 256 + c * 32, T1b, r, h, tl);
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
                        if (SAVE) {
                            save_b<2>(at, GB_T2 + (A * 2 + b) * 64, t0, r, h, tl);
                            save_b<2>(at, GB_T2 + (A * 2 + b) * 64 + 32, t1, r, h, tl);
                        }
                        f32x16 T3, O;
                        init_bias(T3, bias + BO_E3, h, 7, 1);
                        T3 = mfma_bf16(frag(w3, 0, lane), t0.k[0], T3);
                        T3 = mfma_bf16(frag(w3, 1, lane), t0.k[1], T3);
                        T3 = mfma_bf16(frag(w3, 2, lane), t1.k[0], T3);
                        T3 = mfma_bf16(frag(w3, 3, lane), t1.k[1], T3);
                        const BTile t3 = to_b(T3);
                        if (SAVE) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) store_pair(at, GB_T3 + (A * 2 + b) * 8 + 4 * h + 2 * q, r, t3.k[0][q]);
                        }
                        init_bias(O, bias + BO_E4, h, 7, 1);
                        O = mfma_bf16(frag(w3, 4, lane), t3.k[0], O);
                        // rows 0,1 (the two output channels) live in registers 0,1 of lane half 0
                        if (pvalid && h == 0 && yrow) *(float2*)(yrow + b * 2) = make_float2(O[0], O[1]);
                    }
                }
                __syncthreads();
                par ^= 1;
            }
        }
    }
    if (SAVE) tape_flush(tl, r, h);
}

// ------------------------------------------------------------------------------------------
// Data-gradient chain dY -> dX of the block (the generator part of tape.gradient, models.py:144-146) on bf16
// MFMAs: the mirror of the forward kernel with transposed weight fragments (A operand of step (ks, j) for
// lane (i, h): W[input row i][output channel chan(8 ks + j, h)]).  Every intermediate gradient a weight
// gradient needs goes to the bf16 gradient side buffer `delta` (row map: gblock_rows.h).  Phases per
// sub-row a: two halves of the Dense tail (both sub-columns against the same staged weights), then the six
// branches; fragments reach LDS by LDS-DMA one phase ahead, as in the forward kernel.
// ------------------------------------------------------------------------------------------
constexpr int PT_K0 = 0;                 // [it 4][ks 2]: rows >= 2 zero, the four taps summed
constexpr int PT_TAIL = 8;               // W4T (1) | W3T [ot 2] | pad (1) | per chunk c: W2T [it 2][ks 2], W1T [ot 4][ks 2]
constexpr int PT_TAIL_LO = 4 + 4 * TCH;  // 52 fragments: W4T, W3T, chunks 0-3
constexpr int PT_TAIL_HI = 4 * TCH;      // 48 fragments: chunks 4-7
constexpr int PT_BR = PT_TAIL + PT_TAIL_LO + PT_TAIL_HI;   // 108: 6 branches x 34
constexpr int PBR = 34;
constexpr int PB_D2 = 0;                 //   [it 2][ot 4][ks 2]
constexpr int PB_K2 = 16;                //   [b 2][it 2][ks 2]
constexpr int PB_D1 = 24;                //   [ks 2], rows >= 16 zero
constexpr int PB_K1 = 26;                //   [a 2][ot 4]
constexpr int NFT = PT_BR + 6 * PBR;     // 312
constexpr int TBUF_BYTES = PT_TAIL_LO * 1024;
constexpr int TLDS_BYTES = 8 * 1024 + 2 * TBUF_BYTES + 8 * TL_BYTES;

__global__ __launch_bounds__(256) void gblock_packT_bf16_kernel(mgx_gblock_weights w, unsigned char* __restrict__ packt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NFT * 512) return;
    const int f = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
    const int i = lane & 31, h = lane >> 5;
    float v = 0.f;
    if (f < PT_TAIL) {                                   // K0 (2,2,2,128): [tap][c][n]
        const int it = f >> 1, ks = f & 1;
        if (i < 2)
            for (int tap = 0; tap < 4; ++tap) v += w.k0[(tap * 2 + i) * 128 + 32 * it + chan(8 * ks + j, h)];
    } else if (f < PT_BR) {
        const int g = f - PT_TAIL;
        if (g == 0) {                                    // W4 (8,2): slot (h, j<2) = output channel j (dY as hi + lo parts)
            v = (i < 8 && j < 2) ? w.w4[i * 2 + j] : 0.f;
        } else if (g < 3) {                              // W3 (64,8)
            v = j < 4 ? w.w3[(32 * (g - 1) + i) * 8 + j + 4 * h] : 0.f;
        } else if (g >= 4) {
            const int c = (g - 4) / TCH, q = (g - 4) % TCH;
            if (q < 4) {                                 // W2 (256,64)
                const int it = q >> 1, ks = q & 1;
                v = w.w2[(32 * c + i) * 64 + 32 * it + chan(8 * ks + j, h)];
            } else {                                     // W1 (128,256)
                const int ot = (q - 4) >> 1, ks = q & 1;
                v = w.w1[(32 * ot + i) * 256 + 32 * c + chan(8 * ks + j, h)];
            }
        }
    } else {
        const int k = (f - PT_BR) / PBR, g = (f - PT_BR) % PBR;
        if (g < PB_K2) {                                 // D2 (64,128)
            const int it = g >> 3, ot = (g >> 1) & 3, ks = g & 1;
            v = w.d2[k][(32 * it + i) * 128 + 32 * ot + chan(8 * ks + j, h)];
        } else if (g < PB_D1) {                          // K2 (1,2,64,32): [b][co][ci]
            const int q = g - PB_K2, b = q >> 2, it = (q >> 1) & 1, ks = q & 1;
            v = w.k2[k][(b * 64 + 32 * it + chan(8 * ks + j, h)) * 32 + i];
        } else if (g < PB_K1) {                          // D1 (16,32)
            const int ks = g - PB_D1;
            v = i < 16 ? w.d1[k][i * 32 + chan(8 * ks + j, h)] : 0.f;
        } else {                                         // K1 (2,1,16,128): [a][co][ci]
            const int q = g - PB_K1, a = q >> 2, ot = q & 3;
            v = w.k1[k][(a * 16 + chan(j, h)) * 128 + 32 * ot + i];
        }
    }
    *(bf16_t*)(packt + ((long)(f * 64 + lane) * 8 + j) * 2) = (bf16_t)v;
}

__device__ __forceinline__ void zero_tile(f32x16& t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.f;
}

template <class RM>
__global__ __launch_bounds__(256, 1) void gblock_dgrad_bf16_kernel(const unsigned char* __restrict__ packt,
                                                                   const float* __restrict__ dy,
                                                                   bf16_t* __restrict__ delta, float* __restrict__ dx,
                                                                   long pos0, long npix, int hl, int wl, long ntiles) {
    GB_USE_ROWS(RM);
    constexpr bool lean = RM::lean;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    unsigned char* kbuf = lds;                               // K0T fragments (8)
    unsigned char* pbuf = lds + 8 * 1024;                    // two phase buffers
    TapeOut tl = {lds + 8 * 1024 + 2 * TBUF_BYTES + wv * 2 * TL_BYTES, nullptr, 0, 0, 0};  // this wave's images
    // phase q of a sub-row: 0 = tail-lo, 1 = tail-hi, 2..7 = branches
    auto phase_f0 = [](int q) { return q == 0 ? PT_TAIL : (q == 1 ? PT_TAIL + PT_TAIL_LO : PT_BR + (q - 2) * PBR); };
    auto phase_n = [](int q) { return q == 0 ? PT_TAIL_LO : (q == 1 ? PT_TAIL_HI : PBR); };

    long tile = blockIdx.x;
    if (tile >= ntiles) return;
    stage(packt, PT_K0, 8, kbuf, wv, lane);
    stage(packt, phase_f0(0), phase_n(0), pbuf, wv, lane);
    __syncthreads();
    int par = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const long p = pos0 + tile * 128 + wv * 32 + r;
        const bool pvalid = p < npix;
        const long pc = pvalid ? p : 0;
        const long row = pc / wl;
        const int jcol = (int)(pc - row * wl);
        const float* dy0 = dy + ((row * 2) * (2L * wl) + 2L * jcol) * 2;      // output row 2i, column 2j
        bf16_t* dt = delta + (tile * 4 + wv) * (GB_ROWS * 32L);
        const bool more_tiles = tile + gridDim.x < ntiles;
        f32x16 dF[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) zero_tile(dF[it]);
#pragma unroll 1
        for (int a = 0; a < 2; ++a) {
            f32x16 dS[2][4];
            BTile dT2b[2][2];
            // ---- tail: dY -> dT3 -> dT2, then per chunk c of Dense(256): dT1_c -> dS, both sub-columns ----
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                stage(packt, phase_f0(half + 1), phase_n(half + 1), pbuf + (par ^ 1) * TBUF_BYTES, wv, lane);
                const unsigned char* wb = pbuf + par * TBUF_BYTES;
                if (half == 0) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int ab = a * 2 + b;
                        const float2 o = pvalid ? *(const float2*)(dy0 + (a * (2L * wl) + b) * 2) : make_float2(0.f, 0.f);
                        // B operand: slots (h = 0, j < 2) carry dY as bf16, slots (h = 1, j < 2) the remainder
                        const unsigned ohi = bf_pack(o.x, o.y);
                        const unsigned olo = bf_pack(o.x - bf_lo(ohi), o.y - bf_hi(ohi));
                        const u32x4 ob = (u32x4){h == 0 ? ohi : olo, 0u, 0u, 0u};
                        f32x16 dT3;
                        zero_tile(dT3);
                        dT3 = mfma_bf16(frag(wb, 0, lane), ob, dT3);
                        const BTile t3 = to_b(dT3);
                        f32x16 dT2;
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot) {
                            zero_tile(dT2);
                            dT2 = mfma_bf16(frag(wb, 1 + ot, lane), t3.k[0], dT2);
                            dT2b[b][ot] = to_b(dT2);
                            save_b<2>(dt, GB_T2 + ab * 64 + ot * 32, dT2b[b][ot], r, h, tl);
                        }
                        if (h == 0) store_pair(dt, GB_X + ab * 2, r, ohi);
#pragma unroll
                        for (int q = 0; q < 2; ++q) store_pair(dt, GB_T3 + ab * 8 + 4 * h + 2 * q, r, t3.k[0][q]);
#pragma unroll
                        for (int ot = 0; ot < 4; ++ot) zero_tile(dS[b][ot]);
                    }
                }
                const unsigned char* wch = wb + (half == 0 ? 4 * 1024 : 0);
#pragma unroll 1
                for (int cc = 0; cc < 4; ++cc) {
                    const unsigned char* wc = wch + cc * TCH * 1024;
                    const int c = half * 4 + cc;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x16 dT1;
                        zero_tile(dT1);
#pragma unroll
                        for (int it = 0; it < 2; ++it)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) dT1 = mfma_bf16(frag(wc, it * 2 + ks, lane), dT2b[b][it].k[ks], dT1);
                        const BTile t1 = to_b(dT1);
                        if (!lean) save_b<2>(dt, GB_T1 + (a * 2 + b) * 256 + c * 32, t1, r, h, tl);
#pragma unroll
                        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                dS[b][ot] = mfma_bf16(frag(wc, 4 + ot * 2 + ks, lane), t1.k[ks], dS[b][ot]);
                    }
                }
                __syncthreads();
                par ^= 1;
            }
            BTile dSb[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) {
                    dSb[b][ot] = to_b(dS[b][ot]);
                    save_b<2>(dt, GB_S + (a * 2 + b) * 128 + ot * 32, dSb[b][ot], r, h, tl);
                }
            // ---- branches ----
#pragma unroll 1
            for (int k = 0; k < 6; ++k) {
                if (k < 5) stage(packt, phase_f0(k + 3), phase_n(k + 3), pbuf + (par ^ 1) * TBUF_BYTES, wv, lane);
                else if (a == 0 || more_tiles) stage(packt, phase_f0(0), phase_n(0), pbuf + (par ^ 1) * TBUF_BYTES, wv, lane);
                const unsigned char* wb = pbuf + par * TBUF_BYTES;
                bf16_t* dbr = dt + (GB_BR + k * GB_BR_ROWS) * 32;
                f32x16 do2;
                zero_tile(do2);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    BTile do3b[2];
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        f32x16 do3;
                        zero_tile(do3);
#pragma unroll
                        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                do3 = mfma_bf16(frag(wb, PB_D2 + (it * 4 + ot) * 2 + ks, lane), dSb[b][ot].k[ks], do3);
                        do3b[it] = to_b(do3);
                        if (!lean) save_b<2>(dbr, GB_O3 + (a * 2 + b) * 64 + it * 32, do3b[it], r, h, tl);
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            do2 = mfma_bf16(frag(wb, PB_K2 + (b * 2 + it) * 2 + ks, lane), do3b[it].k[ks], do2);
                }
                const BTile do2b = to_b(do2);
                save_b<2>(dbr, GB_O2 + a * 32, do2b, r, h, tl);
                f32x16 do1;
                zero_tile(do1);
                do1 = mfma_bf16(frag(wb, PB_D1, lane), do2b.k[0], do1);
                do1 = mfma_bf16(frag(wb, PB_D1 + 1, lane), do2b.k[1], do1);
                const BTile do1b = to_b(do1);
                save_b<1>(dbr, GB_O1 + a * 16, do1b, r, h, tl);          // rows 0-15 = registers 0-7 of both lane halves
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
                    dF[ot] = mfma_bf16(frag(wb, PB_K1 + (a ? 4 : 0) + ot, lane), do1b.k[0], dF[ot]);
                __syncthreads();
                par ^= 1;
            }
        }
        // ---- dX = (sum of the four taps of K0)^T dF; the up-sampling backward is that tap sum ----
        f32x16 dxa;
        zero_tile(dxa);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const BTile fb = to_b(dF[it]);
            save_b<2>(dt, GB_F + it * 32, fb, r, h, tl);
            dxa = mfma_bf16(frag(kbuf, it * 2, lane), fb.k[0], dxa);
            dxa = mfma_bf16(frag(kbuf, it * 2 + 1, lane), fb.k[1], dxa);
        }
        if (pvalid && h == 0) *(float2*)(dx + pc * 2) = make_float2(dxa[0], dxa[1]);
    }
    tape_flush(tl, r, h);
}

}  // namespace

extern "C" int64_t mgx_gblock_pack_bf16_bytes(void) { return PACKB_BYTES; }
extern "C" int64_t mgx_gblock_packt_bf16_bytes(void) { return (int64_t)NFT * 1024; }

// bf16 elements of a side buffer (activations or gradients) for npos positions
extern "C" int64_t mgx_gblock_side_elems_bf16(int64_t npos) {
    return npos <= 0 ? 0 : ((npos + 127) / 128) * 4 * (GB_ROWS * 32L);
}
extern "C" int64_t mgx_gblock_side_elems_bf16_lean(int64_t npos) {
    return npos <= 0 ? 0 : ((npos + 127) / 128) * 4 * (GbRowsLean::ROWS * 32L);
}

namespace {
bool weights_set(const mgx_gblock_weights& w) {
    const void* const* ptrs = (const void* const*)&w;
    for (size_t i = 0; i < sizeof(w) / sizeof(void*); ++i)
        if (!ptrs[i]) return false;
    return true;
}
}  // namespace

// Waves per workgroup of the forward without a tape: 8 (two per SIMD, 256 registers each; measured 304 against
// 262 clips/s of stage-6 inference with 4, whose single wave per SIMD waits on its own conversions between
// MFMAs); MGX_GBLOCK_WAVES=4 in the environment selects the one-wave-per-SIMD form.
static int fwd_waves() {
    static const int nw = [] {
        const char* e = getenv("MGX_GBLOCK_WAVES");
        return (e && e[0] == '4') ? 4 : 8;
    }();
    return nw;
}

static int gblock_fwd_bf16_any(const mgx_gblock_weights* wp, void* packb, const float* x, float* y, void* act,
                               int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream, int lean) {
    if (!wp || !packb || !x || (!y && !act) || nimg <= 0 || hl <= 0 || wl <= 0) return MGX_ERR_ARG;
    const long npix = (long)nimg * hl * wl;
    if (pos0 < 0 || pos0 >= npix || (pos0 & 127)) return MGX_ERR_ARG;
    const long end = (npos <= 0 || pos0 + npos > npix) ? npix : pos0 + npos;
    const mgx_gblock_weights w = *wp;
    if (!weights_set(w)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long ntiles = (end - pos0 + 127) / 128;
    if (pos0 == 0)      // the packed fragments of a call sequence over ranges are filled by its first call
        hipLaunchKernelGGL(gblock_pack_bf16_kernel, dim3((NFB * 512 + 255) / 256), dim3(256), 0, st, w, (unsigned char*)packb);
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    if (act && lean) {
        if (hipFuncSetAttribute((const void*)gblock_fwd_bf16_kernel<true, 4, GbRowsLean>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MGX_ERR_LAUNCH;
        hipLaunchKernelGGL((gblock_fwd_bf16_kernel<true, 4, GbRowsLean>), dim3(grid), dim3(256), LDS_BYTES, st,
                           (const unsigned char*)packb, x, y, (bf16_t*)act, (long)pos0, end, hl, wl, ntiles);
    } else if (act) {
        if (hipFuncSetAttribute((const void*)gblock_fwd_bf16_kernel<true, 4, GbRowsFull>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MGX_ERR_LAUNCH;
        hipLaunchKernelGGL((gblock_fwd_bf16_kernel<true, 4, GbRowsFull>), dim3(grid), dim3(256), LDS_BYTES, st,
                           (const unsigned char*)packb, x, y, (bf16_t*)act, (long)pos0, end, hl, wl, ntiles);
    } else if (fwd_waves() == 8) {
        const long nt8 = (end - pos0 + 255) / 256;
        const unsigned grid8 = (unsigned)(nt8 < 256 ? nt8 : 256);
        if (hipFuncSetAttribute((const void*)gblock_fwd_bf16_kernel<false, 8, GbRowsFull>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return MGX_ERR_LAUNCH;
        hipLaunchKernelGGL((gblock_fwd_bf16_kernel<false, 8, GbRowsFull>), dim3(grid8), dim3(512), LDS_BYTES, st,
                           (const unsigned char*)packb, x, y, (bf16_t*)nullptr, (long)pos0, end, hl, wl, nt8);
    } else {
        if (hipFuncSetAttribute((const void*)gblock_fwd_bf16_kernel<false, 4, GbRowsFull>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return MGX_ERR_LAUNCH;
        hipLaunchKernelGGL((gblock_fwd_bf16_kernel<false, 4, GbRowsFull>), dim3(grid), dim3(256), LDS_BYTES, st,
                           (const unsigned char*)packb, x, y, (bf16_t*)nullptr, (long)pos0, end, hl, wl, ntiles);
    }
    return mgx_launch_status();
}

extern "C" int mgx_gblock_fwd_bf16(const mgx_gblock_weights* wp, void* packb, const float* x, float* y, void* act,
                                   int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream) {
    return gblock_fwd_bf16_any(wp, packb, x, y, act, nimg, hl, wl, pos0, npos, stream, 0);
}

extern "C" int mgx_gblock_fwd_bf16_lean(const mgx_gblock_weights* wp, void* packb, const float* x, float* y, void* act,
                                        int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream) {
    if (!act) return MGX_ERR_ARG;
    return gblock_fwd_bf16_any(wp, packb, x, y, act, nimg, hl, wl, pos0, npos, stream, 1);
}

static int gblock_dgrad_bf16_any(const mgx_gblock_weights* wp, void* packt, const float* dy, void* delta, float* dx,
                                 int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream, int lean) {
    if (!wp || !packt || !dy || !delta || !dx || nimg <= 0 || hl <= 0 || wl <= 0) return MGX_ERR_ARG;
    const long npix = (long)nimg * hl * wl;
    if (pos0 < 0 || pos0 >= npix || (pos0 & 127)) return MGX_ERR_ARG;
    const long end = (npos <= 0 || pos0 + npos > npix) ? npix : pos0 + npos;
    const mgx_gblock_weights w = *wp;
    if (!weights_set(w)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long ntiles = (end - pos0 + 127) / 128;
    if (pos0 == 0)
        hipLaunchKernelGGL(gblock_packT_bf16_kernel, dim3((NFT * 512 + 255) / 256), dim3(256), 0, st, w, (unsigned char*)packt);
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    auto launch = [&](auto kernel) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES) != hipSuccess)
            return false;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), TLDS_BYTES, st, (const unsigned char*)packt, dy, (bf16_t*)delta, dx,
                           (long)pos0, end, hl, wl, ntiles);
        return true;
    };
    if (!(lean ? launch(gblock_dgrad_bf16_kernel<GbRowsLean>) : launch(gblock_dgrad_bf16_kernel<GbRowsFull>)))
        return MGX_ERR_LAUNCH;
    return mgx_launch_status();
}

extern "C" int mgx_gblock_dgrad_bf16(const mgx_gblock_weights* wp, void* packt, const float* dy, void* delta, float* dx,
                                     int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream) {
    return gblock_dgrad_bf16_any(wp, packt, dy, delta, dx, nimg, hl, wl, pos0, npos, stream, 0);
}

extern "C" int mgx_gblock_dgrad_bf16_lean(const mgx_gblock_weights* wp, void* packt, const float* dy, void* delta, float* dx,
                                          int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream) {
    return gblock_dgrad_bf16_any(wp, packt, dy, delta, dx, nimg, hl, wl, pos0, npos, stream, 1);
}
