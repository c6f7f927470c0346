// Weight gradients of the widest layers of one generator grow block (add_generator_block,
// trainer/models.py:358-399; the tf.GradientTape sweep of models.py:144-146) WITHOUT their operands on the
// tape.  Of the 8138 rows a 32-position tile of the bf16 side buffers holds, 5120 belong to four tensors:
// the output of the six Conv2DTranspose(64) (c2), its gradient, the output of Dense(256) (t1) and its
// gradient.  Each of them is one linear layer away from rows that stay on the tape:
//     c2_k  = K2_k d1_k + b2_k          dc2_k = D2_k dsum
//     t1    = W1^T sum + e1             dt1   = W2 dt2
// so the kernels below stage the source rows of a tile in LDS, recompute the operand with MFMAs, and feed it
// to the weight-gradient MFMAs from registers.  The forward and data-gradient kernels then leave those rows
// unwritten and the backward moves 37 % of the bytes.
//
// Recomputing product: D[pos][co] = sum_ci X[ci][pos] W[ci][co].  Lane (r, h) supplies A[row r = pos][k = 8h+j]
// = X[ci0 + 8h + j][pos r], which ds_read_b64_tr_b16 delivers from rows staged as they lie in the side buffer
// (32 positions = 64 bytes a row), and B[k][col r] = W[ci0 + 8h + j][co r] from a register fragment loaded once
// per kernel.  The rows of A are handed out so that row m carries position 4 G(m >> 2) + (m & 3) with
// G(g) = (g >> 1) + 4 (g & 1) (the transposing read takes its four 8-byte chunks from any positions): lane
// (co, h) of the result then holds position 16 h + reg in register reg, and as bf16 the registers 8s .. 8s+7 are
// one operand of weight-gradient k-step s = positions 16 h + 8 s .. + 7.  The partner operand contracts the same
// positions in the same slots: 16 consecutive bytes of its staged row, one LDS read.
// The rows reach LDS by LDS-DMA two steps ahead of their use (three buffers per wave).  The DMA is issued from
// inline assembly: the compiler orders every LDS read behind all outstanding LDS-DMA it knows of (it cannot
// tell the buffers apart), which would make each step wait for the rows of the steps after it.
// Weight-gradient accumulators, slabs and bias sums have the layout of gblock_wgrad_bf16_kernel (gblock_bwd.hip),
// whose gblock_unpack_kernel adds them into the gradient tensors.
#include "bf16.h"
#include "gblock_jobs.h"
#include "../../include/mgx.h"

#include <type_traits>

namespace {

// Staged rows: 64 bytes each, the four 16-byte chunks of row `row` XOR-swizzled with bits 2-3 of the row so that
// the 16-byte reads of 16 consecutive rows (partner operands) cover all banks; the transposing reads (4
// consecutive rows x 8 chunks of 8 bytes per half wave) are conflict-free with or without it.
__device__ __forceinline__ int row_off(int row, int q8) {
    return row * 64 + ((((q8 >> 1) ^ (row >> 2)) & 3) << 4) + ((q8 & 1) << 3);
}

__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// One LDS-DMA instruction: lane l fetches 16 bytes from its global address to LDS byte `dst` + 16 l (dst wave-uniform).
__device__ __forceinline__ void dma16(const void* gptr, unsigned dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(gptr) : "memory");
}

// NB blocks of 16 consecutive side-buffer rows starting at `src` -> staged rows row0 .. by LDS-DMA: lane l of a
// block fills 16 bytes at row0 * 64 + 16 l and fetches the chunk of the row that belongs there.
template <int NB>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, unsigned buf, int row0, int lane) {
    const int rl = lane >> 2, c = lane & 3;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = row0 + 16 * i + rl;
        const int cg = (c ^ (row >> 2)) & 3;
        dma16(src + (16 * i + rl) * 32 + cg * 8, buf + (row0 + 16 * i) * 64);
    }
}

// A operand of a recomputing MFMA: staged rows row0 + 8h .. row0 + 8h + 7 of position r
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* buf, int row0, int lane) {
    const int g16 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int row = row0 + 8 * (g16 >> 1) + q;
    const int q8 = 2 * (g16 & 1) + (p >> 1) + 4 * (p & 1);          // G(4 (g16 & 1) + p)
    const u32x2 lo = tr_read(buf + row_off(row, q8));
    const u32x2 hi = tr_read(buf + row_off(row + 4, q8));
    return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}

// Operand of weight-gradient k-step ks from staged row `row`: positions 16 h + 8 ks .. + 7
__device__ __forceinline__ u32x4 perm_frag(const unsigned char* buf, int row, int ks, int h) {
    return *(const u32x4*)(buf + row_off(row, 2 * (2 * h + ks)));
}

// registers 8s .. 8s+7 of a recomputed tile as bf16
__device__ __forceinline__ u32x4 to_op(const f32x16& a, int s) {
    u32x4 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = bf_pack(a[8 * s + 2 * q], a[8 * s + 2 * q + 1]);
    return t;
}

// acc + the eight bf16 values of a fragment: v_dot2c_f32_bf16 against (1, 1) adds both halves of a dword at once
__device__ __forceinline__ float add8(float acc, u32x4 v) {
    const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned e = v[q];      // by value: __builtin_bit_cast applied to the element expression itself reads element 0
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, e), ones, acc, false);
    }
    return acc;
}

// eight consecutive float32 weights -> one bf16 fragment (round to nearest even, as the packed forward weights)
__device__ __forceinline__ u32x4 wfrag(const float* __restrict__ p) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
    return (u32x4){bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
}

__device__ __forceinline__ void splat(f32x16& t, float v) {
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = v;
}

// Wait until at most `batches` of the most recent DMA batches (N instructions each) are still in flight: vector
// memory operations complete in order, so everything issued before them has then landed.
template <int N>
__device__ __forceinline__ void dma_wait(int batches) {
    if (batches >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N) : "memory");
    else if (batches == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// position-tile walk of one wave: sub-position ab of tile pt, then ab + 1, ..., then tile pt + nw
struct Walk {
    long pt;
    int ab;
};
__device__ __forceinline__ Walk succ(Walk w, long nw) {
    if (++w.ab == 4) { w.ab = 0; w.pt += nw; }
    return w;
}

// all jobs of one slab index on one XCD: they stage the same rows, which then come from HBM once per XCD
__device__ __forceinline__ void xcd_map(int& xb, int& job) {
    xb = blockIdx.x;
    job = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        const unsigned q = lin >> 3;
        job = (int)(q % gridDim.y);
        xb = (int)((q / gridDim.y) * 8 + (lin & 7));
    }
}

template <int TA, int TB>
__device__ __forceinline__ float* slab_of(float* ws, const GJob& jb, long split) {
    return ws + jb.ws_off + split * (TA * TB * 1024L + TB * 64L);
}

__device__ __forceinline__ void put_tile(float* slab, int tile, const f32x16& acc, int lane) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) slab[(tile * 16 + reg) * 64 + lane] = acc[reg];
}

// ------------------------------------------------------------------------------------------
// branch k: dD2_k[in 64][out 128] = sum c2_k[in] dsum[out],  dK2_k[b][co 64][ci 32] = sum d1_k[ci] dc2_k[co],
// the bias gradients b2_k = sum dc2_k and (branch 0) c2 = sum dsum.  Staged per (tile, sub-position ab = 2a + b):
// rows 0-31 d1_k of sub-row a, rows 32-159 dsum of ab.
// ------------------------------------------------------------------------------------------
constexpr int BR_ROWS = 160, BR_BUF = BR_ROWS * 64;
constexpr int BR_LDS = 4 * 3 * BR_BUF;

__global__ __launch_bounds__(256, 1) void gblock_lean_branch_kernel(mgx_gblock_weights w, const bf16_t* __restrict__ act,
                                                                    const bf16_t* __restrict__ delta, GJobs jobs,
                                                                    float* __restrict__ ws, long ntiles) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    int xb, k;
    xcd_map(xb, k);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    unsigned char* buf = lds + wv * (3 * BR_BUF);
    const long split = (long)xb * 4 + wv, nw = (long)gridDim.x * 4;
    using RM = GbRowsLean;
    const int br = RM::BR + k * RM::BR_ROWS;

    // weight fragments: K2_k[b][co][ci] as [ci][co] of (b, output tile ot, k-step), D2_k[in][out] as [out][in]
    u32x4 k2f[2][2][2], d2f[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int s = 0; s < 2; ++s) k2f[b][ot][s] = wfrag(w.k2[k] + (b * 64 + ot * 32 + r) * 32 + 16 * s + 8 * h);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int s = 0; s < 8; ++s) d2f[it][s] = wfrag(w.d2[k] + (it * 32 + r) * 128 + 16 * s + 8 * h);
    const float b2v[2] = {w.b2[k][r], w.b2[k][32 + r]};

    f32x16 accD[2][4], accK[2][2];
    float bsS[4] = {0.f, 0.f, 0.f, 0.f}, bs2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) splat(accD[t][u], 0.f);
#pragma unroll
        for (int u = 0; u < 2; ++u) splat(accK[t][u], 0.f);
    }

    const unsigned lbase = lds_addr(buf);
    auto issue = [&](Walk wk, int slot) {
        stage_rows<2>(act + (wk.pt * RM::ROWS + br + RM::O2 + (wk.ab >> 1) * 32) * 32L, lbase + slot * BR_BUF, 0, lane);
        stage_rows<8>(delta + (wk.pt * RM::ROWS + RM::S + wk.ab * 128) * 32L, lbase + slot * BR_BUF, 32, lane);
    };
    auto step = [&](auto bc, const unsigned char* cb) {
        constexpr int B = decltype(bc)::value;
        // every LDS read of the step first: the MFMAs below start as their operands arrive
        u32x4 ad[2], as[8], bsum[2][4], ad1[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) ad[s] = tr_frag(cb, 16 * s, lane);
#pragma unroll
        for (int s = 0; s < 8; ++s) as[s] = tr_frag(cb, 32 + 16 * s, lane);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int u = 0; u < 4; ++u) bsum[s][u] = perm_frag(cb, 32 + 32 * u + r, s, h);
            ad1[s] = perm_frag(cb, r, s, h);
        }
        __builtin_amdgcn_sched_barrier(0);
        // c2 of this sub-position from d1 (32 inputs: 2 k-steps), starting from the bias as the forward does;
        // dc2 from dsum (128 inputs: 8 k-steps).  The four tiles advance together: no MFMA waits for the one before.
        f32x16 c2[2], dc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            splat(c2[t], b2v[t]);
            splat(dc[t], 0.f);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) c2[t] = mfma_bf16(ad[s], k2f[B][t][s], c2[t]);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) dc[t] = mfma_bf16(as[s], d2f[t][s], dc[t]);
        u32x4 c2op[2][2], dcop[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) c2op[t][s] = to_op(c2[t], s);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bsS[u] = add8(bsS[u], bsum[s][u]);
#pragma unroll
                for (int t = 0; t < 2; ++t) accD[t][u] = mfma_bf16(c2op[t][s], bsum[s][u], accD[t][u]);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s) dcop[t][s] = to_op(dc[t], s);
            bs2[t] = add8(add8(bs2[t], dcop[t][0]), dcop[t][1]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) accK[B][u] = mfma_bf16(ad1[s], dcop[u][s], accK[B][u]);
    };

    // three staging buffers: the rows of step i + 2 are requested before step i is computed
    Walk w0 = {split, 0};
    Walk w1 = succ(w0, nw), w2 = succ(w1, nw);
    if (w0.pt < ntiles) issue(w0, 0);
    if (w1.pt < ntiles) issue(w1, 1);
    int slot = 0;
    auto advance = [&](auto bc) {
        const int s2 = slot == 0 ? 2 : slot - 1;                     // (slot + 2) % 3: read two steps ago
        if (w2.pt < ntiles) issue(w2, s2);
        dma_wait<10>((w1.pt < ntiles) + (w2.pt < ntiles));
        step(bc, buf + slot * BR_BUF);
        asm volatile("" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
        w0 = w1;
        w1 = w2;
        w2 = succ(w2, nw);
    };
    while (w0.pt < ntiles) {                                          // sub-positions alternate b = 0, 1
        advance(std::integral_constant<int, 0>{});
        advance(std::integral_constant<int, 1>{});
    }

    float* sd = slab_of<2, 4>(ws, jobs.j[4 + k], split);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) put_tile(sd, t * 4 + u, accD[t][u], lane);
#pragma unroll
    for (int u = 0; u < 4; ++u) sd[8 * 1024 + u * 64 + lane] = bsS[u];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float* sk = slab_of<1, 2>(ws, jobs.j[20 + 2 * k + b], split);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            put_tile(sk, u, accK[b][u], lane);
            sk[2 * 1024 + u * 64 + lane] = b == 0 ? bs2[u] : 0.f;      // the sum over both sub-columns sits with b = 0
        }
    }
}

// ------------------------------------------------------------------------------------------
// tail quarter q (channels c0 = 64 q .. c0 + 63 of Dense(256)): dW1[in 128][c] = sum sum[in] dt1[c],
// dW2[c][out 64] = sum t1[c] dt2[out], the bias gradients e1[c] = sum dt1[c] and (quarter 0) e2 = sum dt2.
// Staged per (tile, sub-position ab): rows 0-127 sum, rows 128-191 dt2.
// ------------------------------------------------------------------------------------------
constexpr int TL_ROWS = 192, TL_BUF = TL_ROWS * 64;
constexpr int TL_LDS = 4 * 3 * TL_BUF;

__global__ __launch_bounds__(256, 1) void gblock_lean_tail_kernel(mgx_gblock_weights w, const bf16_t* __restrict__ act,
                                                                  const bf16_t* __restrict__ delta, GJobs jobs,
                                                                  float* __restrict__ ws, long ntiles) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    int xb, q;
    xcd_map(xb, q);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    unsigned char* buf = lds + wv * (3 * TL_BUF);
    const long split = (long)xb * 4 + wv, nw = (long)gridDim.x * 4;
    using RM = GbRowsLean;
    const int c0 = 64 * q;

    // W1[in 128][out 256] as [in][c] of (channel tile uo, k-step); W2[c 256][out 64] as [out][c]
    u32x4 w1f[2][8], w2f[2][4];
#pragma unroll
    for (int uo = 0; uo < 2; ++uo) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w.w1[(16 * s + 8 * h + j) * 256 + c0 + 32 * uo + r];
            w1f[uo][s] = (u32x4){bf_pack(v[0], v[1]), bf_pack(v[2], v[3]), bf_pack(v[4], v[5]), bf_pack(v[6], v[7])};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) w2f[uo][s] = wfrag(w.w2 + (c0 + 32 * uo + r) * 64 + 16 * s + 8 * h);
    }
    const float e1v[2] = {w.e1[c0 + r], w.e1[c0 + 32 + r]};

    f32x16 accW1[4][2], accW2[2][2];
    float bsE1[2] = {0.f, 0.f}, bsE2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) splat(accW1[t][u], 0.f);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) splat(accW2[t][u], 0.f);

    const unsigned lbase = lds_addr(buf);
    auto issue = [&](Walk wk, int slot) {
        stage_rows<8>(act + (wk.pt * RM::ROWS + RM::S + wk.ab * 128) * 32L, lbase + slot * TL_BUF, 0, lane);
        stage_rows<4>(delta + (wk.pt * RM::ROWS + RM::T2 + wk.ab * 64) * 32L, lbase + slot * TL_BUF, 128, lane);
    };

    // three staging buffers: the rows of step i + 2 are requested before step i is computed
    Walk w0 = {split, 0};
    Walk w1 = succ(w0, nw), w2 = succ(w1, nw);
    if (w0.pt < ntiles) issue(w0, 0);
    if (w1.pt < ntiles) issue(w1, 1);
    int slot = 0;
    while (w0.pt < ntiles) {
        const int s2 = slot == 0 ? 2 : slot - 1;                     // (slot + 2) % 3: read two steps ago
        if (w2.pt < ntiles) issue(w2, s2);
        dma_wait<12>((w1.pt < ntiles) + (w2.pt < ntiles));
        const unsigned char* cb = buf + slot * TL_BUF;
        // LDS reads run one group ahead of the MFMAs that consume them (all at once they would not fit the registers;
        // the scheduling barriers keep the compiler from moving a group next to its consumers)
        u32x4 as[8], at[4], asum[2][4], bt2[2][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) at[s] = tr_frag(cb, 128 + 16 * s, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) as[s] = tr_frag(cb, 16 * s, lane);
        __builtin_amdgcn_sched_barrier(0);
        // dt1 of this quarter from dt2 (64 inputs); t1 of this quarter from sum (128 inputs), starting from the bias
        // as the forward does.  Tiles of a pair advance together: no MFMA waits for the one before.
        f32x16 t1[2], d1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            splat(t1[u], e1v[u]);
            splat(d1[u], 0.f);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) d1[u] = mfma_bf16(at[s], w2f[u][s], d1[u]);
#pragma unroll
        for (int s = 4; s < 8; ++s) as[s] = tr_frag(cb, 16 * s, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) t1[u] = mfma_bf16(as[s], w1f[u][s], t1[u]);
#pragma unroll
        for (int t = 0; t < 4; ++t) asum[0][t] = perm_frag(cb, 32 * t + r, 0, h);
#pragma unroll
        for (int u = 0; u < 2; ++u) bt2[0][u] = perm_frag(cb, 128 + 32 * u + r, 0, h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 4; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) t1[u] = mfma_bf16(as[s], w1f[u][s], t1[u]);
#pragma unroll
        for (int t = 0; t < 4; ++t) asum[1][t] = perm_frag(cb, 32 * t + r, 1, h);
#pragma unroll
        for (int u = 0; u < 2; ++u) bt2[1][u] = perm_frag(cb, 128 + 32 * u + r, 1, h);
        u32x4 t1op[2][2], d1op[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int s = 0; s < 2; ++s) d1op[u][s] = to_op(d1[u], s);
            bsE1[u] = add8(add8(bsE1[u], d1op[u][0]), d1op[u][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) accW1[t][u] = mfma_bf16(asum[s][t], d1op[u][s], accW1[t][u]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) t1op[u][s] = to_op(t1[u], s);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bsE2[u] = add8(bsE2[u], bt2[s][u]);
#pragma unroll
                for (int t = 0; t < 2; ++t) accW2[t][u] = mfma_bf16(t1op[t][s], bt2[s][u], accW2[t][u]);
            }
        asm volatile("" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
        w0 = w1;
        w1 = w2;
        w2 = succ(w2, nw);
    }

    // W1 jobs (input half, output half) hold 2 x 4 tiles: this quarter is output tiles 2 (q & 1), 2 (q & 1) + 1 of
    // output half q >> 1; W2 jobs (row half q >> 1) hold 4 x 2 tiles: this quarter is row tiles 2 (q & 1), + 1.
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float* s1 = slab_of<2, 4>(ws, jobs.j[(t >> 1) * 2 + (q >> 1)], split);
#pragma unroll
        for (int u = 0; u < 2; ++u) put_tile(s1, (t & 1) * 4 + 2 * (q & 1) + u, accW1[t][u], lane);
    }
    {
        float* s1 = slab_of<2, 4>(ws, jobs.j[q >> 1], split);
#pragma unroll
        for (int u = 0; u < 2; ++u) s1[8 * 1024 + (2 * (q & 1) + u) * 64 + lane] = bsE1[u];
        float* s2 = slab_of<4, 2>(ws, jobs.j[10 + (q >> 1)], split);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) put_tile(s2, (2 * (q & 1) + t) * 2 + u, accW2[t][u], lane);
        if (q == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) s2[8 * 1024 + u * 64 + lane] = bsE2[u];
        }
    }
}

}  // namespace

int mgx_gblock_lean_launch(const mgx_gblock_weights& w, const void* act, const void* delta, const GJobs& J, float* ws,
                           long ntiles, hipStream_t st) {
    static std::atomic<uint64_t> done_b{0}, done_t{0};
    mgx_allow_dynamic_lds(gblock_lean_branch_kernel, BR_LDS, done_b);
    mgx_allow_dynamic_lds(gblock_lean_tail_kernel, TL_LDS, done_t);
    hipLaunchKernelGGL(gblock_lean_branch_kernel, dim3(lean_grid_x(ntiles, LEAN_BRANCH_JOBS), LEAN_BRANCH_JOBS), dim3(256),
                       BR_LDS, st, w, (const bf16_t*)act, (const bf16_t*)delta, J, ws, ntiles);
    hipLaunchKernelGGL(gblock_lean_tail_kernel, dim3(lean_grid_x(ntiles, LEAN_TAIL_JOBS), LEAN_TAIL_JOBS), dim3(256), TL_LDS,
                       st, w, (const bf16_t*)act, (const bf16_t*)delta, J, ws, ntiles);
    return mgx_launch_status();
}
