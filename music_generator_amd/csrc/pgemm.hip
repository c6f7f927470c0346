// Dense layers on many rows: y[R][N] = x[R][K] W + b with K, N in {64, 128, 256} (the per-pixel
// Dense chain of the stage-0 generator and its data-gradients), float32 on v_mfma_f32_32x32x2_f32.
//
// The whole weight matrix lives in LDS for the launch as [k/4][n][4], so a lane (column n, half h)
// gets four consecutive k with one conflict-free ds_read_b128.  A 4-wave workgroup per CU walks
// tiles of 128 rows; a wave owns 32 rows and keeps N/32 accumulator tiles.  Its A operand comes
// straight from global memory: lane (row r, half h) loads 16 bytes = 4 consecutive k of its row per
// 8-k step, and ALL steps of the NEXT tile are requested before the MFMAs of the current one, so a
// single wave per SIMD has a whole tile of matrix work to cover the HBM latency.  Loads and stores
// go through buffer resources that span exactly the wave's rows: rows past the end read zeros /
// are dropped by the range check, per-register offsets are scalars.
#include "common.h"
#include "../../include/mgx.h"

namespace {

// KS = K / 8 steps, NT = N / 32 column tiles, WT: weights stored [n][k] (data-gradient) instead of [k][n]
template <int KS, int NT, bool WT>
__global__ __launch_bounds__(256, 1) void pgemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, long R,
                                                       long wsk, long wsn, long ntiles) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [K/4][N][4]
    constexpr int K = KS * 8, N = NT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    for (int i = tid; i < (K / 4) * N; i += 256) {
        const int kq = i / N, n = i - kq * N;
        f32x4 v;
        if (WT) v = *(const f32x4*)(w + (long)n * wsn + kq * 4);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = w[(long)(kq * 4 + j) * wsk + n];
        }
        *(f32x4*)(wl + (long)i * 4) = v;
    }
    __syncthreads();

    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = bias ? bias[nt * 32 + r] : 0.f;

    const unsigned a_voff = (unsigned)((r * K + 4 * h) * 4);
    auto load_tile = [&](long tile, f32x4 (&a)[KS]) {
        const long row0 = tile * 128 + wv * 32;
        long rows = R - row0;
        rows = rows < 0 ? 0 : (rows > 32 ? 32 : rows);
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + (row0 < R ? row0 : 0) * K), 0, (unsigned)(rows * K * 4), 0x00020000);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            a[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff, s * 32, 0));
    };
    auto compute_store = [&](long tile, const f32x4 (&a)[KS]) {
        const long row0 = tile * 128 + wv * 32;
        if (row0 >= R) return;                                       // wave-uniform
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
        const float* wb = wl + (h * N + r) * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4 wq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wq[nt] = *(const f32x4*)(wb + (2 * s * N + nt * 32) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a[s][j], wq[nt][j], acc[nt]);
        }
        long rows = R - row0;
        rows = rows > 32 ? 32 : rows;
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void*)(y + row0 * N), 0, (unsigned)(rows * N * 4), 0x00020000);
        const unsigned y_voff = (unsigned)((4 * h * N + r) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[nt][reg] + bv[nt]), ry, y_voff,
                                                      (((reg & 3) + 8 * (reg >> 2)) * N + nt * 32) * 4, 0);
    };

    // two register sets of A operands, swapped by unrolling the tile loop twice
    f32x4 a0[KS], a1[KS];
    long t = blockIdx.x;
    const long step = gridDim.x;
    if (t < ntiles) load_tile(t, a0);
    while (t < ntiles) {
        if (t + step < ntiles) load_tile(t + step, a1);
        compute_store(t, a0);
        t += step;
        if (t >= ntiles) break;
        if (t + step < ntiles) load_tile(t + step, a0);
        compute_store(t, a1);
        t += step;
    }
}

template <int KS, int NT>
int launch_kn(const mgx_conv_desc& d, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const long R = (long)d.nimg * d.hout * d.wout;
    const long ntiles = (R + 127) / 128;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    constexpr int lds = KS * 8 * NT * 32 * 4;
    const bool wt = d.wsk == 1;
#define MGX_PG(WW)                                                                                                   \
    do {                                                                                                             \
        static std::atomic<uint64_t> allowed{0};                                                                     \
        mgx_allow_dynamic_lds(pgemm_kernel<KS, NT, WW>, lds, allowed);                                               \
        hipLaunchKernelGGL((pgemm_kernel<KS, NT, WW>), dim3(grid), dim3(256), lds, st, x, w, bias, y, R, (long)d.wsk, \
                           (long)d.wsn, ntiles);                                                                     \
    } while (0)
    if (wt) MGX_PG(true); else MGX_PG(false);
#undef MGX_PG
    return mgx_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// Data-gradient of a Conv2DTranspose whose kernel (1, T) equals its stride, 64 filters -> 32 channels (the
// (1,15) layers of the stage-0 generator, models.py:422): dX[row][c] = sum_t sum_f dY[row][t][f] W[t][f][c],
// a GEMM with few rows (200 per sample) and K = 64 T.  One wave per tap: a workgroup of 16 waves takes 32
// rows, wave t multiplies its tap's 64-deep slice, and the T partial tiles are added through LDS in tap order.
// (With the rows on the workgroups alone a batch of 16 gives 25 workgroups walking K = 960 one after the
// other: 146 us; this form: all taps at once.)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void tapsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ y, long R, int T) {
    __shared__ float red[16][16][64];
    const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * 32;
    if (t < T) {
        long rows = R - row0;
        rows = rows > 32 ? 32 : rows;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + row0 * T * 64), 0,
                                                                            (unsigned)(rows * T * 64 * 4), 0x00020000);
        const unsigned a_voff = (unsigned)(((r * T + t) * 64 + 4 * h) * 4);
        f32x4 a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            a[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff, s * 32, 0));
        const float* wt = w + (long)t * 2048 + (4 * h) * 32 + r;
        float b[8][4];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) b[s][j] = wt[(8 * s + j) * 32];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma32(a[s][j], b[s][j], acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) red[t][i][lane] = acc[i];
    }
    __syncthreads();
    const int reg = tid >> 6;
    float v = red[0][reg][lane];
    for (int q = 1; q < T; ++q) v += red[q][reg][lane];
    const long row = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (row < R) y[row * 32 + r] = v;
}

}  // namespace

bool mgx_tapsum_applicable(const mgx_conv_desc& d, bool mul) {
    if (mul || d.accumulate || d.act_alpha > 0.f || d.nbias != 0 || d.flip) return false;
    if (d.kh != 1 || d.sh != 1 || d.kw < 2 || d.kw > 16 || d.sw != d.kw || d.dkh != 1 || d.dkw != 1) return false;
    if (d.pt != 0 || d.pl != 0 || d.hin != d.hout || d.win != d.wout * d.kw) return false;
    if (d.cin != 64 || d.ngemm != 32 || d.cout != 32) return false;
    if (d.wsn != 1 || d.wsk != 32 || d.wtap != 2048) return false;
    return (long)32 * d.kw * 64 * 4 < (1L << 31);
}

int mgx_tapsum_launch(const mgx_conv_desc& d, const float* x, const float* w, float* y, hipStream_t st) {
    const long R = (long)d.nimg * d.hout * d.wout;
    const long grid = (R + 31) / 32;
    if (grid >= (1L << 31)) return MGX_ERR_ARG;
    hipLaunchKernelGGL(tapsum_kernel, dim3((unsigned)grid), dim3(1024), 0, st, x, w, y, R, (int)d.kw);
    return mgx_launch_status();
}

// 1x1 stride-1 contraction without input multiplier, activation, accumulation or scatter, K and N in
// {64, 128, 256} with K*N*4 bytes of weights <= 128 KiB, dense [k][n] or [n][k] weights.
bool mgx_pgemm_applicable(const mgx_conv_desc& d, bool mul) {
    if (mul || d.accumulate || d.act_alpha > 0.f) return false;
    if (d.kh != 1 || d.kw != 1 || d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1 || d.pt != 0 || d.pl != 0)
        return false;
    if (d.hout != d.hin || d.wout != d.win || d.cout != d.ngemm) return false;
    const bool ksize = d.cin == 64 || d.cin == 128 || d.cin == 256;
    const bool nsize = d.ngemm == 64 || d.ngemm == 128 || d.ngemm == 256;
    if (!ksize || !nsize || (long)d.cin * d.ngemm * 4 > 131072) return false;
    if (d.nbias != 0 && d.nbias != d.ngemm) return false;
    const bool kn = d.wsn == 1 && d.wsk == d.ngemm, nk = d.wsk == 1 && d.wsn == d.cin;
    return kn || nk;
}

int mgx_pgemm_launch(const mgx_conv_desc& d, const float* x, const float* w, const float* bias, float* y,
                     hipStream_t st) {
    const float* b = d.nbias > 0 ? bias : nullptr;
    const int ks = d.cin / 8, nt = d.ngemm / 32;
#define MGX_KN(KK, NN) if (ks == KK && nt == NN) return launch_kn<KK, NN>(d, x, w, b, y, st)
    MGX_KN(8, 2); MGX_KN(8, 4); MGX_KN(8, 8);
    MGX_KN(16, 2); MGX_KN(16, 4); MGX_KN(16, 8);
    MGX_KN(32, 2); MGX_KN(32, 4);
#undef MGX_KN
    return MGX_ERR_ARG;
}
