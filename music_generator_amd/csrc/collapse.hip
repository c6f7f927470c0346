// Generator inference through the collapsed affine map (an opt-in inference path, not the train step).
//
// Every layer of the generator in front of its final LayerNormalization is linear and acts on one position
// or on disjoint windows (trainer/models.py:358-399, 412-450: Conv2D with stride = kernel, Conv2DTranspose
// with stride = kernel, Dense, Add, UpSampling2D, WeightedSum), so the tensor that enters the normalisation
// is, for latent column pair c (4 numbers z[c][0..3]) and position p of the H x WP patch that pair produces,
//     pre[h][c * WP + j][ch] = T[p][ch][0] + sum_k z[c][k] * T[p][ch][1 + k],        p = h * WP + j,
// with one table T for all column pairs.  The host obtains T by running the layer kernels on five probe
// latents (zero and the four unit vectors).  These kernels apply T:
//   * patch_gram_kernel: sums and Gram matrix of the five table vectors in float64 (once per table);
//   * collapse_stats_kernel: mean and 1/sqrt(var + eps) of every clip from the Gram matrix - sum and sum of
//     squares of `pre` are a linear and a quadratic form in the clip's latent numbers, no pass over the clip;
//   * collapse_apply_kernel: out = gamma * (pre - mean) * rstd + beta, one write of the clip, gamma / beta read
//     once per launch for all its clips.
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int NK = 5;            // table vectors: offset + four latent numbers
constexpr int NG = 20;           // 5 sums + 15 products (k <= l)
constexpr int CG = 5;            // latent column pairs per workgroup of the apply kernel

// probe[k][h][c * WP + j][ch], k = 0 the zero latent, k = 1..4 the unit latents: the table is taken from the
// patch of column pair 0
__global__ __launch_bounds__(256) void collapse_table_kernel(const float* __restrict__ probe, float* __restrict__ table,
                                                            int h, int wp, int ncols) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;          // (position, channel)
    if (v >= (long)h * wp * 2) return;
    const long p = v >> 1;
    const int ch = (int)(v & 1), row = (int)(p / wp), j = (int)(p - (long)row * wp);
    const long w = (long)ncols * wp, clip = (long)h * w * 2;
    const long e = ((long)row * w + j) * 2 + ch;
    const float b = probe[e];
    table[v * NK] = b;
#pragma unroll
    for (int k = 1; k < NK; ++k) table[v * NK + k] = probe[k * clip + e] - b;
}

__global__ __launch_bounds__(1024) void patch_gram_kernel(const float* __restrict__ table, long nvec,
                                                           double* __restrict__ gram) {
    __shared__ double red[16][NG];
    double acc[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) acc[i] = 0.0;
    for (long v = threadIdx.x; v < nvec; v += 1024) {
        double t[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) t[k] = (double)table[v * NK + k];
        int o = NK;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            acc[k] += t[k];
#pragma unroll
            for (int l = k; l < NK; ++l) acc[o++] += t[k] * t[l];
        }
    }
    // fixed-order reduction: lanes of a wave by halving, then the 16 waves in order
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        double a = acc[i];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) red[wv][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < NG) {
        double a = 0.0;
        for (int w = 0; w < 16; ++w) a += red[w][threadIdx.x];
        gram[threadIdx.x] = a;
    }
}

// one wave per clip; lane c < ncols owns column pair c
__global__ __launch_bounds__(64) void collapse_stats_kernel(const float* __restrict__ z, const double* __restrict__ gram,
                                                           float* __restrict__ stats, int ncols, double count,
                                                           double eps) {
    const int n = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int c = lane; c < ncols; c += 64) {
        double t[NK];
        t[0] = 1.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[1 + k] = (double)z[((long)n * ncols + c) * 4 + k];
        int o = NK;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            s += gram[k] * t[k];
#pragma unroll
            for (int l = k; l < NK; ++l) q += (k == l ? 1.0 : 2.0) * gram[o++] * t[k] * t[l];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        q += __shfl_down(q, off, 64);
    }
    if (lane == 0) {
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * n] = (float)mean;
        stats[2 * n + 1] = (float)(1.0 / sqrt(var + eps));
    }
}

// grid (ceil(H * WP / PP / 256), ceil(ncols / CG)); thread = PP adjacent patch positions of a row, both channels
// (PP = 2 when WP is even: 16-byte stores)
template <int PP>
__global__ __launch_bounds__(256) void collapse_apply_kernel(const float* __restrict__ table, const float* __restrict__ z,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            int nclips, int h, int wp, int ncols) {
    typedef float vec_t __attribute__((ext_vector_type(2 * PP)));
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * PP;
    if (p >= (long)h * wp) return;
    const int row = (int)(p / wp), j = (int)(p - (long)row * wp);
    float t[2 * PP][NK];
#pragma unroll
    for (int v = 0; v < 2 * PP; ++v)
#pragma unroll
        for (int k = 0; k < NK; ++k) t[v][k] = table[(p * 2 + v) * NK + k];
    const long w = (long)ncols * wp;
    const long clip = (long)h * w * 2;
    const int c0 = blockIdx.y * CG;
    for (int c = c0; c < c0 + CG && c < ncols; ++c) {
        const long e = ((long)row * w + (long)c * wp + j) * 2;
        const vec_t g = *(const vec_t*)(gamma + e), b = *(const vec_t*)(beta + e);
        for (int n = 0; n < nclips; ++n) {
            const float4 zv = *(const float4*)(z + ((long)n * ncols + c) * 4);       // uniform over the workgroup
            const float mean = stats[2 * n], rstd = stats[2 * n + 1];
            vec_t o;
#pragma unroll
            for (int v = 0; v < 2 * PP; ++v) {
                float x = t[v][0];
                x = fmaf(zv.x, t[v][1], x);
                x = fmaf(zv.y, t[v][2], x);
                x = fmaf(zv.z, t[v][3], x);
                x = fmaf(zv.w, t[v][4], x);
                o[v] = fmaf((x - mean) * rstd, g[v], b[v]);
            }
            __builtin_nontemporal_store(o, (vec_t*)(out + n * clip + e));
        }
    }
}

}  // namespace

extern "C" int mgx_collapse_table(const float* probe, float* table, double* gram, int h, int wp, int ncols,
                                  void* stream) {
    if (!probe || !table || !gram || h <= 0 || wp <= 0 || ncols <= 0) return MGX_ERR_ARG;
    const long nvec = (long)h * wp * 2;
    if (nvec >= (1L << 31)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(collapse_table_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, probe, table, h, wp,
                       ncols);
    hipLaunchKernelGGL(patch_gram_kernel, dim3(1), dim3(1024), 0, st, table, nvec, gram);
    return mgx_launch_status();
}

extern "C" int mgx_collapse_apply(const float* table, const double* gram, const float* z, const float* gamma,
                                  const float* beta, float* out, float* stats, int nclips, int h, int wp, int ncols,
                                  float eps, void* stream) {
    if (!table || !gram || !z || !gamma || !beta || !out || !stats) return MGX_ERR_ARG;
    if (nclips <= 0 || h <= 0 || wp <= 0 || ncols <= 0 || nclips > 65535) return MGX_ERR_ARG;
    const long positions = (long)h * wp;
    if (positions >= (1L << 31) || (long)h * wp * ncols * 2 >= (1L << 40)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(collapse_stats_kernel, dim3(nclips), dim3(64), 0, st, z, gram, stats, ncols,
                       (double)positions * ncols * 2.0, (double)eps);
    const unsigned gy = (unsigned)((ncols + CG - 1) / CG);
    if (wp % 2 == 0)
        hipLaunchKernelGGL(collapse_apply_kernel<2>, dim3((unsigned)((positions / 2 + 255) / 256), gy), dim3(256), 0, st,
                           table, z, stats, gamma, beta, out, nclips, h, wp, ncols);
    else
        hipLaunchKernelGGL(collapse_apply_kernel<1>, dim3((unsigned)((positions + 255) / 256), gy), dim3(256), 0, st,
                           table, z, stats, gamma, beta, out, nclips, h, wp, ncols);
    return mgx_launch_status();
}
