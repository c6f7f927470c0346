// Reduction-shaped layers: LayerNormalization(axis=[1,2,3]), MinibatchStdDev, Flatten+Dense(1),
// the WGAN losses and their logit-gradient seeds.  All cross-workgroup sums are two-stage with
// fixed order (partials in a caller workspace, double precision), so results are reproducible.
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int kBlock = 256;

inline long ln_blocks_per_sample(long nimg, long e) {
    long b = 2048 / (nimg > 0 ? nimg : 1);
    const long bmax = (e + 4095) / 4096;
    if (b > bmax) b = bmax;
    if (b < 1) b = 1;
    return b;
}

// ---------------- LayerNormalization forward ----------------
__global__ __launch_bounds__(kBlock) void ln_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                            long e, long bps) {
    __shared__ double red[4];
    const long img = blockIdx.y;
    const long b = blockIdx.x;
    const long chunk = ((e + bps - 1) / bps + 3) & ~3L;
    const long beg = b * chunk;
    long end = beg + chunk;
    if (end > e) end = e;
    const float* src = x + img * e;
    double s = 0.0, q = 0.0;
    if ((e & 3) == 0) {
        // 16-byte loads (chunk and e are multiples of 4, the sample base is 16-byte aligned); the four values
        // of a load are summed in float32 pairs, the running sums are double
        const float4* s4 = (const float4*)src;
        for (long i = beg / 4 + threadIdx.x; i < end / 4; i += kBlock) {
            const float4 v = s4[i];
            s += (double)((v.x + v.y) + (v.z + v.w));
            q += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
        }
    } else {
        for (long i = beg + threadIdx.x; i < end; i += kBlock) {
            const double v = (double)src[i];
            s += v;
            q += v * v;
        }
    }
    const double ts = block_sum256_d(s, red);
    const double tq = block_sum256_d(q, red);
    if (threadIdx.x == 0) {
        part[(img * bps + b) * 2] = ts;
        part[(img * bps + b) * 2 + 1] = tq;
    }
}

__global__ __launch_bounds__(64) void ln_stats_kernel(const double* __restrict__ part, float* __restrict__ stats, long e,
                                                      long bps, float eps) {
    const long img = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (long b = threadIdx.x; b < bps; b += 64) {
        s += part[(img * bps + b) * 2];
        q += part[(img * bps + b) * 2 + 1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x == 0) {
        const double mean = s / (double)e;
        double var = q / (double)e - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[img * 2] = (float)mean;
        stats[img * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(kBlock) void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ y,
                                                          long e) {
    const long img = blockIdx.y;
    const float mean = stats[img * 2], rstd = stats[img * 2 + 1];
    const float* src = x + img * e;
    float* dst = y + img * e;
    if ((e & 3) == 0) {
        const long e4 = e / 4;
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e4; i += (long)gridDim.x * kBlock) {
            const float4 v = ((const float4*)src)[i], g = ((const float4*)gamma)[i], b = ((const float4*)beta)[i];
            ((float4*)dst)[i] = make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y,
                                            (v.z - mean) * rstd * g.z + b.z, (v.w - mean) * rstd * g.w + b.w);
        }
        return;
    }
    const long e2 = e / 2;  // e is always even (2 channels innermost)
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e2; i += (long)gridDim.x * kBlock) {
        const float2 v = ((const float2*)src)[i], g = ((const float2*)gamma)[i], b = ((const float2*)beta)[i];
        ((float2*)dst)[i] = make_float2((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y);
    }
    if ((e & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        dst[e - 1] = (src[e - 1] - mean) * rstd * gamma[e - 1] + beta[e - 1];
}

// The same for samples whose gamma / beta do not stay in the L2 between samples (1 M elements and more): one
// thread per four element positions loads gamma and beta once and walks the batch, so the two parameter
// tensors are fetched once per call instead of once per sample.
__global__ __launch_bounds__(kBlock) void ln_apply_batch4_kernel(const float4* __restrict__ x,
                                                                 const float4* __restrict__ gamma,
                                                                 const float4* __restrict__ beta,
                                                                 const float* __restrict__ stats,
                                                                 float4* __restrict__ y, long nimg, long e4) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e4; i += (long)gridDim.x * kBlock) {
        const float4 g = gamma[i], b = beta[i];
#pragma unroll 4
        for (long n = 0; n < nimg; ++n) {
            const float mean = stats[n * 2], rstd = stats[n * 2 + 1];
            const f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + n * e4 + i));
            const f32x4 r = {(v[0] - mean) * rstd * g.x + b.x, (v[1] - mean) * rstd * g.y + b.y,
                             (v[2] - mean) * rstd * g.z + b.z, (v[3] - mean) * rstd * g.w + b.w};
            __builtin_nontemporal_store(r, (f32x4*)(y + n * e4 + i));
        }
    }
}

// ---------------- LayerNormalization backward ----------------
// per-sample s1 = sum(dy*gamma), s2 = sum(dy*gamma*xhat)
__global__ __launch_bounds__(kBlock) void ln_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ stats,
                                                                double* __restrict__ part, long e, long bps) {
    __shared__ double red[4];
    const long img = blockIdx.y;
    const long b = blockIdx.x;
    const long chunk = ((e + bps - 1) / bps + 3) & ~3L;
    const long beg = b * chunk;
    long end = beg + chunk;
    if (end > e) end = e;
    const float mean = stats[img * 2], rstd = stats[img * 2 + 1];
    const float* g = dy + img * e;
    const float* src = x + img * e;
    double s1 = 0.0, s2 = 0.0;
    if ((e & 3) == 0) {
        for (long i = beg / 4 + threadIdx.x; i < end / 4; i += kBlock) {
            const float4 d = ((const float4*)g)[i], gm = ((const float4*)gamma)[i], v = ((const float4*)src)[i];
            const float g0 = d.x * gm.x, g1 = d.y * gm.y, g2 = d.z * gm.z, g3 = d.w * gm.w;
            s1 += (double)((g0 + g1) + (g2 + g3));
            s2 += (double)(g0 * ((v.x - mean) * rstd) + g1 * ((v.y - mean) * rstd)) +
                  (double)(g2 * ((v.z - mean) * rstd) + g3 * ((v.w - mean) * rstd));
        }
    } else {
        for (long i = beg + threadIdx.x; i < end; i += kBlock) {
            const float gg = g[i] * gamma[i];
            const float xh = (src[i] - mean) * rstd;
            s1 += (double)gg;
            s2 += (double)gg * (double)xh;
        }
    }
    const double t1 = block_sum256_d(s1, red);
    const double t2 = block_sum256_d(s2, red);
    if (threadIdx.x == 0) {
        part[(img * bps + b) * 2] = t1;
        part[(img * bps + b) * 2 + 1] = t2;
    }
}

__global__ __launch_bounds__(64) void ln_bwd_sums_kernel(const double* __restrict__ part, float* __restrict__ sums,
                                                         long e, long bps) {
    const long img = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (long b = threadIdx.x; b < bps; b += 64) {
        s += part[(img * bps + b) * 2];
        q += part[(img * bps + b) * 2 + 1];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x == 0) {
        sums[img * 2] = (float)(s / (double)e);
        sums[img * 2 + 1] = (float)(q / (double)e);
    }
}

// one thread per element position, loop over the batch: dx for every sample, dgamma/dbeta sums.
__global__ __launch_bounds__(kBlock) void ln_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ sums, float* __restrict__ dx,
                                                              int acc_x, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, long nimg, long e) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e; i += (long)gridDim.x * kBlock) {
        const float gm = gamma[i];
        float dg = 0.f, db = 0.f;
        for (long n = 0; n < nimg; ++n) {
            const float mean = stats[n * 2], rstd = stats[n * 2 + 1];
            const float m1 = sums[n * 2], m2 = sums[n * 2 + 1];
            const long o = n * e + i;
            const float g = dy[o];
            const float xh = (x[o] - mean) * rstd;
            dg += g * xh;
            db += g;
            if (dx) {
                const float v = rstd * (g * gm - m1 - xh * m2);
                dx[o] = acc_x ? dx[o] + v : v;
            }
        }
        if (dgamma) dgamma[i] += dg;
        if (dbeta) dbeta[i] += db;
    }
}

// the same with four element positions per thread (e a multiple of 4)
__global__ __launch_bounds__(kBlock) void ln_bwd_apply4_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                               const float4* __restrict__ gamma,
                                                               const float* __restrict__ stats,
                                                               const float* __restrict__ sums, float4* __restrict__ dx,
                                                               int acc_x, float4* __restrict__ dgamma,
                                                               float4* __restrict__ dbeta, long nimg, long e4) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e4; i += (long)gridDim.x * kBlock) {
        const float4 gm = gamma[i];
        float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
        for (long n = 0; n < nimg; ++n) {
            const float mean = stats[n * 2], rstd = stats[n * 2 + 1];
            const float m1 = sums[n * 2], m2 = sums[n * 2 + 1];
            const long o = n * e4 + i;
            const float4 g = dy[o], v = x[o];
            const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
            dg.x += g.x * xh.x; dg.y += g.y * xh.y; dg.z += g.z * xh.z; dg.w += g.w * xh.w;
            db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
            if (dx) {
                float4 r = make_float4(rstd * (g.x * gm.x - m1 - xh.x * m2), rstd * (g.y * gm.y - m1 - xh.y * m2),
                                       rstd * (g.z * gm.z - m1 - xh.z * m2), rstd * (g.w * gm.w - m1 - xh.w * m2));
                if (acc_x) {
                    const float4 p = dx[o];
                    r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w;
                }
                dx[o] = r;
            }
        }
        if (dgamma) {
            float4 p = dgamma[i];
            p.x += dg.x; p.y += dg.y; p.z += dg.z; p.w += dg.w;
            dgamma[i] = p;
        }
        if (dbeta) {
            float4 p = dbeta[i];
            p.x += db.x; p.y += db.y; p.z += db.z; p.w += db.w;
            dbeta[i] = p;
        }
    }
}

// ---------------- MinibatchStdDev ----------------
inline long mb_blocks(long e) {
    long b = (e + 2047) / 2048;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return b;
}

// grid (B, M): partial sums of sqrt(var_g + 1e-8) over a chunk of element positions.
__global__ __launch_bounds__(kBlock) void mb_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                            long e, int g, long m_groups, long bps) {
    __shared__ double red[4];
    const long m = blockIdx.y, b = blockIdx.x;
    const long chunk = (e + bps - 1) / bps;
    const long beg = b * chunk;
    long end = beg + chunk;
    if (end > e) end = e;
    double s = 0.0;
    for (long i = beg + threadIdx.x; i < end; i += kBlock) {
        float v[4];
        float mean = 0.f;
        for (int k = 0; k < g; ++k) { v[k] = x[(k * m_groups + m) * e + i]; mean += v[k]; }
        mean /= (float)g;
        float var = 0.f;
        for (int k = 0; k < g; ++k) { const float d = v[k] - mean; var += d * d; }
        var /= (float)g;
        s += (double)sqrtf(var + 1e-8f);
    }
    const double t = block_sum256_d(s, red);
    if (threadIdx.x == 0) part[m * bps + b] = t;
}

__global__ __launch_bounds__(64) void mb_stat_kernel(const double* __restrict__ part, float* __restrict__ stat, long e,
                                                     long bps) {
    const long m = blockIdx.x;
    double s = 0.0;
    for (long b = threadIdx.x; b < bps; b += 64) s += part[m * bps + b];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) stat[m] = (float)(s / (double)e);
}

// y[i, 0:e] = x[i];  y[i, e:e+row] = stat[i % M]
__global__ __launch_bounds__(kBlock) void mb_write_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                          float* __restrict__ y, long e, long row, long m_groups) {
    const long i = blockIdx.y;
    const float s = stat[i % m_groups];
    const float* src = x + i * e;
    float* dst = y + i * (e + row);
    for (long j = (long)blockIdx.x * kBlock + threadIdx.x; j < e + row; j += (long)gridDim.x * kBlock)
        dst[j] = j < e ? src[j] : s;
}

// dstat[m] = sum over members i (i % M == m) and the appended row of dy
__global__ __launch_bounds__(kBlock) void mb_dstat_kernel(const float* __restrict__ dy, float* __restrict__ dstat, long e,
                                                          long row, int g, long m_groups) {
    __shared__ double red[4];
    const long m = blockIdx.x;
    double s = 0.0;
    for (int k = 0; k < g; ++k) {
        const float* src = dy + (k * m_groups + m) * (e + row) + e;
        for (long j = threadIdx.x; j < row; j += kBlock) s += (double)src[j];
    }
    const double t = block_sum256_d(s, red);
    if (threadIdx.x == 0) dstat[m] = (float)t;
}

// dx[i,j] = dy[i,j] + dstat[m] * (x[i,j]-mean_j) / (g * std_j * e)
__global__ __launch_bounds__(kBlock) void mb_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ dstat, float* __restrict__ dx, long e,
                                                        long row, int g, long m_groups) {
    const long m = blockIdx.y;
    const float ds = dstat[m] / ((float)g * (float)e);
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < e; i += (long)gridDim.x * kBlock) {
        float v[4];
        float mean = 0.f;
        for (int k = 0; k < g; ++k) { v[k] = x[(k * m_groups + m) * e + i]; mean += v[k]; }
        mean /= (float)g;
        float var = 0.f;
        for (int k = 0; k < g; ++k) { const float d = v[k] - mean; var += d * d; }
        var /= (float)g;
        const float inv_std = 1.0f / sqrtf(var + 1e-8f);
        for (int k = 0; k < g; ++k) {
            const long smp = k * m_groups + m;
            dx[smp * e + i] = dy[smp * (e + row) + i] + ds * (v[k] - mean) * inv_std;
        }
    }
}

// ---- 16-byte forms (e and the appended row are multiples of 4 floats): two launches per direction ----
// grid (B, M): copies the group's members into y while summing sqrt(var + 1e-8) over the chunk.
__global__ __launch_bounds__(kBlock) void mb_fwd4_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                                         double* __restrict__ part, long e4, long ye4, int g,
                                                         long m_groups, long bps) {
    __shared__ double red[4];
    const long m = blockIdx.y, b = blockIdx.x;
    const long chunk = (e4 + bps - 1) / bps;
    const long beg = b * chunk;
    long end = beg + chunk;
    if (end > e4) end = e4;
    const float inv_g = 1.0f / (float)g;
    double s = 0.0;
    for (long i = beg + threadIdx.x; i < end; i += kBlock) {
        float4 v[4];
        float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < g) {
                v[k] = x[(k * m_groups + m) * e4 + i];
                mean.x += v[k].x; mean.y += v[k].y; mean.z += v[k].z; mean.w += v[k].w;
            }
        mean.x *= inv_g; mean.y *= inv_g; mean.z *= inv_g; mean.w *= inv_g;
        float4 var = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < g) {
                const float dx = v[k].x - mean.x, dy = v[k].y - mean.y, dz = v[k].z - mean.z, dw = v[k].w - mean.w;
                var.x += dx * dx; var.y += dy * dy; var.z += dz * dz; var.w += dw * dw;
                y[(k * m_groups + m) * ye4 + i] = v[k];
            }
        s += (double)((sqrtf(var.x * inv_g + 1e-8f) + sqrtf(var.y * inv_g + 1e-8f)) +
                      (sqrtf(var.z * inv_g + 1e-8f) + sqrtf(var.w * inv_g + 1e-8f)));
    }
    const double t = block_sum256_d(s, red);
    if (threadIdx.x == 0) part[m * bps + b] = t;
}

// grid (N): the statistic of the sample's group from the partial sums, written to the appended row
__global__ __launch_bounds__(kBlock) void mb_row4_kernel(const double* __restrict__ part, float* __restrict__ stat,
                                                         float4* __restrict__ y, long e, long e4, long ye4, long row4,
                                                         long m_groups, long bps) {
    __shared__ double red[4];
    __shared__ float sval;
    const long i = blockIdx.x, m = i % m_groups;
    double s = 0.0;
    for (long b = threadIdx.x; b < bps; b += kBlock) s += part[m * bps + b];
    const double t = block_sum256_d(s, red);
    if (threadIdx.x == 0) {
        sval = (float)(t / (double)e);
        if (i < m_groups) stat[m] = sval;
    }
    __syncthreads();
    const float v = sval;
    const float4 v4 = make_float4(v, v, v, v);
    float4* dst = y + i * ye4 + e4;
    for (long j = threadIdx.x; j < row4; j += kBlock) dst[j] = v4;
}

// grid (M, 4): partial sums of dy over the appended rows of the group's members
__global__ __launch_bounds__(kBlock) void mb_dstat4_kernel(const float4* __restrict__ dy, double* __restrict__ dpart,
                                                           long e4, long ye4, long row4, int g, long m_groups) {
    __shared__ double red[4];
    const long m = blockIdx.x;
    const int k = blockIdx.y;
    double s = 0.0;
    if (k < g) {
        const float4* src = dy + (k * m_groups + m) * ye4 + e4;
        for (long j = threadIdx.x; j < row4; j += kBlock) {
            const float4 v = src[j];
            s += (double)((v.x + v.y) + (v.z + v.w));
        }
    }
    const double t = block_sum256_d(s, red);
    if (threadIdx.x == 0) dpart[m * 4 + k] = t;
}

__global__ __launch_bounds__(kBlock) void mb_bwd4_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                         const double* __restrict__ dpart, float4* __restrict__ dx,
                                                         long e, long e4, long ye4, int g, long m_groups, long bps) {
    const long m = blockIdx.y, b = blockIdx.x;
    const long chunk = (e4 + bps - 1) / bps;
    const long beg = b * chunk;
    long end = beg + chunk;
    if (end > e4) end = e4;
    const float dstat = (float)((dpart[m * 4] + dpart[m * 4 + 1]) + (dpart[m * 4 + 2] + dpart[m * 4 + 3]));
    const float ds = dstat / ((float)g * (float)e);
    const float inv_g = 1.0f / (float)g;
    for (long i = beg + threadIdx.x; i < end; i += kBlock) {
        float4 v[4];
        float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < g) {
                v[k] = x[(k * m_groups + m) * e4 + i];
                mean.x += v[k].x; mean.y += v[k].y; mean.z += v[k].z; mean.w += v[k].w;
            }
        mean.x *= inv_g; mean.y *= inv_g; mean.z *= inv_g; mean.w *= inv_g;
        float4 var = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < g) {
                const float ax = v[k].x - mean.x, ay = v[k].y - mean.y, az = v[k].z - mean.z, aw = v[k].w - mean.w;
                var.x += ax * ax; var.y += ay * ay; var.z += az * az; var.w += aw * aw;
            }
        const float4 is = make_float4(ds / sqrtf(var.x * inv_g + 1e-8f), ds / sqrtf(var.y * inv_g + 1e-8f),
                                      ds / sqrtf(var.z * inv_g + 1e-8f), ds / sqrtf(var.w * inv_g + 1e-8f));
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < g) {
                const long smp = k * m_groups + m;
                const float4 d = dy[smp * ye4 + i];
                dx[smp * e4 + i] = make_float4(d.x + (v[k].x - mean.x) * is.x, d.y + (v[k].y - mean.y) * is.y,
                                               d.z + (v[k].z - mean.z) * is.z, d.w + (v[k].w - mean.w) * is.w);
            }
    }
}

// ---------------- Flatten + Dense(1) ----------------
// One block of 1024 threads per sample (the batch is small where this layer's latency matters: a sample is
// ~10^5 elements, so 256 threads would walk it in ~120 dependent steps); the 16 wave sums are added in order.
__global__ __launch_bounds__(1024) void dense1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ y, long e) {
    __shared__ double red[16];
    const long i = blockIdx.x;
    const float* src = x + i * e;
    double s = 0.0;
    const long e4 = (e & 3) == 0 ? e / 4 : 0;        // 16-byte loads need every sample's base aligned
    for (long j = threadIdx.x; j < e4; j += 1024) {
        const float4 a = ((const float4*)src)[j], c = ((const float4*)w)[j];
        s += (double)(a.x * c.x + a.y * c.y) + (double)(a.z * c.z + a.w * c.w);
    }
    for (long j = e4 * 4 + threadIdx.x; j < e; j += 1024) s += (double)src[j] * w[j];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q];
        y[i] = (float)(t + (double)(b ? b[0] : 0.f));
    }
}

__global__ __launch_bounds__(kBlock) void dense1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, long n,
                                                            long e) {
    for (long j = (long)blockIdx.x * kBlock + threadIdx.x; j < e; j += (long)gridDim.x * kBlock) {
        const float wj = w[j];
        float acc = 0.f;
        for (long i = 0; i < n; ++i) {
            const float g = dy[i];
            if (dx) dx[i * e + j] = g * wj;
            if (dw) acc += g * x[i * e + j];
        }
        if (dw) dw[j] += acc;
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (long i = 0; i < n; ++i) s += dy[i];
        db[0] += s;
    }
}

// ---------------- losses ----------------
__global__ __launch_bounds__(kBlock) void logit_sums_kernel(const float* __restrict__ fake, const float* __restrict__ real,
                                                            long n, float* __restrict__ sums, int accumulate) {
    __shared__ double red[4];
    double sf = 0.0, sr = 0.0;
    for (long i = threadIdx.x; i < n; i += kBlock) {
        if (fake) sf += (double)fake[i];
        if (real) sr += (double)real[i];
    }
    const double tf = block_sum256_d(sf, red);
    const double tr = block_sum256_d(sr, red);
    if (threadIdx.x == 0) {
        sums[0] = (float)(tf + (accumulate ? (double)sums[0] : 0.0));
        sums[1] = (float)(tr + (accumulate ? (double)sums[1] : 0.0));
    }
}

// out = ca[0]*a + cb[0]*b (b optional): combines gradients taken with unit logit seeds.  out may be
// the same buffer as a (the step scales in place), so neither carries a no-alias promise.
__global__ __launch_bounds__(kBlock) void combine_kernel(float* out, const float* a,
                                                         const float* __restrict__ b, const float* __restrict__ ca,
                                                         const float* __restrict__ cb, long n) {
    const float fa = ca[0], fb = b ? cb[0] : 0.f;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock)
        out[i] = b ? fa * a[i] + fb * b[i] : fa * a[i];
}

__global__ __launch_bounds__(kBlock) void wgan_loss_kernel(const float* __restrict__ sums, long n_global, int mode,
                                                           float* __restrict__ out4, float* __restrict__ dfake,
                                                           float* __restrict__ dreal, long n_local) {
    const float cu = sums[0] / (float)n_global;
    const float ci = sums[1] / (float)n_global;
    const float lamb = cu - ci;
    const float delta = fabsf(lamb);
    const float sgn = delta == 0.f ? 0.f : lamb / delta;  // divide_no_nan
    const float inv_n = 1.0f / (float)n_global;
    float loss, gf, gr;
    if (mode == 0) {
        const float sign = sgn + 0.0001f;
        const float sign2 = (sgn + 0.0000999f) * -1.0f;
        loss = sign * ci + sign2 * cu;
        gr = sign * inv_n;
        gf = sign2 * inv_n;
    } else {
        const float sign = sgn + 0.0001f;
        loss = sign * ci + delta;
        gr = 0.f;
        gf = sgn * inv_n;
    }
    if (threadIdx.x == 0 && out4) { out4[0] = loss; out4[1] = cu; out4[2] = ci; out4[3] = delta; }
    for (long i = threadIdx.x; i < n_local; i += kBlock) {
        if (dfake) dfake[i] = gf;
        if (dreal) dreal[i] = gr;
    }
}

// ---- per-clip checksums of a chunk of generated audio (sampling runs, utils.py:80-81 / 103-104) ----
// One read of the chunk: block (part, clip) reduces a contiguous share of the clip to (sum, sum of squares,
// peak) with float64 accumulators; a second kernel adds the parts of a clip in order.
constexpr int kChecksumParts = 512;

__global__ __launch_bounds__(256) void clip_checksum_partial_kernel(const f32x4* __restrict__ x, long quads_per_clip,
                                                                    double* __restrict__ partial) {
    __shared__ double red[3][4];
    const long clip = blockIdx.y;
    const long per = (quads_per_clip + gridDim.x - 1) / gridDim.x;
    const long q0 = (long)blockIdx.x * per, q1 = q0 + per < quads_per_clip ? q0 + per : quads_per_clip;
    const f32x4* src = x + clip * quads_per_clip;
    double s = 0.0, ss = 0.0;
    float pk = 0.f;
    for (long q = q0 + threadIdx.x; q < q1; q += 256) {
        const f32x4 v = __builtin_nontemporal_load(src + q);
        const float t = (v[0] + v[1]) + (v[2] + v[3]);
        const float t2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        s += (double)t;
        ss += (double)t2;
        pk = fmaxf(fmaxf(pk, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    double pd = (double)pk;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
        pd = fmax(pd, __shfl_down(pd, off, 64));
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wv] = s;
        red[1][wv] = ss;
        red[2][wv] = pd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* dst = partial + (clip * gridDim.x + blockIdx.x) * 3;
        dst[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        dst[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        dst[2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    }
}

__global__ __launch_bounds__(64) void clip_checksum_finish_kernel(const double* __restrict__ partial, int parts,
                                                                  double* __restrict__ out) {
    const long clip = blockIdx.x;
    const int lane = threadIdx.x;
    double s = 0.0, ss = 0.0, pk = 0.0;
    for (int p = lane; p < parts; p += 64) {
        const double* src = partial + (clip * parts + p) * 3;
        s += src[0];
        ss += src[1];
        pk = fmax(pk, src[2]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
        pk = fmax(pk, __shfl_down(pk, off, 64));
    }
    if (lane == 0) {
        out[clip * 3 + 0] = s;
        out[clip * 3 + 1] = ss;
        out[clip * 3 + 2] = pk;
    }
}

}  // namespace

extern "C" int64_t mgx_clip_checksum_workspace(int64_t nclips) {
    return nclips > 0 ? nclips * kChecksumParts * 3 * (int64_t)sizeof(double) : -1;
}

extern "C" int mgx_clip_checksum(const float* x, int64_t nclips, int64_t elems, double* out, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace || nclips <= 0 || nclips > 65535 || elems <= 0 || (elems & 3)) return MGX_ERR_ARG;
    if (((uintptr_t)x & 15) != 0) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_clip_checksum_workspace(nclips)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long quads = elems / 4;
    const int parts = (int)(quads < kChecksumParts ? quads : kChecksumParts);
    hipLaunchKernelGGL(clip_checksum_partial_kernel, dim3(parts, (unsigned)nclips), dim3(256), 0, st, (const f32x4*)x, quads,
                       (double*)workspace);
    hipLaunchKernelGGL(clip_checksum_finish_kernel, dim3((unsigned)nclips), dim3(64), 0, st, (const double*)workspace, parts,
                       out);
    return mgx_launch_status();
}

// second pass of the forward: per-sample grid, or the batch walk when gamma / beta are larger than an L2
static void launch_ln_apply(const float* x, const float* gamma, const float* beta, const float* stats, float* y,
                            int64_t nimg, int64_t e, hipStream_t st) {
    if ((e & 3) == 0 && e >= (1L << 20) && nimg > 1) {
        const long e4 = e / 4;
        long g4 = (e4 + kBlock - 1) / kBlock;
        if (g4 > 8192) g4 = 8192;
        hipLaunchKernelGGL(ln_apply_batch4_kernel, dim3((unsigned)g4), dim3(kBlock), 0, st, (const float4*)x,
                           (const float4*)gamma, (const float4*)beta, stats, (float4*)y, (long)nimg, e4);
        return;
    }
    long gx = (e / 2 + kBlock - 1) / kBlock;
    const long cap = 4096 / nimg > 0 ? 4096 / nimg : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(ln_apply_kernel, dim3((unsigned)gx, (unsigned)nimg), dim3(kBlock), 0, st, x, gamma, beta, stats, y,
                       (long)e);
}

extern "C" int64_t mgx_layernorm_workspace(int64_t nimg, int64_t e) {
    if (nimg <= 0 || e <= 0) return -1;
    const long bps = ln_blocks_per_sample(nimg, e);
    return nimg * bps * 2 * (int64_t)sizeof(double) + nimg * 2 * (int64_t)sizeof(float);
}

extern "C" int mgx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                 int64_t nimg, int64_t e, float eps, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    if (!x || !gamma || !beta || !y || !stats || !workspace || nimg <= 0 || e <= 0 || nimg > 65535) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_layernorm_workspace(nimg, e)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long bps = ln_blocks_per_sample(nimg, e);
    double* part = (double*)workspace;
    hipLaunchKernelGGL(ln_partial_kernel, dim3((unsigned)bps, (unsigned)nimg), dim3(kBlock), 0, st, x, part, (long)e, bps);
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)nimg), dim3(64), 0, st, part, stats, (long)e, bps, eps);
    launch_ln_apply(x, gamma, beta, stats, y, nimg, e, st);
    return mgx_launch_status();
}

// ------------------------------------------------------------------------------------------
// mgx_layernorm_fwd for callers that cannot allocate (code under HIP graph capture, tight serving
// loops): the per-(sample, block) partial sums live in a file-scope device array instead of a
// caller workspace.  Every GPU the library runs on holds its own copy of the array (32 KiB) until the
// process ends; the host keeps the copy's device address per GPU (the array's name in host code is
// only a handle for hipGetSymbolAddress, never a device pointer).
// ------------------------------------------------------------------------------------------
constexpr int kLnStaticPairs = 2048;                 // (sample, block) pairs = 2 doubles each
static __device__ __align__(256) double ln_static_scratch[2 * kLnStaticPairs];

namespace {
constexpr int kMaxDevices = 64;
double* g_ln_scratch[kMaxDevices];                   // device address of the array on each GPU, once resolved

int ln_static_address(double** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MGX_ERR_LAUNCH;
    double* p = __atomic_load_n(&g_ln_scratch[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        void* addr = nullptr;
        if (hipGetSymbolAddress(&addr, HIP_SYMBOL(ln_static_scratch)) != hipSuccess || !addr) return MGX_ERR_LAUNCH;
        p = (double*)addr;
        __atomic_store_n(&g_ln_scratch[dev], p, __ATOMIC_RELEASE);   // racing first calls resolve the same address
    }
    *out = p;
    return MGX_OK;
}
}  // namespace

extern "C" int64_t mgx_layernorm_fwd_static_max_samples(void) { return kLnStaticPairs; }

extern "C" int mgx_layernorm_fwd_static(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                        int64_t nimg, int64_t e, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !stats || nimg <= 0 || e <= 0) return MGX_ERR_ARG;
    if (nimg > kLnStaticPairs) return MGX_ERR_WORKSPACE;        // more samples than the static scratch has pairs for
    const long bps = ln_blocks_per_sample(nimg, e);
    if (nimg * bps > kLnStaticPairs) return MGX_ERR_WORKSPACE;
    double* part = nullptr;
    const int rc = ln_static_address(&part);
    if (rc != MGX_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // every pair the second kernel reads is written by the first one of the same call: nothing to clear
    hipLaunchKernelGGL(ln_partial_kernel, dim3((unsigned)bps, (unsigned)nimg), dim3(kBlock), 0, st, x, part, (long)e, bps);
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)nimg), dim3(64), 0, st, part, stats, (long)e, bps, eps);
    launch_ln_apply(x, gamma, beta, stats, y, nimg, e, st);
    return mgx_launch_status();
}

extern "C" int mgx_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats, float* dx,
                                 int acc_x, float* dgamma, float* dbeta, int64_t nimg, int64_t e, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    if (!dy || !x || !gamma || !stats || !workspace || nimg <= 0 || e <= 0 || nimg > 65535) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_layernorm_workspace(nimg, e)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long bps = ln_blocks_per_sample(nimg, e);
    double* part = (double*)workspace;
    float* sums = (float*)(part + nimg * bps * 2);
    hipLaunchKernelGGL(ln_bwd_partial_kernel, dim3((unsigned)bps, (unsigned)nimg), dim3(kBlock), 0, st, dy, x, gamma,
                       stats, part, (long)e, bps);
    hipLaunchKernelGGL(ln_bwd_sums_kernel, dim3((unsigned)nimg), dim3(64), 0, st, part, sums, (long)e, bps);
    if ((e & 3) == 0) {
        long g4 = (e / 4 + kBlock - 1) / kBlock;
        if (g4 > 8192) g4 = 8192;
        hipLaunchKernelGGL(ln_bwd_apply4_kernel, dim3((unsigned)g4), dim3(kBlock), 0, st, (const float4*)dy,
                           (const float4*)x, (const float4*)gamma, stats, sums, (float4*)dx, acc_x, (float4*)dgamma,
                           (float4*)dbeta, (long)nimg, (long)(e / 4));
        return mgx_launch_status();
    }
    long gx = (e + kBlock - 1) / kBlock;
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3((unsigned)gx), dim3(kBlock), 0, st, dy, x, gamma, stats, sums, dx, acc_x,
                       dgamma, dbeta, (long)nimg, (long)e);
    return mgx_launch_status();
}

extern "C" int64_t mgx_mbstd_workspace(int64_t n, int64_t e) {
    if (n <= 0 || e <= 0) return -1;
    const long g = n < 4 ? n : 4;
    const long m = n / g;
    const long parts = mb_blocks(e) > 4 ? mb_blocks(e) : 4;       // forward: blocks per group; backward: 4 row sums
    return m * parts * (int64_t)sizeof(double) + m * (int64_t)sizeof(float);
}

extern "C" int mgx_mbstd_fwd(const float* x, float* y, float* stat, int64_t n, int h, int w, int c, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    if (!x || !y || !stat || !workspace || n <= 0 || h <= 0 || w <= 0 || c <= 0) return MGX_ERR_ARG;
    const int g = n < 4 ? (int)n : 4;
    if (n % g != 0) return MGX_ERR_ARG;  // the reference's reshape requires it (models.py:176)
    const long m = n / g;
    const long e = (long)h * w * c, row = (long)w * c;
    if (m > 65535 || n > 65535) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_mbstd_workspace(n, e)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long bps = mb_blocks(e);
    double* part = (double*)workspace;
    if ((e & 3) == 0 && (row & 3) == 0) {
        hipLaunchKernelGGL(mb_fwd4_kernel, dim3((unsigned)bps, (unsigned)m), dim3(kBlock), 0, st, (const float4*)x,
                           (float4*)y, part, e / 4, (e + row) / 4, g, m, bps);
        hipLaunchKernelGGL(mb_row4_kernel, dim3((unsigned)n), dim3(kBlock), 0, st, part, stat, (float4*)y, e, e / 4,
                           (e + row) / 4, row / 4, m, bps);
        return mgx_launch_status();
    }
    hipLaunchKernelGGL(mb_partial_kernel, dim3((unsigned)bps, (unsigned)m), dim3(kBlock), 0, st, x, part, e, g, m, bps);
    hipLaunchKernelGGL(mb_stat_kernel, dim3((unsigned)m), dim3(64), 0, st, part, stat, e, bps);
    long gx = (e + row + kBlock - 1) / kBlock;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(mb_write_kernel, dim3((unsigned)gx, (unsigned)n), dim3(kBlock), 0, st, x, stat, y, e, row, m);
    return mgx_launch_status();
}

extern "C" int mgx_mbstd_bwd(const float* dy, const float* x, float* dx, int64_t n, int h, int w, int c, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    if (!dy || !x || !dx || !workspace || n <= 0 || h <= 0 || w <= 0 || c <= 0) return MGX_ERR_ARG;
    const int g = n < 4 ? (int)n : 4;
    if (n % g != 0) return MGX_ERR_ARG;
    const long m = n / g;
    const long e = (long)h * w * c, row = (long)w * c;
    if (m > 65535) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_mbstd_workspace(n, e)) return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long bps = mb_blocks(e);
    if ((e & 3) == 0 && (row & 3) == 0) {
        double* dpart = (double*)workspace;
        hipLaunchKernelGGL(mb_dstat4_kernel, dim3((unsigned)m, 4), dim3(kBlock), 0, st, (const float4*)dy, dpart, e / 4,
                           (e + row) / 4, row / 4, g, m);
        hipLaunchKernelGGL(mb_bwd4_kernel, dim3((unsigned)bps, (unsigned)m), dim3(kBlock), 0, st, (const float4*)dy,
                           (const float4*)x, dpart, (float4*)dx, e, e / 4, (e + row) / 4, g, m, bps);
        return mgx_launch_status();
    }
    float* dstat = (float*)((double*)workspace + m * (bps > 4 ? bps : 4));
    hipLaunchKernelGGL(mb_dstat_kernel, dim3((unsigned)m), dim3(kBlock), 0, st, dy, dstat, e, row, g, m);
    hipLaunchKernelGGL(mb_bwd_kernel, dim3((unsigned)bps, (unsigned)m), dim3(kBlock), 0, st, dy, x, dstat, dx, e, row, g, m);
    return mgx_launch_status();
}

extern "C" int mgx_dense1_fwd(const float* x, const float* w, const float* b, float* y, int64_t n, int64_t e,
                              void* stream) {
    if (!x || !w || !y || n <= 0 || e <= 0 || n > 0x7fffffffL) return MGX_ERR_ARG;
    hipLaunchKernelGGL(dense1_fwd_kernel, dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, x, w, b, y, (long)e);
    return mgx_launch_status();
}

extern "C" int mgx_dense1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                              int64_t n, int64_t e, void* stream) {
    if (!dy || !x || !w || n <= 0 || e <= 0) return MGX_ERR_ARG;
    long gx = (e + kBlock - 1) / kBlock;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(dense1_bwd_kernel, dim3((unsigned)gx), dim3(kBlock), 0, (hipStream_t)stream, dy, x, w, dx, dw, db,
                       (long)n, (long)e);
    return mgx_launch_status();
}

extern "C" int mgx_logit_sums(const float* fake, const float* real, int64_t n, float* sums, int accumulate,
                              void* stream) {
    if (!sums || n <= 0 || (!fake && !real)) return MGX_ERR_ARG;
    hipLaunchKernelGGL(logit_sums_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, fake, real, (long)n, sums,
                       accumulate);
    return mgx_launch_status();
}

extern "C" int mgx_combine_grads(float* out, const float* a, const float* b, const float* ca, const float* cb,
                                 int64_t n, void* stream) {
    if (!out || !a || !ca || n <= 0 || (b && !cb)) return MGX_ERR_ARG;
    long g = (n + kBlock - 1) / kBlock;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream, out, a, b, ca, cb,
                       (long)n);
    return mgx_launch_status();
}

extern "C" int mgx_wgan_loss(const float* sums, int64_t n_global, int mode, float* out4, float* dfake, float* dreal,
                             int64_t n_local, void* stream) {
    if (!sums || n_global <= 0 || (mode != 0 && mode != 1) || n_local < 0) return MGX_ERR_ARG;
    hipLaunchKernelGGL(wgan_loss_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, sums, (long)n_global, mode, out4,
                       dfake, dreal, (long)n_local);
    return mgx_launch_status();
}
