// HBM-bound element-wise layers: UpSampling2D, AveragePooling2D, Add, WeightedSum,
// SoftRectifier/StaticOptTanh, Adam.  One float4 (16 B) per lane where the layout allows it,
// grid-stride loops capped at 2048 workgroups of 256 threads.
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int kBlock = 256;
constexpr long kMaxGrid = 2048;

inline unsigned grid_for(long work_items) {
    long g = (work_items + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------- UpSampling2D nearest x2 ----------------
// one work item per INPUT pixel: read c floats, write them to the 2x2 output block.
__global__ __launch_bounds__(kBlock) void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              long nimg, int h, int w, int c) {
    const long total = nimg * h * w;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long row = i / w;  // img*h + hi
        const int wi = (int)(i - row * w);
        const float* src = x + i * c;
        float* d0 = y + ((row * 2) * (2L * w) + 2L * wi) * c;
        float* d1 = d0 + 2L * w * c;
        if (c == 2) {
            const float2 v = *(const float2*)src;
            const float4 vv = make_float4(v.x, v.y, v.x, v.y);
            *(float4*)d0 = vv;
            *(float4*)d1 = vv;
        } else {
            for (int k = 0; k < c; ++k) {
                const float v = src[k];
                d0[k] = v; d0[c + k] = v; d1[k] = v; d1[c + k] = v;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                              long nimg, int h, int w, int c, int accumulate) {
    const long total = nimg * h * w;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long row = i / w;
        const int wi = (int)(i - row * w);
        const float* s0 = dy + ((row * 2) * (2L * w) + 2L * wi) * c;
        const float* s1 = s0 + 2L * w * c;
        float* dst = dx + i * c;
        if (c == 2) {
            const float4 a = *(const float4*)s0, b = *(const float4*)s1;
            float2 v = make_float2((a.x + a.z) + (b.x + b.z), (a.y + a.w) + (b.y + b.w));
            if (accumulate) { const float2 o = *(float2*)dst; v.x += o.x; v.y += o.y; }
            *(float2*)dst = v;
        } else {
            for (int k = 0; k < c; ++k) {
                float v = (s0[k] + s0[c + k]) + (s1[k] + s1[c + k]);
                if (accumulate) v += dst[k];
                dst[k] = v;
            }
        }
    }
}

// ---------------- AveragePooling2D 2x2/2 ----------------
// one work item per OUTPUT pixel; h, w are input sizes (even).
__global__ __launch_bounds__(kBlock) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             long nimg, int h, int w, int c) {
    const int ho = h / 2, wo = w / 2;
    const long total = nimg * ho * wo;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long row = i / wo;  // img*ho + oh
        const int ow = (int)(i - row * wo);
        const float* s0 = x + ((row * 2) * (long)w + 2L * ow) * c;
        const float* s1 = s0 + (long)w * c;
        float* dst = y + i * c;
        if (c == 2) {
            const float4 a = *(const float4*)s0, b = *(const float4*)s1;
            *(float2*)dst = make_float2(((a.x + a.z) + (b.x + b.z)) * 0.25f, ((a.y + a.w) + (b.y + b.w)) * 0.25f);
        } else {
            for (int k = 0; k < c; ++k) dst[k] = ((s0[k] + s0[c + k]) + (s1[k] + s1[c + k])) * 0.25f;
        }
    }
}

__global__ __launch_bounds__(kBlock) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                             long nimg, int h, int w, int c, int accumulate) {
    const int ho = h / 2, wo = w / 2;
    const long total = nimg * ho * wo;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long row = i / wo;
        const int ow = (int)(i - row * wo);
        const float* src = dy + i * c;
        float* d0 = dx + ((row * 2) * (long)w + 2L * ow) * c;
        float* d1 = d0 + (long)w * c;
        for (int k = 0; k < c; ++k) {
            const float v = src[k] * 0.25f;
            if (accumulate) { d0[k] += v; d0[c + k] += v; d1[k] += v; d1[c + k] += v; }
            else { d0[k] = v; d0[c + k] = v; d1[k] = v; d1[c + k] = v; }
        }
    }
}

// ---------------- Add of up to 8 tensors ----------------
struct AddPtrs { const float* p[8]; };

__global__ __launch_bounds__(kBlock) void add_n_kernel(AddPtrs in, int nin, float* __restrict__ y, long n4, long n) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float4 s = ((const float4*)in.p[0])[i];
        for (int k = 1; k < nin; ++k) {
            const float4 v = ((const float4*)in.p[k])[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        ((float4*)y)[i] = s;
    }
    // tail (n % 4 elements)
    const long tail0 = n4 * 4;
    const long t = tail0 + (long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) {
        float s = in.p[0][t];
        for (int k = 1; k < nin; ++k) s += in.p[k][t];
        y[t] = s;
    }
}

__global__ __launch_bounds__(kBlock) void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                             long n4, long n) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float4 a = ((float4*)dst)[i];
        const float4 b = ((const float4*)src)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        ((float4*)dst)[i] = a;
    }
    const long t = n4 * 4 + (long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) dst[t] += src[t];
}

// ---------------- WeightedSum ----------------
__global__ __launch_bounds__(kBlock) void wsum_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ alpha, float* __restrict__ y,
                                                          long n4, long n) {
    const float al = alpha[0], om = 1.0f - al;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        const float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i];
        ((float4*)y)[i] = make_float4(mgx_blend(om, u.x, al, v.x), mgx_blend(om, u.y, al, v.y), mgx_blend(om, u.z, al, v.z),
                                      mgx_blend(om, u.w, al, v.w));
    }
    const long t = n4 * 4 + (long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) y[t] = mgx_blend(om, a[t], al, b[t]);
}

// da (+)= (1-alpha) dy ; db (+)= alpha dy ; block partial of sum dy*(b-a) -> part[blockIdx.x]
__global__ __launch_bounds__(kBlock) void wsum_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                          const float* __restrict__ b, const float* __restrict__ alpha,
                                                          float* __restrict__ da, int acc_a, float* __restrict__ db,
                                                          int acc_b, double* __restrict__ part, long n) {
    __shared__ double red[4];
    const float al = alpha[0], om = 1.0f - al;
    double s = 0.0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float g = dy[i];
        if (da) da[i] = acc_a ? da[i] + om * g : om * g;
        if (db) db[i] = acc_b ? db[i] + al * g : al * g;
        if (part) s += (double)g * (double)(b[i] - a[i]);
    }
    if (part) {
        const double tot = block_sum256_d(s, red);
        if (threadIdx.x == 0) part[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(64) void finish_sum_kernel(const double* __restrict__ part, int nparts,
                                                        float* __restrict__ out) {
    // single wave, fixed order
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] += (float)s;
}

// ---------------- SoftRectifier / StaticOptTanh ----------------
__global__ __launch_bounds__(kBlock) void softrect_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                              float alpha) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float v = x[i];
        float t, s2;
        tanh_sech2(v, t, s2);
        y[i] = t + v / alpha;
    }
}

__global__ __launch_bounds__(kBlock) void softrect_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ dx, int accumulate, long n,
                                                              float alpha) {
    const float inv = 1.0f / alpha;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        float t, s2;
        tanh_sech2(x[i], t, s2);
        const float g = dy[i] * (s2 + inv);
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// ---------------- Adam, beta_1 = 0 ----------------
__global__ __launch_bounds__(1024) void adam_multi_kernel(float* const* __restrict__ p, const int64_t* __restrict__ n,
                                                          const int64_t* __restrict__ off, const float* __restrict__ g,
                                                          float* __restrict__ v, const int32_t* __restrict__ blk_tensor,
                                                          const int64_t* __restrict__ blk_start, float lr_host,
                                                          const float* __restrict__ lr_dev, float beta2, float eps) {
    const float lr_t = lr_dev ? lr_dev[0] : lr_host;
    const int t = blk_tensor[blockIdx.x];
    const int64_t start = blk_start[blockIdx.x];
    const int64_t cnt = n[t];
    float* pp = p[t];
    const float* gg = g + off[t];
    float* vv = v + off[t];
    const float omb = 1.0f - beta2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = start + (int64_t)k * 1024 + threadIdx.x;
        if (i < cnt) {
            const float gi = gg[i];
            const float vi = beta2 * vv[i] + omb * gi * gi;
            vv[i] = vi;
            pp[i] -= lr_t * gi / (sqrtf(vi) + eps);
        }
    }
}

// Adam with a first moment (keras' default beta_1 = 0.9, the evaluator's optimiser, evaluacion.py:160)
__global__ __launch_bounds__(1024) void adam_multi_m_kernel(float* const* __restrict__ p, const int64_t* __restrict__ n,
                                                            const int64_t* __restrict__ off, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            const int32_t* __restrict__ blk_tensor,
                                                            const int64_t* __restrict__ blk_start, float lr_t, float beta1,
                                                            float beta2, float eps) {
    const int t = blk_tensor[blockIdx.x];
    const int64_t start = blk_start[blockIdx.x];
    const int64_t cnt = n[t];
    float* pp = p[t];
    const float* gg = g + off[t];
    float* mm = m + off[t];
    float* vv = v + off[t];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = start + (int64_t)k * 1024 + threadIdx.x;
        if (i < cnt) {
            const float gi = gg[i];
            const float mi = beta1 * mm[i] + (1.0f - beta1) * gi;
            const float vi = beta2 * vv[i] + (1.0f - beta2) * gi * gi;
            mm[i] = mi;
            vv[i] = vi;
            pp[i] -= lr_t * mi / (sqrtf(vi) + eps);
        }
    }
}

// Softmax + categorical cross-entropy of one batch (evaluacion.py:160).  tf.keras evaluates
// categorical_crossentropy on the output of a Softmax op from that op's logits
// (softmax_cross_entropy_with_logits), so there is no probability clipping: per row probabilities,
// loss_rows[r] = -sum_c y (x - max - log sum exp) and the gradient of the batch-mean loss with
// respect to the logits, (p - y) / rows.
__global__ __launch_bounds__(64) void softmax_xent_kernel(const float* __restrict__ x, const float* __restrict__ onehot,
                                                          float* __restrict__ prob, float* __restrict__ loss_rows,
                                                          float* __restrict__ dx, long rows, int cols) {
    const int lane = threadIdx.x;
    const float inv_rows = 1.0f / (float)rows;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* xr = x + r * cols;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += __expf(xr[c] - mx);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum, lse = mx + logf(sum);
        float loss = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float pc = __expf(xr[c] - mx) * inv, yc = onehot[r * cols + c];
            prob[r * cols + c] = pc;
            if (dx) dx[r * cols + c] = (pc - yc) * inv_rows;
            loss -= yc * (xr[c] - lse);
        }
        for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o);
        if (lane == 0) loss_rows[r] = loss;
    }
}

// Row softmax (the Dense(9, activation='softmax') head of the evaluator, evaluacion.py:97): one
// wavefront per row, three passes over the row (max, sum of exp, write) with wave reductions.
__global__ __launch_bounds__(64) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                          int cols) {
    const int lane = threadIdx.x;
    for (long r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* xr = x + r * cols;
        float m = -INFINITY;
        for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += __expf(xr[c] - m);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.f / sum;
        for (int c = lane; c < cols; c += 64) y[r * cols + c] = __expf(xr[c] - m) * inv;
    }
}

}  // namespace

extern "C" int mgx_upsample2x_fwd(const float* x, float* y, int64_t nimg, int h, int w, int c, void* stream) {
    if (!x || !y || nimg <= 0 || h <= 0 || w <= 0 || c <= 0) return MGX_ERR_ARG;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for(nimg * h * w)), dim3(kBlock), 0, (hipStream_t)stream, x, y,
                       (long)nimg, h, w, c);
    return mgx_launch_status();
}

extern "C" int mgx_upsample2x_bwd(const float* dy, float* dx, int64_t nimg, int h, int w, int c, int accumulate,
                                  void* stream) {
    if (!dy || !dx || nimg <= 0 || h <= 0 || w <= 0 || c <= 0) return MGX_ERR_ARG;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(nimg * h * w)), dim3(kBlock), 0, (hipStream_t)stream, dy, dx,
                       (long)nimg, h, w, c, accumulate);
    return mgx_launch_status();
}

extern "C" int mgx_avgpool2x_fwd(const float* x, float* y, int64_t nimg, int h, int w, int c, void* stream) {
    if (!x || !y || nimg <= 0 || h <= 0 || w <= 0 || c <= 0 || (h & 1) || (w & 1)) return MGX_ERR_ARG;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(nimg * (h / 2) * (w / 2))), dim3(kBlock), 0,
                       (hipStream_t)stream, x, y, (long)nimg, h, w, c);
    return mgx_launch_status();
}

extern "C" int mgx_avgpool2x_bwd(const float* dy, float* dx, int64_t nimg, int h, int w, int c, int accumulate,
                                 void* stream) {
    if (!dy || !dx || nimg <= 0 || h <= 0 || w <= 0 || c <= 0 || (h & 1) || (w & 1)) return MGX_ERR_ARG;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(nimg * (h / 2) * (w / 2))), dim3(kBlock), 0,
                       (hipStream_t)stream, dy, dx, (long)nimg, h, w, c, accumulate);
    return mgx_launch_status();
}

extern "C" int mgx_add_n(const float* const* xs, int nin, float* y, int64_t n, void* stream) {
    if (!xs || !y || nin < 1 || nin > 8 || n <= 0) return MGX_ERR_ARG;
    AddPtrs in;
    for (int i = 0; i < 8; ++i) in.p[i] = i < nin ? xs[i] : nullptr;
    for (int i = 0; i < nin; ++i)
        if (!in.p[i]) return MGX_ERR_ARG;
    const long n4 = n / 4;
    hipLaunchKernelGGL(add_n_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kBlock), 0, (hipStream_t)stream, in, nin, y,
                       n4, (long)n);
    return mgx_launch_status();
}

// Micro-batch of whole MinibatchStdDev groups: output sample a * cnt + b is input sample a * m + m0 + b
// (a < g, b < cnt): g contiguous runs of cnt samples.  blockIdx.y = output sample.
__global__ __launch_bounds__(kBlock) void gather_groups_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                               long row4, int m, int m0, int cnt) {
    const int o = blockIdx.y, a = o / cnt, b = o - a * cnt;
    const float4* s = src + ((long)a * m + m0 + b) * row4;
    float4* d = dst + (long)o * row4;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < row4; i += (long)gridDim.x * kBlock)
        d[i] = s[i];
}

extern "C" int mgx_gather_groups(const float* src, float* dst, int64_t row_floats, int g, int m, int m0, int cnt,
                                 void* stream) {
    if (!src || !dst || row_floats <= 0 || row_floats % 4 || g <= 0 || cnt <= 0 || m0 < 0 || m0 + cnt > m)
        return MGX_ERR_ARG;
    if ((long)g * cnt > 65535) return MGX_ERR_ARG;
    const long row4 = row_floats / 4;
    long gx = (row4 + kBlock - 1) / kBlock;
    const long cap = (4096 + (long)g * cnt - 1) / ((long)g * cnt);      // ~4096 blocks in all
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gather_groups_kernel, dim3((unsigned)gx, (unsigned)(g * cnt)), dim3(kBlock), 0,
                       (hipStream_t)stream, (const float4*)src, (float4*)dst, row4, m, m0, cnt);
    return mgx_launch_status();
}

extern "C" int mgx_add_inplace(float* dst, const float* src, int64_t n, void* stream) {
    if (!dst || !src || n <= 0) return MGX_ERR_ARG;
    const long n4 = n / 4;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kBlock), 0, (hipStream_t)stream, dst,
                       src, n4, (long)n);
    return mgx_launch_status();
}

extern "C" int mgx_weighted_sum_fwd(const float* a, const float* b, const float* alpha, float* y, int64_t n,
                                    void* stream) {
    if (!a || !b || !alpha || !y || n <= 0) return MGX_ERR_ARG;
    const long n4 = n / 4;
    hipLaunchKernelGGL(wsum_fwd_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kBlock), 0, (hipStream_t)stream, a, b,
                       alpha, y, n4, (long)n);
    return mgx_launch_status();
}

extern "C" int64_t mgx_reduce_workspace(void) { return kMaxGrid * (int64_t)sizeof(double); }

extern "C" int mgx_weighted_sum_bwd(const float* dy, const float* a, const float* b, const float* alpha, float* da,
                                    int acc_a, float* db, int acc_b, float* dalpha, void* workspace,
                                    int64_t workspace_bytes, int64_t n, void* stream) {
    if (!dy || !alpha || n <= 0) return MGX_ERR_ARG;
    if (dalpha && (!a || !b || !workspace)) return MGX_ERR_ARG;
    if (dalpha && workspace_bytes < mgx_reduce_workspace()) return MGX_ERR_WORKSPACE;
    const unsigned grid = grid_for(n);
    double* part = dalpha ? (double*)workspace : nullptr;
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, dy, a, b, alpha, da, acc_a, db,
                       acc_b, part, (long)n);
    if (dalpha)
        hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, (int)grid, dalpha);
    return mgx_launch_status();
}

extern "C" int mgx_adam_multi_m(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* m,
                                float* v, const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks,
                                float lr_t, float beta1, float beta2, float eps, void* stream) {
    if (!p || !n || !off || !g || !m || !v || !blk_tensor || !blk_start || nblocks <= 0 || nblocks > 0x7fffffffL)
        return MGX_ERR_ARG;
    hipLaunchKernelGGL(adam_multi_m_kernel, dim3((unsigned)nblocks), dim3(1024), 0, (hipStream_t)stream, p, n, off, g,
                       m, v, blk_tensor, blk_start, lr_t, beta1, beta2, eps);
    return mgx_launch_status();
}

extern "C" int mgx_softmax_xent(const float* x, const float* onehot, float* prob, float* loss_rows, float* dx,
                                int64_t rows, int cols, void* stream) {
    if (!x || !onehot || !prob || !loss_rows || rows <= 0 || cols <= 0) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, x, onehot, prob, loss_rows, dx,
                       (long)rows, cols);
    return mgx_launch_status();
}

extern "C" int mgx_softmax_rows(const float* x, float* y, int64_t rows, int cols, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)(rows < 65536 ? rows : 65536);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, x, y, (long)rows, cols);
    return mgx_launch_status();
}

extern "C" int mgx_softrect_fwd(const float* x, float* y, int64_t n, float alpha, void* stream) {
    if (!x || !y || n <= 0 || alpha == 0.f) return MGX_ERR_ARG;
    hipLaunchKernelGGL(softrect_fwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, (long)n,
                       alpha);
    return mgx_launch_status();
}

extern "C" int mgx_softrect_bwd(const float* dy, const float* x, float* dx, int accumulate, int64_t n, float alpha,
                                void* stream) {
    if (!dy || !x || !dx || n <= 0 || alpha == 0.f) return MGX_ERR_ARG;
    hipLaunchKernelGGL(softrect_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, dy, x, dx,
                       accumulate, (long)n, alpha);
    return mgx_launch_status();
}

extern "C" int mgx_adam_multi(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* v,
                              const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks, float lr_t,
                              float beta2, float eps, void* stream) {
    if (!p || !n || !off || !g || !v || !blk_tensor || !blk_start || nblocks <= 0 || nblocks > 0x7fffffffL)
        return MGX_ERR_ARG;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nblocks), dim3(1024), 0, (hipStream_t)stream, p, n, off, g, v,
                       blk_tensor, blk_start, lr_t, (const float*)nullptr, beta2, eps);
    return mgx_launch_status();
}

extern "C" int mgx_adam_multi_devlr(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* v,
                                    const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks,
                                    const float* lr_t, float beta2, float eps, void* stream) {
    if (!p || !n || !off || !g || !v || !blk_tensor || !blk_start || !lr_t || nblocks <= 0 || nblocks > 0x7fffffffL)
        return MGX_ERR_ARG;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nblocks), dim3(1024), 0, (hipStream_t)stream, p, n, off, g, v,
                       blk_tensor, blk_start, 0.f, lr_t, beta2, eps);
    return mgx_launch_status();
}
