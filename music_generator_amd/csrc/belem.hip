// HBM-bound kernels of the bf16 discriminator path: float32 <-> bf16 casts, the from-audio 1x1
// convolution (2 float32 input channels -> 128 bf16 channels, trainer/models.py:256) with its weight
// gradient and data-gradient, and WeightedSum (models.py:30-35) on bf16 tensors.  16 bytes per lane on
// the wide side; cross-workgroup sums are two-stage and fixed-order (no float atomics).
#include "bf16.h"
#include "../../include/mgx.h"

namespace {

constexpr int kBlock = 256;
constexpr long kMaxGrid = 2048;

inline unsigned grid_for(long items) {
    long g = (items + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__global__ __launch_bounds__(kBlock) void cast_f2b_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n8,
                                                          long n) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (long)gridDim.x * kBlock) {
        const f32x4 a = ((const f32x4*)x)[2 * i], b = ((const f32x4*)x)[2 * i + 1];
        const u32x4 o = {bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
        ((u32x4*)y)[i] = o;
    }
    const long t = n8 * 8 + (long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) y[t] = (bf16_t)x[t];
}

__global__ __launch_bounds__(kBlock) void cast_b2f_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n8,
                                                          long n) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (long)gridDim.x * kBlock) {
        const u32x4 v = ((const u32x4*)x)[i];
        const f32x4 a = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
        const f32x4 b = {bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
        ((f32x4*)y)[2 * i] = a;
        ((f32x4*)y)[2 * i + 1] = b;
    }
    const long t = n8 * 8 + (long)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) y[t] = (float)x[t];
}

// out = a (.) b with b, out bf16 and a bf16 or float32: dY (.) f'(pre-activation) where the data-gradient
// kernel upstream could not apply the multiplier itself.
// DER: b holds the activation's stored output and the factor is the derivative recovered from it (dact_from_y).
template <bool AF32, bool DER>
__global__ __launch_bounds__(kBlock) void mul_b_kernel(const void* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ out, long n8, float inv_alpha) {
    const float k1 = 1.0f + inv_alpha;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (long)gridDim.x * kBlock) {
        const u32x4 bv = ((const u32x4*)b)[i];
        float av[8];
        if (AF32) {
            const f32x4 a0 = ((const f32x4*)a)[2 * i], a1 = ((const f32x4*)a)[2 * i + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) { av[k] = a0[k]; av[4 + k] = a1[k]; }
        } else {
            const u32x4 v = ((const u32x4*)a)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { av[2 * k] = bf_lo(v[k]); av[2 * k + 1] = bf_hi(v[k]); }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float m0 = bf_lo(bv[k]), m1 = bf_hi(bv[k]);
            if (DER) { m0 = dact_from_y(m0, inv_alpha, k1); m1 = dact_from_y(m1, inv_alpha, k1); }
            o[k] = bf_pack(av[2 * k] * m0, av[2 * k + 1] * m1);
        }
        ((u32x4*)out)[i] = o;
    }
}

// ---------------- from-audio 1x1 convolution: 2 float32 channels -> 128 bf16 channels ----------------
// 16 lanes per position, 8 output channels (16 bytes) each; the lane's 24 weights / biases stay in registers.
template <bool ACT, bool DER>
__global__ __launch_bounds__(kBlock) void pw2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                         bf16_t* __restrict__ dact, long npos, float inv_alpha) {
    const int cg = threadIdx.x & 15;
    float w0[8], w1[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        w0[k] = w[cg * 8 + k];
        w1[k] = w[128 + cg * 8 + k];
        b[k] = bias ? bias[cg * 8 + k] : 0.f;
    }
    const float k1 = 1.0f + inv_alpha;
    for (long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4); p < npos; p += (long)gridDim.x * 16) {
        const float2 xv = *(const float2*)(x + 2 * p);
        float v[8], dv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = __builtin_fmaf(xv.y, w1[k], __builtin_fmaf(xv.x, w0[k], b[k]));
            if (ACT) act_bf16(v[k], inv_alpha, k1, v[k], dv[k]);
        }
        const u32x4 o = {bf_pack(v[0], v[1]), bf_pack(v[2], v[3]), bf_pack(v[4], v[5]), bf_pack(v[6], v[7])};
        *(u32x4*)(y + p * 128 + cg * 8) = o;
        if (ACT && DER) {
            const u32x4 e = {bf_pack(dv[0], dv[1]), bf_pack(dv[2], dv[3]), bf_pack(dv[4], dv[5]), bf_pack(dv[6], dv[7])};
            *(u32x4*)(dact + p * 128 + cg * 8) = e;
        }
    }
}

// dW[c][k] += sum_p x[p][c] g[p][k], dB[k] += sum_p g[p][k]: block partial sums [3][128] -> workspace.
__global__ __launch_bounds__(kBlock) void pw2_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                           float* __restrict__ partial, long npos) {
    __shared__ float red[16][3 * 128 + 4];
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    float s0[8], s1[8], sb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; sb[k] = 0.f; }
    for (long p = (long)blockIdx.x * 16 + pl; p < npos; p += (long)gridDim.x * 16) {
        const float2 xv = *(const float2*)(x + 2 * p);
        const u32x4 gv = *(const u32x4*)(g + p * 128 + cg * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = bf_lo(gv[k]), b = bf_hi(gv[k]);
            s0[2 * k] = __builtin_fmaf(xv.x, a, s0[2 * k]);
            s0[2 * k + 1] = __builtin_fmaf(xv.x, b, s0[2 * k + 1]);
            s1[2 * k] = __builtin_fmaf(xv.y, a, s1[2 * k]);
            s1[2 * k + 1] = __builtin_fmaf(xv.y, b, s1[2 * k + 1]);
            sb[2 * k] += a;
            sb[2 * k + 1] += b;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        red[pl][cg * 8 + k] = s0[k];
        red[pl][128 + cg * 8 + k] = s1[k];
        red[pl][256 + cg * 8 + k] = sb[k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 384; e += kBlock) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][e];
        partial[(long)blockIdx.x * 384 + e] = s;
    }
}

// 1024 threads = 16 consecutive sums x 64 groups of partial blocks (group g adds blocks g, g+64, ... in order),
// then the 64 group sums in a fixed order: reproducible, and ~16 dependent loads a thread instead of ~10^3.
__global__ __launch_bounds__(1024) void pw2_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                float* __restrict__ db, int nblocks) {
    __shared__ double red[64][16];
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;                  // 0..383
    double s = 0.0;
    for (int b = g; b < nblocks; b += 64) s += (double)partial[(long)b * 384 + e];
    red[g][el] = s;
    __syncthreads();
    if (g == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 64; ++q) t += red[q][el];
        if (e < 256) dw[e] += (float)t;
        else if (db) db[e - 256] += (float)t;
    }
}

// dx[p][c] (+)= sum_k g[p][k] W[c][k]  (128 bf16 channels -> 2 float32 channels)
__global__ __launch_bounds__(kBlock) void pw_to2_kernel(const bf16_t* __restrict__ g, const float* __restrict__ w,
                                                        float* __restrict__ dx, long npos, int accumulate) {
    const int cg = threadIdx.x & 15;
    float w0[8], w1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { w0[k] = w[cg * 8 + k]; w1[k] = w[128 + cg * 8 + k]; }
    const long iters = (npos + (long)gridDim.x * 16 - 1) / ((long)gridDim.x * 16);
    for (long it = 0; it < iters; ++it) {
        const long p = (it * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 4);
        const bool ok = p < npos;
        u32x4 gv = {0u, 0u, 0u, 0u};
        if (ok) gv = *(const u32x4*)(g + p * 128 + cg * 8);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = bf_lo(gv[k]), hi = bf_hi(gv[k]);
            a0 = __builtin_fmaf(lo, w0[2 * k], a0);
            a0 = __builtin_fmaf(hi, w0[2 * k + 1], a0);
            a1 = __builtin_fmaf(lo, w1[2 * k], a1);
            a1 = __builtin_fmaf(hi, w1[2 * k + 1], a1);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            a0 += __shfl_xor(a0, off, 16);
            a1 += __shfl_xor(a1, off, 16);
        }
        if (ok && cg == 0) {
            float2 o = make_float2(a0, a1);
            if (accumulate) { const float2 old = *(const float2*)(dx + 2 * p); o.x += old.x; o.y += old.y; }
            *(float2*)(dx + 2 * p) = o;
        }
    }
}

// ---------------- WeightedSum on bf16 tensors ----------------
__global__ __launch_bounds__(kBlock) void wsum_b_fwd_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                            const float* __restrict__ alpha, bf16_t* __restrict__ y,
                                                            long n8) {
    const float al = alpha[0], om = 1.0f - al;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (long)gridDim.x * kBlock) {
        const u32x4 av = ((const u32x4*)a)[i], bv = ((const u32x4*)b)[i];
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = bf_pack(mgx_blend(om, bf_lo(av[k]), al, bf_lo(bv[k])), mgx_blend(om, bf_hi(av[k]), al, bf_hi(bv[k])));
        ((u32x4*)y)[i] = o;
    }
}

__global__ __launch_bounds__(kBlock) void wsum_b_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ a,
                                                            const bf16_t* __restrict__ b, const float* __restrict__ alpha,
                                                            bf16_t* __restrict__ da, bf16_t* __restrict__ db,
                                                            double* __restrict__ part, long n8,
                                                            const bf16_t* __restrict__ ma,
                                                            const bf16_t* __restrict__ mb, float ia_a, float ia_b) {
    // ma / mb: the saved activation derivative of the layer that produced a / b.  The gradient leaves as
    // bf16(bf16((1 - alpha) dy) * ma): rounded where a multiply launch of its own would have rounded it.
    // ia_a / ia_b (1 / alpha of the activation that produced a / b; 0 = none), used where ma / mb is null: the
    // factor is the derivative recovered from a / b themselves, which are those activations' stored outputs.
    __shared__ double red[4];
    const float al = alpha[0], om = 1.0f - al;
    const bool der_a = da && !ma && ia_a > 0.f, der_b = db && !mb && ia_b > 0.f;
    const float k1a = 1.0f + ia_a, k1b = 1.0f + ia_b;
    float s = 0.f;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (long)gridDim.x * kBlock) {
        const u32x4 g = ((const u32x4*)dy)[i];
        u32x4 av = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};
        if (part || der_a) av = ((const u32x4*)a)[i];
        if (part || der_b) bv = ((const u32x4*)b)[i];
        if (da) {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = bf_pack(om * bf_lo(g[k]), om * bf_hi(g[k]));
            if (ma) {
                const u32x4 m = ((const u32x4*)ma)[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = bf_pack(bf_lo(o[k]) * bf_lo(m[k]), bf_hi(o[k]) * bf_hi(m[k]));
            } else if (der_a) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = bf_pack(bf_lo(o[k]) * dact_from_y(bf_lo(av[k]), ia_a, k1a),
                                   bf_hi(o[k]) * dact_from_y(bf_hi(av[k]), ia_a, k1a));
            }
            ((u32x4*)da)[i] = o;
        }
        if (db) {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = bf_pack(al * bf_lo(g[k]), al * bf_hi(g[k]));
            if (mb) {
                const u32x4 m = ((const u32x4*)mb)[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = bf_pack(bf_lo(o[k]) * bf_lo(m[k]), bf_hi(o[k]) * bf_hi(m[k]));
            } else if (der_b) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = bf_pack(bf_lo(o[k]) * dact_from_y(bf_lo(bv[k]), ia_b, k1b),
                                   bf_hi(o[k]) * dact_from_y(bf_hi(bv[k]), ia_b, k1b));
            }
            ((u32x4*)db)[i] = o;
        }
        if (part) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t = __builtin_fmaf(bf_lo(g[k]), bf_lo(bv[k]) - bf_lo(av[k]), t);
                t = __builtin_fmaf(bf_hi(g[k]), bf_hi(bv[k]) - bf_hi(av[k]), t);
            }
            s += t;
        }
    }
    if (part) {
        const double t = block_sum256_d((double)s, red);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void finish_sum_b_kernel(const double* __restrict__ part, int n, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] += (float)s;
}

}  // namespace

extern "C" int mgx_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return MGX_ERR_ARG;
    const long n8 = n / 8;
    hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n8 > 0 ? n8 : 1)), dim3(kBlock), 0, (hipStream_t)stream, x,
                       (bf16_t*)y, n8, (long)n);
    return mgx_launch_status();
}

extern "C" int mgx_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return MGX_ERR_ARG;
    const long n8 = n / 8;
    hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n8 > 0 ? n8 : 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const bf16_t*)x, y, n8, (long)n);
    return mgx_launch_status();
}

extern "C" int mgx_mul_bf16(const void* a, int a_is_f32, const void* b, void* out, int64_t n, void* stream) {
    if (!a || !b || !out || n <= 0 || n % 8) return MGX_ERR_ARG;
    const long n8 = n / 8;
    if (a_is_f32)
        hipLaunchKernelGGL((mul_b_kernel<true, false>), dim3(grid_for(n8)), dim3(kBlock), 0, (hipStream_t)stream, a,
                           (const bf16_t*)b, (bf16_t*)out, n8, 0.f);
    else
        hipLaunchKernelGGL((mul_b_kernel<false, false>), dim3(grid_for(n8)), dim3(kBlock), 0, (hipStream_t)stream, a,
                           (const bf16_t*)b, (bf16_t*)out, n8, 0.f);
    return mgx_launch_status();
}

extern "C" int mgx_mul_dact_bf16(const void* a, int a_is_f32, const void* y, float act_alpha, void* out, int64_t n,
                                 void* stream) {
    if (!a || !y || !out || n <= 0 || n % 8 || !(act_alpha > 0.f)) return MGX_ERR_ARG;
    const long n8 = n / 8;
    const float ia = 1.0f / act_alpha;
    if (a_is_f32)
        hipLaunchKernelGGL((mul_b_kernel<true, true>), dim3(grid_for(n8)), dim3(kBlock), 0, (hipStream_t)stream, a,
                           (const bf16_t*)y, (bf16_t*)out, n8, ia);
    else
        hipLaunchKernelGGL((mul_b_kernel<false, true>), dim3(grid_for(n8)), dim3(kBlock), 0, (hipStream_t)stream, a,
                           (const bf16_t*)y, (bf16_t*)out, n8, ia);
    return mgx_launch_status();
}

extern "C" int mgx_pointwise2_bf16_fwd(const float* x, const float* w, const float* bias, void* y, void* dact,
                                       int64_t npos, float act_alpha, void* stream) {
    if (!x || !w || !y || npos <= 0) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)((npos + 15) / 16 < kMaxGrid ? (npos + 15) / 16 : kMaxGrid);
    hipStream_t st = (hipStream_t)stream;
    const float ia = act_alpha > 0.f ? 1.0f / act_alpha : 0.f;
    if (act_alpha > 0.f && dact)
        hipLaunchKernelGGL((pw2_fwd_kernel<true, true>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, (bf16_t*)y,
                           (bf16_t*)dact, (long)npos, ia);
    else if (act_alpha > 0.f)
        hipLaunchKernelGGL((pw2_fwd_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, (bf16_t*)y,
                           (bf16_t*)nullptr, (long)npos, ia);
    else
        hipLaunchKernelGGL((pw2_fwd_kernel<false, false>), dim3(grid), dim3(kBlock), 0, st, x, w, bias, (bf16_t*)y,
                           (bf16_t*)nullptr, (long)npos, ia);
    return mgx_launch_status();
}

namespace {
inline unsigned pw2_wgrad_blocks(int64_t npos) {
    const long b = (npos + 16 * 64 - 1) / (16 * 64);      // >= 64 positions per position lane
    return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
}  // namespace

extern "C" int64_t mgx_pointwise2_bf16_wgrad_workspace(int64_t npos) {
    return (int64_t)pw2_wgrad_blocks(npos) * 384 * 4;
}

extern "C" int mgx_pointwise2_bf16_wgrad(const float* x, const void* g, float* dw, float* db, void* workspace,
                                         int64_t workspace_bytes, int64_t npos, void* stream) {
    if (!x || !g || !dw || !workspace || npos <= 0) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_pointwise2_bf16_wgrad_workspace(npos)) return MGX_ERR_WORKSPACE;
    const unsigned blocks = pw2_wgrad_blocks(npos);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pw2_wgrad_kernel, dim3(blocks), dim3(kBlock), 0, st, x, (const bf16_t*)g, (float*)workspace,
                       (long)npos);
    hipLaunchKernelGGL(pw2_wgrad_reduce_kernel, dim3(24), dim3(1024), 0, st, (const float*)workspace, dw, db, (int)blocks);
    return mgx_launch_status();
}

extern "C" int mgx_pointwise_to2_bf16(const void* g, const float* w, float* dx, int64_t npos, int accumulate,
                                      void* stream) {
    if (!g || !w || !dx || npos <= 0) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)((npos + 15) / 16 < kMaxGrid ? (npos + 15) / 16 : kMaxGrid);
    hipLaunchKernelGGL(pw_to2_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const bf16_t*)g, w, dx,
                       (long)npos, accumulate);
    return mgx_launch_status();
}

extern "C" int mgx_weighted_sum_bf16_fwd(const void* a, const void* b, const float* alpha, void* y, int64_t n,
                                         void* stream) {
    if (!a || !b || !alpha || !y || n <= 0 || n % 8) return MGX_ERR_ARG;
    hipLaunchKernelGGL(wsum_b_fwd_kernel, dim3(grid_for(n / 8)), dim3(kBlock), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, alpha, (bf16_t*)y, (long)(n / 8));
    return mgx_launch_status();
}

extern "C" int mgx_weighted_sum_bf16_bwd(const void* dy, const void* a, const void* b, const float* alpha, void* da,
                                         void* db, float* dalpha, void* workspace, int64_t workspace_bytes, int64_t n,
                                         void* stream) {
    return mgx_weighted_sum_bf16_bwd_mul(dy, a, b, alpha, da, db, dalpha, nullptr, nullptr, workspace, workspace_bytes,
                                         n, stream);
}

namespace {
int wsum_b_bwd_launch(const void* dy, const void* a, const void* b, const float* alpha, void* da, void* db,
                      float* dalpha, const void* ma, const void* mb, float act_alpha_a, float act_alpha_b, void* workspace,
                      int64_t workspace_bytes, int64_t n, void* stream) {
    if (!dy || !alpha || n <= 0 || n % 8) return MGX_ERR_ARG;
    if ((ma && !da) || (mb && !db)) return MGX_ERR_ARG;
    if (act_alpha_a < 0.f || act_alpha_b < 0.f) return MGX_ERR_ARG;
    if ((act_alpha_a > 0.f && (!da || !a)) || (act_alpha_b > 0.f && (!db || !b))) return MGX_ERR_ARG;
    if (dalpha && (!a || !b || !workspace)) return MGX_ERR_ARG;
    if (dalpha && workspace_bytes < (int64_t)(kMaxGrid * sizeof(double))) return MGX_ERR_WORKSPACE;
    const unsigned grid = grid_for(n / 8);
    double* part = dalpha ? (double*)workspace : nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wsum_b_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, (const bf16_t*)dy, (const bf16_t*)a,
                       (const bf16_t*)b, alpha, (bf16_t*)da, (bf16_t*)db, part, (long)(n / 8), (const bf16_t*)ma,
                       (const bf16_t*)mb, act_alpha_a > 0.f ? 1.0f / act_alpha_a : 0.f,
                       act_alpha_b > 0.f ? 1.0f / act_alpha_b : 0.f);
    if (dalpha) hipLaunchKernelGGL(finish_sum_b_kernel, dim3(1), dim3(64), 0, st, part, (int)grid, dalpha);
    return mgx_launch_status();
}
}  // namespace

extern "C" int mgx_weighted_sum_bf16_bwd_mul(const void* dy, const void* a, const void* b, const float* alpha, void* da,
                                             void* db, float* dalpha, const void* ma, const void* mb, void* workspace,
                                             int64_t workspace_bytes, int64_t n, void* stream) {
    return wsum_b_bwd_launch(dy, a, b, alpha, da, db, dalpha, ma, mb, 0.f, 0.f, workspace, workspace_bytes, n, stream);
}

extern "C" int mgx_weighted_sum_bf16_bwd_act(const void* dy, const void* a, const void* b, const float* alpha, void* da,
                                             void* db, float* dalpha, float act_alpha_a, float act_alpha_b, void* workspace,
                                             int64_t workspace_bytes, int64_t n, void* stream) {
    return wsum_b_bwd_launch(dy, a, b, alpha, da, db, dalpha, nullptr, nullptr, act_alpha_a, act_alpha_b, workspace,
                             workspace_bytes, n, stream);
}
