// Shared device helpers for the gfx950 (MI355X) kernels of libmgx.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define MGX_OK 0
#define MGX_ERR_ARG 1
#define MGX_ERR_LAUNCH 2
#define MGX_ERR_WORKSPACE 3

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact f32 fma chain.
// Lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; element `reg` of the
// accumulator in lane l is D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `red` holds >= 4 floats.
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum256_d(double v, double* red) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}

// tanh(x) and its derivative sech^2(x) = 1 - tanh^2 without the cancellation of 1 - t*t in the
// saturated range: with e = exp(-2|x|), tanh = sign(x) (1-e)/(1+e) and sech^2 = 4e/(1+e)^2.
// exp2 and the reciprocal are the hardware approximations (v_exp_f32, v_rcp_f32: ~1 ulp), which
// keeps the result within ~3e-7 of the exact value: far inside the float32 parity tolerance, and a
// fraction of the instructions of the IEEE-exact library calls in the convolution epilogues.
__device__ __forceinline__ void tanh_sech2(float x, float& t, float& s) {
    const float e = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x));   // exp(-2|x|)
    const float inv = __builtin_amdgcn_rcpf(1.0f + e);
    const float m = (1.0f - e) * inv;
    t = copysignf(m, x);
    s = 4.0f * e * inv * inv;
}

// WeightedSum of trainer/models.py:30-35 for one element, (1 - alpha) a + alpha b with om = 1 - alpha: one
// rounding order for the blend kernels and for the convolution epilogues that blend in place of a launch.
__device__ __forceinline__ float mgx_blend(float om, float a, float al, float b) {
    return __builtin_fmaf(om, a, al * b);
}

static inline int mgx_launch_status() {
    return hipGetLastError() == hipSuccess ? MGX_OK : MGX_ERR_LAUNCH;
}

// More than 64 KiB of dynamic LDS must be allowed per kernel function AND per device.  `done` (one
// static per call site) holds one bit per device already served; two threads racing on a first call
// both set the attribute, which is harmless.  A failure is left pending in the runtime's last-error
// slot (and the launch that follows fails as well), so mgx_launch_status() reports it.
template <typename K>
static inline void mgx_allow_dynamic_lds(K kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
        done.fetch_or(bit, std::memory_order_release);
}
