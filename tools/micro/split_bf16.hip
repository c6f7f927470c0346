// Micro-benchmark: a float32 GEMM tile on the bf16 matrix instruction by splitting every operand into three bf16
// parts (x = hi + mid + lo, 8 + 8 + 8 mantissa bits) and keeping the six largest of the nine partial products,
// against the float32 matrix instruction - speed (cycles per K = 16 slab of a 32 x 32 tile, the split of the B
// operand done in the loop as a convolution kernel would do it on staged activations, the A operand split once
// like weights) and accuracy (one 32 x 32 x 64 product against float64 on the host).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/split_bf16.hip -o tools/micro/split_bf16 && tools/micro/split_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Split {
    bf16x8 hi, mid, lo;
};

__device__ __forceinline__ Split split8(const float (&x)[8]) {
    Split s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r1 = x[i] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        s.hi[i] = h;
        s.mid[i] = m;
        s.lo[i] = (__bf16)r2;
    }
    return s;
}

__device__ __forceinline__ f32x16 mma6(const Split& a, const Split& b, f32x16 acc) {
    // smallest terms first, so that they are not lost against the large partial sum
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

// ---- speed: MODE 0 float32 MFMAs (8 per slab), 1 split + 6 bf16 MFMAs, 2 the 6 bf16 MFMAs without the split ----
template <int MODE>
__global__ __launch_bounds__(512) void speed(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float xa[8], xb[8];
    for (int i = 0; i < 8; ++i) { xa[i] = 0.37f + 0.001f * lane + 0.01f * i; xb[i] = 0.11f + 0.002f * lane - 0.01f * i; }
    const Split a = split8(xa);
    Split b = split8(xb);
    for (int it = 0; it < iters; ++it) {
        // two slabs per iteration on two accumulators; the B operand changes every slab (new activations)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[i] += 1e-4f;
            f32x16& acc = s ? acc1 : acc0;
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[k], xb[k], acc, 0, 0, 0);
            } else {
                if (MODE == 1) b = split8(xb);
                acc = mma6(a, b, acc);
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) r += xb[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run_speed(int threads, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(speed<MODE>, dim3(256), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(speed<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 2);        // cycles per slab and wave slot at 2.4 GHz
}

// ---- accuracy: C = A (32 x 64) * B (64 x 32), one wave ----
__global__ __launch_bounds__(64) void accuracy(const float* A, const float* B, float* c32, float* c6, float* c1) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 s32, s6, s1;
    for (int i = 0; i < 16; ++i) { s32[i] = 0.f; s6[i] = 0.f; s1[i] = 0.f; }
    for (int k0 = 0; k0 < 64; k0 += 2) s32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 64 + k0 + h], B[(k0 + h) * 32 + r], s32, 0, 0, 0);
    for (int k0 = 0; k0 < 64; k0 += 16) {
        float xa[8], xb[8];
        for (int j = 0; j < 8; ++j) { xa[j] = A[r * 64 + k0 + 8 * h + j]; xb[j] = B[(k0 + 8 * h + j) * 32 + r]; }
        const Split a = split8(xa), b = split8(xb);
        s6 = mma6(a, b, s6);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, s1, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i / 4) * 8 + h * 4 + (i & 3);
        c32[row * 32 + r] = s32[i];
        c6[row * 32 + r] = s6[i];
        c1[row * 32 + r] = s1[i];
    }
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    for (int threads : {256, 512}) {
        const double f = run_speed<0>(threads, out), s = run_speed<1>(threads, out), p = run_speed<2>(threads, out);
        printf("waves/SIMD=%d  cycles per K=16 slab: 8 x f32 MFMA %.1f | split + 6 x bf16 MFMA %.1f (x%.2f) | 6 x bf16 MFMA "
               "alone %.1f (x%.2f)\n", threads / 256, f, s, f / s, p, f / p);
    }
    std::vector<float> A(32 * 64), B(64 * 32);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dA, *dB, *d32, *d6, *d1;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
    hipMalloc(&d32, 4096); hipMalloc(&d6, 4096); hipMalloc(&d1, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, d32, d6, d1);
    std::vector<float> c32(1024), c6(1024), c1(1024);
    hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(c6.data(), d6, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), d1, 4096, hipMemcpyDeviceToHost);
    double e32 = 0, e6 = 0, e1 = 0, scale = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < 64; ++k) { ref += (double)A[i * 64 + k] * B[k * 32 + j]; mag += fabs((double)A[i * 64 + k] * B[k * 32 + j]); }
            e32 = fmax(e32, fabs(c32[i * 32 + j] - ref) / mag);
            e6 = fmax(e6, fabs(c6[i * 32 + j] - ref) / mag);
            e1 = fmax(e1, fabs(c1[i * 32 + j] - ref) / mag);
            scale = fmax(scale, mag);
        }
    printf("max |error| / sum |a b| over a 32 x 32 x 64 product: f32 MFMA %.2e | six split products %.2e | plain bf16 %.2e\n",
           e32, e6, e1);
    return 0;
}
