// Micro-benchmark: cycles per MFMA when every MFMA continues the SAME accumulator against NACC accumulators in
// rotation, for v_mfma_f32_32x32x2_f32 (16 passes) and v_mfma_f32_32x32x16_bf16 (8 passes), one and two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool F32, int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * threadIdx.x); b[i] = (__bf16)(0.002f * i); }
    const float fa = 0.001f * threadIdx.x, fb = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (F32) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[m % NACC], 0, 0, 0);
            else acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool F32, int NACC>
double run(int threads, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<F32, NACC>), dim3(256), dim3(threads), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<F32, NACC>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)iters * 8) / (threads / 256);   // SIMD cycles per MFMA at 2.4 GHz
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    for (int threads : {256, 512}) {
        printf("waves/SIMD=%d  f32 32x32x2:  1 acc %.1f  2 acc %.1f  4 acc %.1f   bf16 32x32x16: 1 acc %.1f  2 acc %.1f  4 acc %.1f  (SIMD cycles per MFMA)\n",
               threads / 256, run<true, 1>(threads, out), run<true, 2>(threads, out), run<true, 4>(threads, out),
               run<false, 1>(threads, out), run<false, 2>(threads, out), run<false, 4>(threads, out));
        fflush(stdout);
    }
    (void)hipFree(out);
    return 0;
}
