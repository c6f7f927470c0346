// Micro-benchmark: how much vector-ALU / transcendental / LDS work of ONE wave hides behind its own
// bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8 passes = 32 cycles) on gfx950, and the same with two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_overlap.hip -o tools/micro/issue_overlap && ./issue_overlap
// -DPROBE_F32 builds the same probes behind v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles).
// Prints cycles per loop iteration (s_memtime based wall clock / iterations) for k = 0..12 extra instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: k v_fma_f32; 1: k v_exp_f32; 2: k ds_read_b128; 3: k v_pk_fma_f32; 4: k v_cvt_pk_bf16_f32
// NM: MFMAs per iteration (independent accumulators round-robin over 2)
template <int MODE, int K, int NM>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)(0.002f * i); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = 0.5f + 0.01f * lane + i;
    f32x4 l[12];
    for (int i = 0; i < 12; ++i) l[i] = f32x4{0, 0, 0, 0};
    const float c1 = 0.999f, c2 = 1e-3f;
    const float* lp = lds + (lane & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#ifdef PROBE_F32
            if ((m & 1) && MODE != 5) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c2, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c2, acc0, 0, 0, 0);
#else
            if ((m & 1) && MODE != 5) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#endif
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 0 || MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c1), "v"(c2));
                else if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                else if (MODE == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[k]) : "v"((unsigned)(size_t)lp), "n"(0));
                else if (MODE == 3) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 t = {v[k], v[(k + 6) % 12]};
                    f2 cc1 = {c1, c1}, cc2 = {c2, c2};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(cc1), "v"(cc2));
                    v[k] = t[0];
                } else if (MODE == 4) {
                    unsigned r;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[k]), "v"(c1));
                    v[k] = __builtin_bit_cast(float, r);
                }
            }
        }
        if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 12; ++i) s += v[i] + l[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int K, int NM>
double run(int threads, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, K, NM>), dim3(256), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K, NM>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)iters * NM);   // cycles per MFMA slot at 2.4 GHz
}

template <int MODE>
void sweep(const char* name, float* out) {
    const int iters = 20000;
    for (int threads : {256, 512}) {
        printf("%-18s waves/SIMD=%d  cycles per (MFMA + k ops):", name, threads / 256);
        printf(" k0=%.1f", run<MODE, 0, 4>(threads, out, iters));
        printf(" k1=%.1f", run<MODE, 1, 4>(threads, out, iters));
        printf(" k2=%.1f", run<MODE, 2, 4>(threads, out, iters));
        printf(" k4=%.1f", run<MODE, 4, 4>(threads, out, iters));
        printf(" k6=%.1f", run<MODE, 6, 4>(threads, out, iters));
        printf(" k8=%.1f", run<MODE, 8, 4>(threads, out, iters));
        printf(" k12=%.1f\n", run<MODE, 12, 4>(threads, out, iters));
        fflush(stdout);
    }
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    sweep<5>("dependent chain+fma", out);
    sweep<0>("v_fma_f32", out);
    sweep<1>("v_exp_f32", out);
    sweep<3>("v_pk_fma_f32", out);
    sweep<4>("v_cvt_pk_bf16_f32", out);
    sweep<2>("ds_read_b128", out);
    hipFree(out);
    return 0;
}
