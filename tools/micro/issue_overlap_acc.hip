// Micro-benchmark: does it matter for the overlap of a wave's own vector-ALU work with its bf16 MFMAs
// (v_mfma_f32_32x32x16_bf16) WHERE the MFMA's operands live?  The vector ALU and the matrix unit read the same
// architectural register file; accumulation registers have a file of their own.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/issue_overlap_acc.hip -o tools/micro/issue_overlap_acc
// Variants: accumulator C/D in vector (v) or accumulation (a) registers, A operand in v or a, B always in v.
// One wave per SIMD, 8 workgroups (the shader clock stays at its maximum), s_memtime ticks converted with the
// all-MFMA run taken as 32 cycles per MFMA.  Prints cycles per (MFMA + k instructions).
// Extra instruction kinds: F = v_fma_f32; T = the activation tanh(x) + x/alpha as the convolution kernels issue
// it (v_mul, v_exp, v_add, v_rcp, v_fma, v_fma per value; k counts VALUES, 6 instructions each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int ACC_A, int OPA_A>
__device__ __forceinline__ void mfma(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (ACC_A && OPA_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
    else if constexpr (ACC_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else if constexpr (OPA_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// KIND 0: k v_fma_f32 per MFMA; KIND 1: k activation values (6 instructions each) per MFMA
template <int ACC_A, int OPA_A, int KIND, int K, int NM>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* clocks, int iters) {
    const int lane = threadIdx.x;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x3c003c00u + lane; b[i] = 0x3a003a00u + i; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.01f * lane + 0.1f * i;
    const float c1 = 0.999f, c2 = 1e-3f, k2 = 2.885390f, one = 1.0f, m2 = -2.0f, ia = 1e-4f;
    float pv[32];
    for (int i = 0; i < 32; ++i) pv[i] = 0.3f + 0.01f * i;
    float ee[16], rr[16];
    for (int i = 0; i < 16; ++i) { ee[i] = 1.f; rr[i] = 1.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            mfma<ACC_A, OPA_A>(acc[m & 3], a, b);
            if constexpr (KIND == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(c1), "v"(c2));
            } else if constexpr (KIND == 2) {
                // the dependency structure of T with every instruction a plain v_fma_f32
                float e[K > 0 ? K : 1], r[K > 0 ? K : 1];
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(e[k]) : "v"(v[k]), "v"(k2));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(e[k]) : "v"(c1));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(e[k]) : "v"(one));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(r[k]) : "v"(e[k]), "v"(c1));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(m2), "v"(one));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(v[k]), "v"(ia), "v"(r[k]));
            } else if constexpr (KIND == 3) {
                // T on 2 K values spread over two MFMA slots: the MFMA sits in the middle of the instruction rows,
                // every instruction 2 K instructions behind the one it waits for
                if ((m & 1) == 0) {
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ee[k]) : "v"(v[k]), "v"(k2));
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(ee[k]));
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ee[k]) : "v"(one));
                } else {
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_rcp_f32 %0, %1" : "=v"(rr[k]) : "v"(ee[k]));
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(rr[k]) : "v"(m2), "v"(one));
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(v[k]), "v"(ia), "v"(rr[k]));
                }
            } else if constexpr (KIND == 6) {
                // T as a rotating pipeline: every MFMA slot issues K groups of the six instructions, each instruction
                // of a group on a DIFFERENT value (stage i of value (slot - i)): nothing in a slot waits for anything
                // issued in the same slot
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float* q = pv + 8 * k;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q[5]) : "v"(q[5]), "v"(ia), "v"(q[4]));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(q[4]) : "v"(q[3]), "v"(m2), "v"(one));
                    asm volatile("v_rcp_f32 %0, %1" : "=v"(q[3]) : "v"(q[2]));
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[2]) : "v"(q[1]), "v"(one));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(q[1]) : "v"(q[0]));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q[0]) : "v"(q[5]), "v"(k2));
                }
            } else if constexpr (KIND == 4) {
                // k v_exp_f32 only
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
            } else if constexpr (KIND == 5) {
                // k v_rcp_f32 only
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
            } else {
                // K values, instruction by instruction (no instruction waits for the one in front of it when K >= 2)
                float e[K > 0 ? K : 1], r[K > 0 ? K : 1];
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e[k]) : "v"(v[k]), "v"(k2));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(e[k]));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(e[k]) : "v"(one));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[k]) : "v"(e[k]));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(m2), "v"(one));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(v[k]), "v"(ia), "v"(r[k]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) {
        f32x16 t = acc[j];
        if constexpr (ACC_A) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(t));
        for (int i = 0; i < 16; ++i) s += t[i];
    }
    for (int i = 0; i < 16; ++i) s += v[i] + ee[i] + rr[i] + pv[i] + pv[16 + i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((lane & 63) == 0) clocks[blockIdx.x * 4 + (lane >> 6)] = t1 - t0;
}

template <int ACC_A, int OPA_A, int KIND, int K>
double run(float* out, unsigned long long* clocks) {
    constexpr int NM = 8;
    const int iters = 4000, blocks = 8;
    std::vector<unsigned long long> h(blocks * 4);
    hipLaunchKernelGGL((probe<ACC_A, OPA_A, KIND, K, NM>), dim3(blocks), dim3(256), 0, 0, out, clocks, 16);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<ACC_A, OPA_A, KIND, K, NM>), dim3(blocks), dim3(256), 0, 0, out, clocks, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long t = 0;
    for (auto x : h) t = std::max(t, x);
    return (double)t / ((double)iters * NM);          // ticks per MFMA slot
}

template <int ACC_A, int OPA_A>
void sweep(const char* name, float* out, unsigned long long* clocks, double tick) {
    printf("%-34s F (v_fma):", name);
    printf(" k0=%.1f", run<ACC_A, OPA_A, 0, 0>(out, clocks) / tick);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 0, 2>(out, clocks) / tick);
    printf(" k4=%.1f", run<ACC_A, OPA_A, 0, 4>(out, clocks) / tick);
    printf(" k6=%.1f", run<ACC_A, OPA_A, 0, 6>(out, clocks) / tick);
    printf(" k8=%.1f", run<ACC_A, OPA_A, 0, 8>(out, clocks) / tick);
    printf(" k12=%.1f", run<ACC_A, OPA_A, 0, 12>(out, clocks) / tick);
    printf(" k16=%.1f\n", run<ACC_A, OPA_A, 0, 16>(out, clocks) / tick);
    printf("%-34s T (tanh values):", name);
    printf(" k1=%.1f", run<ACC_A, OPA_A, 1, 1>(out, clocks) / tick);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 1, 2>(out, clocks) / tick);
    printf(" k3=%.1f", run<ACC_A, OPA_A, 1, 3>(out, clocks) / tick);
    printf(" k4=%.1f\n", run<ACC_A, OPA_A, 1, 4>(out, clocks) / tick);
    printf("%-34s T's dependencies, all v_fma:", name);
    printf(" k1=%.1f", run<ACC_A, OPA_A, 2, 1>(out, clocks) / tick);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 2, 2>(out, clocks) / tick);
    printf(" k4=%.1f\n", run<ACC_A, OPA_A, 2, 4>(out, clocks) / tick);
    printf("%-34s T spread over two MFMA slots:", name);
    printf(" k1=%.1f", run<ACC_A, OPA_A, 3, 1>(out, clocks) / tick);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 3, 2>(out, clocks) / tick);
    printf(" k4=%.1f\n", run<ACC_A, OPA_A, 3, 4>(out, clocks) / tick);
    printf("%-34s T as a rotating pipeline (k groups of 6 independent instructions):", name);
    printf(" k1=%.1f", run<ACC_A, OPA_A, 6, 1>(out, clocks) / tick);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 6, 2>(out, clocks) / tick);
    printf(" k3=%.1f", run<ACC_A, OPA_A, 6, 3>(out, clocks) / tick);
    printf(" k4=%.1f\n", run<ACC_A, OPA_A, 6, 4>(out, clocks) / tick);
    printf("%-34s v_exp only:", name);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 4, 2>(out, clocks) / tick);
    printf(" k4=%.1f", run<ACC_A, OPA_A, 4, 4>(out, clocks) / tick);
    printf(" k8=%.1f", run<ACC_A, OPA_A, 4, 8>(out, clocks) / tick);
    printf(" k16=%.1f\n", run<ACC_A, OPA_A, 4, 16>(out, clocks) / tick);
    printf("%-34s v_rcp only:", name);
    printf(" k2=%.1f", run<ACC_A, OPA_A, 5, 2>(out, clocks) / tick);
    printf(" k4=%.1f", run<ACC_A, OPA_A, 5, 4>(out, clocks) / tick);
    printf(" k8=%.1f", run<ACC_A, OPA_A, 5, 8>(out, clocks) / tick);
    printf(" k16=%.1f\n", run<ACC_A, OPA_A, 5, 16>(out, clocks) / tick);
    fflush(stdout);
}

int main() {
    float* out;
    unsigned long long* clocks;
    if (hipMalloc(&out, 8 * 256 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    hipMalloc(&clocks, 64 * sizeof(unsigned long long));
    for (int w = 0; w < 3; ++w) run<0, 0, 0, 0>(out, clocks);
    const double tick = run<0, 0, 0, 0>(out, clocks) / 32.0;        // s_memtime ticks per shader cycle
    printf("calibration: all-MFMA loop = %.3f s_memtime ticks per MFMA, taken as 32 cycles\n", tick * 32.0);
    sweep<0, 0>("acc v, A v", out, clocks, tick);
    sweep<1, 1>("acc a, A a", out, clocks, tick);
    hipFree(out);
    hipFree(clocks);
    return 0;
}
