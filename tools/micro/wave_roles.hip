// Micro-benchmark: do the MFMAs of one wave overlap with the vector-ALU work of ANOTHER wave on the same SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wave_roles.hip -o tools/micro/wave_roles && ./wave_roles
// A workgroup of 8 waves puts two waves on every SIMD of a CU.  Each wave reads the SIMD it sits on from HW_ID; the
// first wave on a SIMD plays role A, the second role B; a role that is switched off returns at once.  Every wave clocks its own loop with
// s_memtime and the host reports the slowest wave of each role, in SIMD cycles per iteration:
//   A alone, B alone, A and B together.   Together = max(A, B): the roles overlap completely;
//                                          together = A + B:    they take turns.
// Roles: M = `NM` v_mfma_f32_32x32x16_bf16 per iteration (accumulators in rotation);
//        T = the activation tanh(x) + x/alpha of 16 values a lane as the convolution kernels issue it
//            (v_mul, v_exp, v_add, v_rcp, v_fma, v_fma per value = 96 instructions);
//        P = the same with the four non-transcendental instructions as packed pairs held in aligned register
//            pairs for the whole loop (v_pk_mul, v_pk_add, 2 x v_pk_fma per two values);
//        L = 16 ds_read_b128 per iteration (the B-operand reads of a tile row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { ROLE_NONE = 0, ROLE_M = 1, ROLE_T = 2, ROLE_P = 3, ROLE_L = 4 };

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// A role runs `iters` iterations, or, with iters < 0, until the LDS word *stop becomes non-zero (polled once per
// iteration); `done` receives the number of iterations it ran.
// (two plain loop forms chosen at compile time: a condition that mixes both makes the compiler copy the accumulators
// around in every iteration)
#define LOOP_HEAD for (int it = 0; OPEN ? ((it & 7) != 0 || *stop == 0) : (it < iters); ++it, ++done)

template <int NM, bool OPEN>
__device__ float role_mfma(int iters, int lane, volatile int* stop, int& done) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * lane); b[i] = (__bf16)(0.002f * i); }
    LOOP_HEAD {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    return s;
}

template <bool OPEN>
__device__ float role_tanh(int iters, int lane, volatile int* stop, int& done) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.01f * lane + 0.1f * i;
    const float k2 = 2.885390f, one = 1.0f, m2 = -2.0f, ia = 1e-4f;
    LOOP_HEAD {
        // four values at a time, instruction by instruction: no instruction waits for the one in front of it
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            float e[4], r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e[j]) : "v"(v[i + j]), "v"(k2));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(e[j]) : "v"(one));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[j]) : "v"(e[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(m2), "v"(one));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i + j]) : "v"(v[i + j]), "v"(ia), "v"(r[j]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    return s;
}

template <bool OPEN>
__device__ float role_tanh_packed(int iters, int lane, volatile int* stop, int& done) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{0.01f * lane + 0.1f * i, 0.02f * lane + 0.05f * i};
    const f32x2 k2 = {2.885390f, 2.885390f}, one = {1.0f, 1.0f}, m2 = {-2.0f, -2.0f}, ia = {1e-4f, 1e-4f};
    LOOP_HEAD {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 e[2], r[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(e[j]) : "v"(v[i + j]), "v"(k2));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(e[j][0]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(e[j][1]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e[j]) : "v"(one));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                asm volatile("v_rcp_f32 %0, %1" : "=v"(r[j][0]) : "v"(e[j][0]));
                asm volatile("v_rcp_f32 %0, %1" : "=v"(r[j][1]) : "v"(e[j][1]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(m2), "v"(one));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v[i + j]) : "v"(v[i + j]), "v"(ia), "v"(r[j]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    return s;
}

template <bool OPEN>
__device__ float role_lds(int iters, int lane, const float* lds, volatile int* stop, int& done) {
    f32x4 l[4];
    for (int i = 0; i < 4; ++i) l[i] = f32x4{0, 0, 0, 0};
    const unsigned base = (unsigned)(size_t)(lds + (lane & 63) * 4);
    LOOP_HEAD {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[i & 3]) : "v"(base), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return l[0][0] + l[1][1] + l[2][2] + l[3][3];
}

template <int NM>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* clocks, int iters_a, int iters_b, int role_a,
                                             int role_b, int prio_a, int prio_b) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    __shared__ int on_simd[4];
    __shared__ int stop_simd[4];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-6f;
    if (threadIdx.x < 4) { on_simd[threadIdx.x] = 0; stop_simd[threadIdx.x] = 0; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // which SIMD this wave sits on (HW_ID bits 5:4), and whether it is the first or the second wave there
    const unsigned hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    const int simd = (hw_id >> 4) & 3;
    int rank = 0;
    if (lane == 0) rank = atomicAdd(&on_simd[simd], 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
    const int role = rank == 0 ? role_a : role_b;
    const int iters = rank == 0 ? iters_a : iters_b;
    float s = 0.f;
    int done = 0;
    volatile int* stop = &stop_simd[simd];
    if ((rank == 0 ? prio_a : prio_b) == 3) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = now();
    if (iters >= 0) {
        if (role == ROLE_M) s = role_mfma<NM, false>(iters, lane, stop, done);
        else if (role == ROLE_T) s = role_tanh<false>(iters, lane, stop, done);
        else if (role == ROLE_P) s = role_tanh_packed<false>(iters, lane, stop, done);
        else if (role == ROLE_L) s = role_lds<false>(iters, lane, lds, stop, done);
    } else {
        if (role == ROLE_M) s = role_mfma<NM, true>(iters, lane, stop, done);
        else if (role == ROLE_T) s = role_tanh<true>(iters, lane, stop, done);
        else if (role == ROLE_P) s = role_tanh_packed<true>(iters, lane, stop, done);
        else if (role == ROLE_L) s = role_lds<true>(iters, lane, lds, stop, done);
    }
    const unsigned long long t1 = now();
    if (iters >= 0 && lane == 0) *stop = 1;             // the counted role releases the open-ended one on its SIMD
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + done;
    if (lane == 0) clocks[2048 + blockIdx.x * 8 + wave] = (unsigned long long)done;
    // bit 63 marks a role-B wave; bits 60-61 the SIMD
    if (lane == 0) clocks[blockIdx.x * 8 + wave] = (role == ROLE_NONE ? 0ull : t1 - t0) | ((unsigned long long)(rank != 0) << 63)
                                                   | ((unsigned long long)simd << 60);
}

// s_memtime counts at 100 MHz on this part; the shader clock is taken from the all-MFMA run (8 passes = 32 cycles)
struct Result { double a, b; };

// Role A runs `iters` iterations; role B runs iters_b of its own (alone) or, with iters_b < 0, for as long as the role-A
// wave of its SIMD is busy.  Returns ticks per iteration of the slowest wave of each role.
template <int NM>
Result run(float* out, unsigned long long* clocks, int iters, int role_a, int role_b, int iters_b = 0, int prio_a = 0,
           int prio_b = 0) {
    const int blocks = 8;          // one workgroup per XCD: a chip full of MFMA waves lowers the shader clock under the power limit
    if (!iters_b) iters_b = iters;
    std::vector<unsigned long long> h(4096);
    hipLaunchKernelGGL((probe<NM>), dim3(blocks), dim3(512), 0, 0, out, clocks, 16, 16, role_a, role_b == ROLE_NONE ? role_b : role_b, 0, 0);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<NM>), dim3(blocks), dim3(512), 0, 0, out, clocks, iters, iters_b, role_a, role_b, prio_a, prio_b);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int blk = 0; blk < blocks; ++blk)
        for (int w = 0; w < 8; ++w) {
            const unsigned long long t = h[blk * 8 + w] & ((1ull << 60) - 1);
            const double n = (double)h[2048 + blk * 8 + w];
            if (n == 0) continue;
            if (!(h[blk * 8 + w] >> 63)) a = std::max(a, (double)t / n);
            else b = std::max(b, (double)t / n);
        }
    return {a, b};
}

int main() {
    float* out;
    unsigned long long* clocks;
    if (hipMalloc(&out, 256 * 512 * sizeof(float)) != hipSuccess) { printf("no device\n"); return 1; }
    hipMalloc(&clocks, 4096 * sizeof(unsigned long long));
    const int iters = 4000;
    constexpr int NM = 18;                       // the MFMAs of one 3x3 tile row, 32 -> 32 channels
    // clock ratio: NM MFMAs alone take NM x 32 shader cycles (one wave per SIMD, accumulators in rotation)
    for (int warm = 0; warm < 3; ++warm) run<NM>(out, clocks, iters, ROLE_M, ROLE_M);      // let the shader clock settle
    Result cal = run<NM>(out, clocks, iters, ROLE_M, ROLE_NONE);
    const double ticks_per_cycle = cal.a / (NM * 32.9);
    printf("calibration: %d MFMAs alone = %.2f s_memtime ticks (taken as %d x 32.9 shader cycles)\n", NM, cal.a, NM);
    auto cyc = [&](double t) { return t / ticks_per_cycle; };
    struct { const char* name; int role; } bs[] = {{"T (tanh, 96 plain VALU)", ROLE_T}, {"P (tanh, packed pairs)", ROLE_P},
                                                   {"L (16 ds_read_b128)", ROLE_L}, {"M (a second MFMA wave)", ROLE_M}};
    for (auto& b : bs) {
        Result alone = run<NM>(out, clocks, iters, ROLE_NONE, b.role);
        printf("%-26s alone %7.1f cycles/iteration\n", b.name, cyc(alone.b));
        for (int pa : {0, 3})
            for (int pb : {0, 3}) {
                if (pa == 3 && pb == 3) continue;
                // the MFMA wave runs a fixed count, this wave for as long as the MFMA wave of its SIMD is busy ...
                Result x = run<NM>(out, clocks, iters, ROLE_M, b.role, -1, pa, pb);
                // ... and the other way round (roles swapped: this wave counted, the MFMA wave open-ended)
                Result y = run<NM>(out, clocks, iters, b.role, ROLE_M, -1, pb, pa);
                printf("    priorities (MFMA %d, this %d): counted MFMA wave %7.1f beside this wave at %7.1f | counted this wave %7.1f "
                       "beside an MFMA wave at %7.1f\n", pa, pb, cyc(x.a), cyc(x.b), cyc(y.a), cyc(y.b));
            }
        fflush(stdout);
    }
    Result cal2 = run<NM>(out, clocks, iters, ROLE_M, ROLE_NONE);
    printf("calibration repeated at the end: %.2f ticks (%.3f of the first)\n", cal2.a, cal2.a / cal.a);
    Result tt = run<NM>(out, clocks, iters, ROLE_T, ROLE_T);
    printf("two tanh waves on a SIMD: %.1f / %.1f cycles per iteration each\n", cyc(tt.a), cyc(tt.b));
    hipFree(out);
    hipFree(clocks);
    return 0;
}
