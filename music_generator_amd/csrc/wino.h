// Helpers shared by the Winograd kernels (wconv.hip forward / data-gradient, wwgrad.hip weight
// gradient): MFMAs and accumulator copies as asm statements that keep a 256-register accumulator
// set in the accumulation registers, packed fp32 arithmetic.
#pragma once
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_acc(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The result of a transcendental instruction (v_exp_f32, v_rcp_f32) needs a wait state before an
// ordinary vector instruction may read it.  The compiler inserts it for the instructions it
// selects itself, not for the operands of an asm statement: values that go from a transcendental
// into one of the packed helpers above pass through this first.
__device__ __forceinline__ void trans_settle(f32x2& v) { asm volatile("s_nop 1" : "+v"(v)); }
// Copy of an accumulator tile out of the accumulation registers AT THIS POINT of the program: left
// to the compiler, the copies of all 16 tiles are placed right behind the last MFMA and occupy
// every vector register before the first of them is used.
__device__ __forceinline__ f32x16 acc_read16(const f32x16& a) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a[i]));
        r[i] = v;
    }
    return r;
}
__device__ __forceinline__ f32x2 pair16(f32x16 v, int k) { return f32x2{v[2 * k], v[2 * k + 1]}; }
__device__ __forceinline__ f32x16 add16(f32x16 a, f32x16 b) {
    f32x16 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 t = pk_add(pair16(a, k), pair16(b, k));
        r[2 * k] = t[0], r[2 * k + 1] = t[1];
    }
    return r;
}
__device__ __forceinline__ f32x16 sub16(f32x16 a, f32x16 b) {
    f32x16 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 t = pk_sub(pair16(a, k), pair16(b, k));
        r[2 * k] = t[0], r[2 * k + 1] = t[1];
    }
    return r;
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

