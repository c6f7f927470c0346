// bf16 storage helpers for the gfx950 kernels that keep activations in bfloat16 and accumulate in
// float32 (v_mfma_f32_32x32x16_bf16).
#pragma once
#include "common.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The two bf16 halves of a dword as float32 (exact).
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Two float32 -> one dword of two bf16, round to nearest even (v_cvt_pk_bf16_f32; NaN stays NaN).
__device__ __forceinline__ unsigned bf_pack(float lo, float hi) {
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}

__device__ __forceinline__ float bf_to_f(bf16_t v) { return (float)v; }

// One v_mfma_f32_32x32x16_bf16: D(32x32) = A(32x16) * B(16x32) + C.  Lane l (r = l&31, h = l>>5)
// supplies A[row r][k = 8h + j] and B[k = 8h + j][col r] in element j of its fragment; element
// `reg` of the accumulator in lane l is D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r].
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
}

// Wait states between the last MFMA that writes an accumulator and an inline-asm statement that reads it
// (the compiler pads vector-ALU readers it schedules itself, but nothing inside or in front of asm).
__device__ __forceinline__ void mfma_result_wait() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"); }
// The same, tied to the accumulator it protects: without a data dependency the compiler is free to schedule
// the wait in front of the MFMAs it is meant to follow.
__device__ __forceinline__ void mfma_result_wait(f32x16& acc) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));
}

// After this exchange between lanes l and l^32, lane (r, h) holds rows 8h .. 8h+7 of column r of a 32x32
// accumulator tile in a[0..7] and rows 16+8h .. 16+8h+7 in a[8..15]: as bf16 the two lanes of a column hold
// bytes [16h, 16h+16) and [32+16h, 32+16h+16) of its 64, so one 16-byte store instruction of the wave
// writes 32 consecutive bytes of every column (whole 32-byte sectors).
__device__ __forceinline__ void acc_rows_interleaved(f32x16& a) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // first operand: registers 8g+e (lanes >= 32: rows 16g+4+e), second: registers 8g+4+e (lanes < 32:
            // rows 16g+8+e); the swap exchanges the first's upper lanes with the second's lower lanes
            float lo = a[8 * g + e], hi = a[8 * g + 4 + e];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
            a[8 * g + e] = lo;
            a[8 * g + 4 + e] = hi;
        }
    }
}

// After this exchange between lanes l and l^32, lane (r, h) holds rows 16h .. 16h+15 of column r
// of a 32x32 accumulator tile in a[0..15] (in order).
__device__ __forceinline__ void acc_rows_contiguous(f32x16& a) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // registers 4g+e (rows 8g+4h+e) and 4(g+2)+e (rows 8(g+2)+4h+e): v_permlane32_swap exchanges the
            // first operand's lanes 32-63 with the second operand's lanes 0-31.  Inline asm: the builtin of
            // this toolchain returns the first result twice.  The s_nop covers the wait states between a
            // vector-ALU write of an operand and the swap (nothing is padded inside an asm statement).
            float lo = a[4 * g + e], hi = a[4 * (g + 2) + e];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
            a[4 * g + e] = lo;
            a[4 * (g + 2) + e] = hi;
        }
    }
    // lanes < 32 now hold: a[0..3] rows 0-3, a[8..11] rows 4-7, a[4..7] rows 8-11, a[12..15] rows 12-15
    // lanes >= 32:         a[0..3] rows 16-19, a[8..11] rows 20-23, a[4..7] rows 24-27, a[12..15] rows 28-31
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = a[4 + e];
        a[4 + e] = a[8 + e];
        a[8 + e] = t;
    }
}

// tanh(x) + x/alpha and its derivative 1 - tanh(x)^2 + 1/alpha for bf16 outputs: exp2 and rcp are
// the hardware approximations, 1 - t*t is taken directly (its cancellation error ~1e-7 is far
// below the 2^-9 rounding of the stored value).  inv_alpha = 1/alpha, k1 = 1 + 1/alpha.
__device__ __forceinline__ void act_bf16(float x, float inv_alpha, float k1, float& y, float& dv) {
    const float e = __builtin_amdgcn_exp2f(2.885390081777927f * x);     // exp(2x); inf / 0 at the ends
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const float t = __builtin_fmaf(-2.0f, r, 1.0f);
    y = __builtin_fmaf(x, inv_alpha, t);
    dv = __builtin_fmaf(-t, t, k1);
}

// The derivative 1 - tanh(v)^2 + 1/alpha recovered from the STORED activation output y = bf16(tanh(v) + v/alpha), for
// backward passes that keep no derivative tensor: 1 + 1/alpha - y^2.  The v/alpha term of y shifts it by 2 tanh(v) v/alpha
// (below the 2 |y| 2^-9 that the rounding of y itself contributes for every alpha the reference uses, 11 000 and up);
// the true derivative is never below 1/alpha, which the maximum restores where a y rounded up past 1 would turn
// the difference negative.  inv_alpha = 1/alpha, k1 = 1 + 1/alpha.
__device__ __forceinline__ float dact_from_y(float y, float inv_alpha, float k1) {
    return __builtin_fmaxf(__builtin_fmaf(-y, y, k1), inv_alpha);
}

// ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements; lane 4q+p
// supplies the address of row q, columns 4p..4p+3 (8-byte aligned), lane i receives column i of the 4 rows.
// All 64 lanes must be active.
typedef short mgx_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) mgx_s16x4* mgx_lds_s16x4_ptr;
__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((mgx_lds_s16x4_ptr)p));
}

