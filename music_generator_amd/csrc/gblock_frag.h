// Register-fragment helpers shared by the fused generator-block kernels (gblock.hip forward,
// gblock_bwd.hip data-gradient): weight fragments are 64-float records (one float per lane, stored
// four fragments to a lane-major block) streamed through two 16-register buffers while the MFMAs
// of the previous group run.
#pragma once
#include "common.h"

constexpr int FR = 64;   // floats per fragment (one per lane)

// Position of lane `lane`'s float of fragment `fr` in a packed buffer: fragments are stored in
// blocks of four, lane-major inside a block, so that a lane fetches its floats of four
// consecutive fragments with one 16-byte load.  Every fragment group starts at a multiple of 4.
__host__ __device__ constexpr int frag_slot(int fr, int lane) { return (fr >> 2) * (4 * FR) + lane * 4 + (fr & 3); }

// channel held by accumulator register s in lane half h (= contraction index of MFMA step s)
__device__ __forceinline__ int chan(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

struct WBuf {
    float v[2][16];
    float touch[2];   // destinations of the look-ahead reads of pf(); set to 0 before the first pf()
};

// Packed fragment buffers carry this many floats of slack behind their last fragment so that the
// look-ahead read below never leaves the allocation.
constexpr int FRAG_SLACK = 2048;

template <int P, int NS>
__device__ __forceinline__ void pf(WBuf& B, const float* __restrict__ frag, int lane) {
    if (NS == 16) {
        // Pull the 4 KiB behind this group (the group that most often follows it) towards the CU's
        // L1: every lane reads one dword of one of its 32 cache lines and drops it.  Issued one
        // group before the real loads of that data, it turns their L2 latency into an L1 hit.
        // The look-ahead value read two groups ago is "used" here (an empty asm), which keeps its
        // register reserved until that load has returned.
        asm volatile("" ::"v"(B.touch[P]));
        B.touch[P] = frag[16 * FR + (lane & 31) * 32];
    }
    // four fragments per 16-byte load (frag_slot): a vector-memory instruction costs the issuing
    // SIMD tens of MFMA cycles whatever its width
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) {
        const f32x4 t = *(const f32x4*)(frag + q * (4 * FR) + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) B.v[P][4 * q + i] = t[i];
    }
}

// acc += sum_{s < NS} B.v[P][s] (A operand: weights) x tile[S0 + s] (B operand: activations)
template <int P, int S0, int NS>
__device__ __forceinline__ void mma(f32x16& acc, const f32x16& tile, const WBuf& B) {
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = mfma32(B.v[P][s], tile[S0 + s], acc);
}

#define MGX_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ const float* fptr(const float* pack, int frag) { return pack + (long)frag * FR; }

// rows chan(s, h), s < NREG, of an accumulator tile -> 32-position rows of a tile-major side buffer
template <int NREG>
__device__ __forceinline__ void save_rows(float* __restrict__ buf, int row0, const f32x16& t, int r, int h) {
#pragma unroll
    for (int s = 0; s < NREG; ++s) buf[(row0 + chan(s, h)) * 32 + r] = t[s];
}
