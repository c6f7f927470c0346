// The head of the stage-0 generator (define_generator, trainer/models.py:412-444): from the latent
// tensor z (n, 1, 100, 2) through
//     featured = Conv2D(128, (1,2), strides (1,2))                               (n, 1, 50, 128)
//     per branch b = 0..3:  ConvT(4,(2,1)) -> Dense(8) -> ConvT(16,(2,1)) -> Dense(32)   (n, 4, 50, 32)
// to the four tensors the Conv2DTranspose(64, (1,15)) layers read.  Seventeen layers without an
// activation between them, 0.9 MFLOP per sample: layer at a time they are seventeen launches of a few
// hundred threads each (and four times that in the backward), which is most of what a small-batch step
// of the low stages waits for.  Every layer maps a column w of the (h, w) grid to the same column, so a
// wave takes one (sample, column) at a time through all layers, the weights (8 176 floats) in LDS.
//
// Backward: the same walk in reverse from the gradients of the four outputs; the intermediate tensors
// are re-computed from z (cheaper than reading them back), the weight-gradient sums of all 34 tensors
// stay in registers over the columns of a wave (128 per lane) and leave as one partial slab per wave;
// a second kernel adds the slabs in wave order into the gradient tensors (bitwise reproducible).
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int ZW = 100, FW = 50;                     // latent width, columns of the head
constexpr int NBR = 4;
// LDS / slab layout of the weights (floats): k0 [4][128], b0 [128], then per branch
//   k1 [8][128] (row = h*4+f), b1 [4], d1 [4][8], c1 [8], k2 [2*16][8] (row = dh*16+f), b2 [16],
//   d2 [16][33] (rows padded: lanes that differ in the row read different banks), c2 [32]
constexpr int O_K0 = 0, O_B0 = 512, O_BR = 640;
constexpr int B_K1 = 0, B_B1 = 1024, B_D1 = 1028, B_C1 = 1060, B_K2 = 1068, B_B2 = 1324, B_D2 = 1340, B_C2 = 1868;
constexpr int D2S = 33;
constexpr int BR_FLOATS = 1900;
constexpr int W_FLOATS = O_BR + NBR * BR_FLOATS;     // 8240
// per-wave scratch (floats): F[128], A1[4][2][4], A2[4][2][8], A3[4][4][16], and in the backward
// G4[4][4][32], G3[4][4][16], G2[4][2][8], G1[4][2][4]
constexpr int S_F = 0, S_A1 = 128, S_A2 = 160, S_A3 = 224, S_G4 = 480, S_G3 = 992, S_G2 = 1248, S_G1 = 1312;
constexpr int S_FLOATS = 1344;

__device__ __forceinline__ void stage_weights(float* wl, const mgx_ghead_weights& w) {
    const int t = threadIdx.x;
    for (int i = t; i < 512; i += 256) wl[O_K0 + i] = w.k0[i];
    if (t < 128) wl[O_B0 + t] = w.b0[t];
#pragma unroll
    for (int b = 0; b < NBR; ++b) {
        float* d = wl + O_BR + b * BR_FLOATS;
        for (int i = t; i < 1024; i += 256) d[B_K1 + i] = w.k1[b][i];
        if (t < 4) d[B_B1 + t] = w.b1[b][t];
        if (t < 32) d[B_D1 + t] = w.d1[b][t];
        if (t < 8) d[B_C1 + t] = w.c1[b][t];
        d[B_K2 + t] = w.k2[b][t];
        if (t < 16) d[B_B2 + t] = w.b2[b][t];
        for (int i = t; i < 512; i += 256) d[B_D2 + (i >> 5) * D2S + (i & 31)] = w.d2[b][i];
        if (t < 32) d[B_C2 + t] = w.c2[b][t];
    }
}

// LDS traffic between the lanes of ONE wave: the wave's own instructions are ordered, the data has to
// have landed
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Layers 0..3 of one column into the wave's scratch (F, A1, A2, A3).
__device__ __forceinline__ void column_forward(const float* wl, float* sc, const float* zc, int lane) {
    const float z0 = zc[0], z1 = zc[1], z2 = zc[2], z3 = zc[3];       // z[2w][0], z[2w][1], z[2w+1][0], z[2w+1][1]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane + 64 * q;
        float v = wl[O_B0 + c];
        v = fmaf(z0, wl[O_K0 + c], v);
        v = fmaf(z1, wl[O_K0 + 128 + c], v);
        v = fmaf(z2, wl[O_K0 + 256 + c], v);
        v = fmaf(z3, wl[O_K0 + 384 + c], v);
        sc[S_F + c] = v;
    }
    wave_lds_fence();
    // A1[b][h][f] = b1[f] + sum_c F[c] k1[h][f][c]: lane = half * 32 + (b*8 + h*4 + f), a half per 64 channels
    {
        const int o = lane & 31, half = lane >> 5, b = o >> 3, row = o & 7;
        const float* k = wl + O_BR + b * BR_FLOATS + B_K1 + row * 128 + half * 64;
        const float* f = sc + S_F + half * 64;
        float v = 0.f;
#pragma unroll 16
        for (int c = 0; c < 64; ++c) {
            const int cc = (c + o) & 63;             // rotated start: the 32 rows are read from 32 different banks
            v = fmaf(f[cc], k[cc], v);
        }
        const float other = __shfl_xor(v, 32, 64);
        if (half == 0) sc[S_A1 + o] = (v + other) + wl[O_BR + b * BR_FLOATS + B_B1 + (row & 3)];
    }
    wave_lds_fence();
    // A2[b][h][j] = c1[j] + sum_f A1[b][h][f] d1[f][j]: lane = b*16 + h*8 + j
    {
        const int b = lane >> 4, h = (lane >> 3) & 1, j = lane & 7;
        const float* wb = wl + O_BR + b * BR_FLOATS;
        float v = wb[B_C1 + j];
#pragma unroll
        for (int f = 0; f < 4; ++f) v = fmaf(sc[S_A1 + b * 8 + h * 4 + f], wb[B_D1 + f * 8 + j], v);
        sc[S_A2 + lane] = v;
    }
    wave_lds_fence();
    // A3[b][2h+dh][f] = b2[f] + sum_j A2[b][h][j] k2[dh][f][j]: 256 outputs, 4 per lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = lane + 64 * q, b = o >> 6, hh = (o >> 4) & 3, f = o & 15;
        const float* wb = wl + O_BR + b * BR_FLOATS;
        float v = wb[B_B2 + f];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v = fmaf(sc[S_A2 + b * 16 + (hh >> 1) * 8 + j], wb[B_K2 + ((hh & 1) * 16 + f) * 8 + j], v);
        sc[S_A3 + o] = v;
    }
    wave_lds_fence();
}

__global__ __launch_bounds__(256) void ghead_fwd_kernel(mgx_ghead_weights w, const float* __restrict__ z, long ncol,
                                                        mgx_ghead_tensors out) {
    __shared__ float wl[W_FLOATS];
    __shared__ float scratch[4][S_FLOATS];
    stage_weights(wl, w);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* sc = scratch[wv];
    for (long col = (long)blockIdx.x * 4 + wv; col < ncol; col += (long)gridDim.x * 4) {
        const long s = col / FW;
        const int x = (int)(col - s * FW);
        column_forward(wl, sc, z + (s * ZW + 2 * x) * 2, lane);
        // A4[b][hh][m] = c2[m] + sum_f A3[b][hh][f] d2[f][m]: lane = half * 32 + m, 8 (b, hh) pairs per half
        const int m = lane & 31, half = lane >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pr = 2 * p + half, b = pr >> 2, hh = pr & 3;
            const float* wb = wl + O_BR + b * BR_FLOATS;
            float v = wb[B_C2 + m];
#pragma unroll
            for (int f = 0; f < 16; ++f) v = fmaf(sc[S_A3 + b * 64 + hh * 16 + f], wb[B_D2 + f * D2S + m], v);
            out.y[b][((s * 4 + hh) * FW + x) * 32 + m] = v;
        }
        wave_lds_fence();                    // the next column overwrites the scratch
    }
}

// ------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------
// Which weight-gradient sums a lane keeps (128 registers), in the order they are written to the slab:
//   k1: 32 rows (b, h, f) x channels lane, lane+64                 -> acc[0..63]   (row r: acc[2r], acc[2r+1])
//   d2: per branch 16 x 32 = 512 elements, element e = lane + 64 i -> acc[64 + 8 b + i], i < 8
//   k2: per branch 256 elements, e = lane + 64 i                   -> acc[96 + 4 b + i], i < 4
//   k0: 512 elements, e = lane + 64 i                              -> acc[112 + i], i < 8
//   b0: 128, e = lane + 64 i                                       -> acc[120 + i], i < 2
//   d1: 4 x 32 = 128 elements, e = lane + 64 i                     -> acc[122 + i], i < 2
//   c2: 4 x 32 = 128, e = lane + 64 i                              -> acc[124 + i], i < 2
//   b2: 4 x 16 = 64, e = lane                                      -> acc[126]
//   c1 (4 x 8 = 32, lanes 0..31) and b1 (4 x 4 = 16, lanes 32..47) -> acc[127]
constexpr int SLAB = 128 * 64;

__global__ __launch_bounds__(256) void ghead_bwd_kernel(mgx_ghead_weights w, const float* __restrict__ z, long ncol,
                                                        mgx_ghead_tensors dout, float* __restrict__ slabs,
                                                        float* __restrict__ dz) {
    __shared__ float wl[W_FLOATS];
    __shared__ float scratch[4][S_FLOATS];
    stage_weights(wl, w);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* sc = scratch[wv];
    float acc[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) acc[i] = 0.f;
    for (long col = (long)blockIdx.x * 4 + wv; col < ncol; col += (long)gridDim.x * 4) {
        const long s = col / FW;
        const int x = (int)(col - s * FW);
        const float* zc = z + (s * ZW + 2 * x) * 2;
        column_forward(wl, sc, zc, lane);
        // ---- G4 = d(out) in registers: lane = half * 32 + m holds pairs (b, hh) = 2p + half ----
        const int m = lane & 31, half = lane >> 5;
        float g4[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pr = 2 * p + half, b = pr >> 2, hh = pr & 3;
            g4[p] = dout.y[b][((s * 4 + hh) * FW + x) * 32 + m];
            sc[S_G4 + pr * 32 + m] = g4[p];
        }
        // c2[b][m] += sum_hh G4: pairs of one branch are p = 2b', 2b'+1 (this half) and the other half's
#pragma unroll
        for (int b = 0; b < NBR; ++b) {
            const float mine = g4[2 * b] + g4[2 * b + 1];
            const float sum = mine + __shfl_xor(mine, 32, 64);
            // element e = b*32 + m = lane + 64 i  <=>  i = b >> 1, lane = (b & 1) * 32 + m
            if (half == (b & 1)) acc[124 + (b >> 1)] += sum;
        }
        wave_lds_fence();
        // G3[b][hh][f] = sum_m G4[b][hh][m] d2[f][m]: 256 outputs, 4 per lane (lane = hh*16 + f, one branch per pass)
#pragma unroll
        for (int b = 0; b < NBR; ++b) {
            const int hh = lane >> 4, f = lane & 15;
            const float* wb = wl + O_BR + b * BR_FLOATS + B_D2 + f * D2S;
            const float* g = sc + S_G4 + (b * 4 + hh) * 32;
            float v = 0.f;
#pragma unroll
            for (int mm = 0; mm < 32; ++mm) v = fmaf(g[mm], wb[mm], v);
            sc[S_G3 + b * 64 + lane] = v;
        }
        // d2 sums: element e = f*32 + m' owned by lane = e & 63, i = e >> 6, i.e. f = 2i + (lane >> 5), m' = lane & 31:
        // the owner needs G4[b][hh][m'] of BOTH halves' pairs -> exchange through the other half
#pragma unroll
        for (int b = 0; b < NBR; ++b) {
            // this lane holds G4 for hh = half and half + 2 of branch b (p = 2b, 2b+1); the partner lane the others
            const float ga = g4[2 * b], gb = g4[2 * b + 1];
            const float oa = __shfl_xor(ga, 32, 64), ob = __shfl_xor(gb, 32, 64);
            const float gh[4] = {half ? oa : ga, half ? ga : oa, half ? ob : gb, half ? gb : ob};   // G4[b][hh][m], hh = 0..3
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = 2 * i + half;
                float v = acc[64 + 8 * b + i];
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) v = fmaf(sc[S_A3 + b * 64 + hh * 16 + f], gh[hh], v);
                acc[64 + 8 * b + i] = v;
            }
        }
        wave_lds_fence();
        // b2[b][f] += sum_hh G3[b][hh][f]: e = b*16 + f = lane
        {
            const int b = lane >> 4, f = lane & 15;
            float v = acc[126];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) v += sc[S_G3 + b * 64 + hh * 16 + f];
            acc[126] = v;
        }
        // k2[b][dh][f][j] += sum_h A2[b][h][j] G3[b][2h+dh][f]: e = (dh*16+f)*8 + j = lane + 64 i
#pragma unroll
        for (int b = 0; b < NBR; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = lane + 64 * i, j = e & 7, f = (e >> 3) & 15, dh = e >> 7;
                float v = acc[96 + 4 * b + i];
#pragma unroll
                for (int h = 0; h < 2; ++h) v = fmaf(sc[S_A2 + b * 16 + h * 8 + j], sc[S_G3 + b * 64 + (2 * h + dh) * 16 + f], v);
                acc[96 + 4 * b + i] = v;
            }
        // G2[b][h][j] = sum_dh sum_f G3[b][2h+dh][f] k2[dh][f][j]: lane = b*16 + h*8 + j
        {
            const int b = lane >> 4, h = (lane >> 3) & 1, j = lane & 7;
            const float* wb = wl + O_BR + b * BR_FLOATS;
            float v = 0.f;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int f = 0; f < 16; ++f)
                    v = fmaf(sc[S_G3 + b * 64 + (2 * h + dh) * 16 + f], wb[B_K2 + (dh * 16 + f) * 8 + j], v);
            sc[S_G2 + lane] = v;
        }
        wave_lds_fence();
        // c1[b][j] += sum_h G2 (lanes 0..31: b = lane >> 3, j = lane & 7); d1[b][f][j] += sum_h A1[b][h][f] G2[b][h][j]
        if (lane < 32) {
            const int b = lane >> 3, j = lane & 7;
            acc[127] += sc[S_G2 + b * 16 + j] + sc[S_G2 + b * 16 + 8 + j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, b = e >> 5, f = (e >> 3) & 3, j = e & 7;
            float v = acc[122 + i];
#pragma unroll
            for (int h = 0; h < 2; ++h) v = fmaf(sc[S_A1 + b * 8 + h * 4 + f], sc[S_G2 + b * 16 + h * 8 + j], v);
            acc[122 + i] = v;
        }
        // G1[b][h][f] = sum_j G2[b][h][j] d1[f][j]: lanes 0..31 = b*8 + h*4 + f
        if (lane < 32) {
            const int b = lane >> 3, h = (lane >> 2) & 1, f = lane & 3;
            const float* wb = wl + O_BR + b * BR_FLOATS;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v = fmaf(sc[S_G2 + b * 16 + h * 8 + j], wb[B_D1 + f * 8 + j], v);
            sc[S_G1 + lane] = v;
        }
        wave_lds_fence();
        // b1[b][f] += sum_h G1 (lanes 32..47: b = (lane-32) >> 2, f = lane & 3)
        if (lane >= 32 && lane < 48) {
            const int b = (lane - 32) >> 2, f = lane & 3;
            acc[127] += sc[S_G1 + b * 8 + f] + sc[S_G1 + b * 8 + 4 + f];
        }
        // k1[b][h][f][c] += F[c] G1[b][h][f]  and  GF[c] = sum_rows G1[row] k1[row][c], c = lane, lane + 64
        {
            const float f0 = sc[S_F + lane], f1 = sc[S_F + 64 + lane];
            float gf0 = 0.f, gf1 = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const float g = sc[S_G1 + r];
                acc[2 * r] = fmaf(f0, g, acc[2 * r]);
                acc[2 * r + 1] = fmaf(f1, g, acc[2 * r + 1]);
                const float* k = wl + O_BR + (r >> 3) * BR_FLOATS + B_K1 + (r & 7) * 128;
                gf0 = fmaf(g, k[lane], gf0);
                gf1 = fmaf(g, k[64 + lane], gf1);
            }
            // b0[c] += GF[c]; k0[j][c] += z[j] GF[c]: k0 element e = j*128 + c = lane + 64 i -> j = i >> 1, c = lane + 64 (i & 1)
            acc[120] += gf0;
            acc[121] += gf1;
            const float z0 = zc[0], z1 = zc[1], z2 = zc[2], z3 = zc[3];
            acc[112] = fmaf(z0, gf0, acc[112]);
            acc[113] = fmaf(z0, gf1, acc[113]);
            acc[114] = fmaf(z1, gf0, acc[114]);
            acc[115] = fmaf(z1, gf1, acc[115]);
            acc[116] = fmaf(z2, gf0, acc[116]);
            acc[117] = fmaf(z2, gf1, acc[117]);
            acc[118] = fmaf(z3, gf0, acc[118]);
            acc[119] = fmaf(z3, gf1, acc[119]);
            if (dz) {
                // d(loss)/dz[2x+dw][ci] = sum_c GF[c] k0[dw*2+ci][c]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = fmaf(gf0, wl[O_K0 + j * 128 + lane], gf1 * wl[O_K0 + j * 128 + 64 + lane]);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                    if (lane == 0) dz[(s * ZW + 2 * x) * 2 + j] = v;
                }
            }
        }
        wave_lds_fence();
    }
    if (!slabs) return;
    float* slab = slabs + ((long)blockIdx.x * 4 + wv) * SLAB;
#pragma unroll
    for (int i = 0; i < 128; ++i) slab[i * 64 + lane] = acc[i];
}

// grads += sum over the slabs.  A block of 1024 threads covers 64 consecutive slots of a slab x 16 groups of
// slabs (group q adds slabs q, q+16, ... in order); the 16 group sums are added in a fixed tree.
__global__ __launch_bounds__(1024) void ghead_reduce_kernel(const float* __restrict__ slabs, int nslabs, mgx_ghead_grads g) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x;                              // register index of the slab: slot = i * 64 + lane
    float v = 0.f;
    for (int s = q; s < nslabs; s += 16) v += slabs[(long)s * SLAB + i * 64 + lane];
    red[q][lane] = v;
    __syncthreads();
    if (q != 0) return;
    {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = red[k][lane];
#pragma unroll
        for (int wd = 8; wd > 0; wd >>= 1)
#pragma unroll
            for (int k = 0; k < wd; ++k) t[k] += t[k + wd];
        v = t[0];
    }
    if (i < 64) {                                          // k1: row r = i >> 1 = b*8 + h*4 + f, channel lane + 64 (i & 1)
        const int r = i >> 1;
        g.k1[r >> 3][(r & 7) * 128 + lane + 64 * (i & 1)] += v;
    } else if (i < 96) {                                   // d2
        const int b = (i - 64) >> 3, e = lane + 64 * ((i - 64) & 7);
        g.d2[b][e] += v;
    } else if (i < 112) {                                  // k2
        const int b = (i - 96) >> 2, e = lane + 64 * ((i - 96) & 3);
        g.k2[b][e] += v;
    } else if (i < 120) {
        g.k0[lane + 64 * (i - 112)] += v;
    } else if (i < 122) {
        g.b0[lane + 64 * (i - 120)] += v;
    } else if (i < 124) {                                  // d1: e = b*32 + f*8 + j
        const int e = lane + 64 * (i - 122);
        g.d1[e >> 5][e & 31] += v;
    } else if (i < 126) {                                  // c2: e = b*32 + m
        const int e = lane + 64 * (i - 124);
        g.c2[e >> 5][e & 31] += v;
    } else if (i == 126) {                                 // b2: e = b*16 + f
        g.b2[lane >> 4][lane & 15] += v;
    } else {
        if (lane < 32) g.c1[lane >> 3][lane & 7] += v;
        else if (lane < 48) g.b1[(lane - 32) >> 2][lane & 3] += v;
    }
}

// forward: a column per wave and pass; backward: at least 4 columns per wave, so that a slab of partial sums
// (32 KiB) is written per 4 columns or more
unsigned ghead_grid(long ncol, int cols_per_wave) {
    const long wgs = (ncol + 4 * cols_per_wave - 1) / (4 * cols_per_wave);
    return (unsigned)(wgs < 256 ? wgs : 256);
}

bool weights_ok(const mgx_ghead_weights* w) {
    if (!w || !w->k0 || !w->b0) return false;
    for (int b = 0; b < NBR; ++b)
        if (!w->k1[b] || !w->b1[b] || !w->d1[b] || !w->c1[b] || !w->k2[b] || !w->b2[b] || !w->d2[b] || !w->c2[b])
            return false;
    return true;
}

}  // namespace

extern "C" int mgx_ghead_fwd_f32(const mgx_ghead_weights* w, const float* z, int64_t nimg, const mgx_ghead_tensors* out,
                                 void* stream) {
    if (!weights_ok(w) || !z || !out || nimg <= 0) return MGX_ERR_ARG;
    for (int b = 0; b < NBR; ++b)
        if (!out->y[b]) return MGX_ERR_ARG;
    const long ncol = (long)nimg * FW;
    hipLaunchKernelGGL(ghead_fwd_kernel, dim3(ghead_grid(ncol, 1)), dim3(256), 0, (hipStream_t)stream, *w, z, ncol, *out);
    return mgx_launch_status();
}

extern "C" int64_t mgx_ghead_bwd_workspace(int64_t nimg) {
    if (nimg <= 0) return -1;
    return (int64_t)ghead_grid((long)nimg * FW, 4) * 4 * SLAB * (int64_t)sizeof(float);
}

extern "C" int mgx_ghead_bwd_f32(const mgx_ghead_weights* w, const float* z, int64_t nimg, const mgx_ghead_tensors* dout,
                                 const mgx_ghead_grads* g, float* dz, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    if (!weights_ok(w) || !z || !dout || nimg <= 0 || (!g && !dz)) return MGX_ERR_ARG;
    for (int b = 0; b < NBR; ++b)
        if (!dout->y[b]) return MGX_ERR_ARG;
    if (g) {
        if (!g->k0 || !g->b0 || !workspace) return MGX_ERR_ARG;
        for (int b = 0; b < NBR; ++b)
            if (!g->k1[b] || !g->b1[b] || !g->d1[b] || !g->c1[b] || !g->k2[b] || !g->b2[b] || !g->d2[b] || !g->c2[b])
                return MGX_ERR_ARG;
        if (workspace_bytes < mgx_ghead_bwd_workspace(nimg)) return MGX_ERR_WORKSPACE;
    }
    const long ncol = (long)nimg * FW;
    const unsigned grid = ghead_grid(ncol, 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ghead_bwd_kernel, dim3(grid), dim3(256), 0, st, *w, z, ncol, *dout, g ? (float*)workspace : nullptr,
                       dz);
    if (g)
        hipLaunchKernelGGL(ghead_reduce_kernel, dim3(128), dim3(1024), 0, st, (const float*)workspace, (int)grid * 4, *g);
    return mgx_launch_status();
}
