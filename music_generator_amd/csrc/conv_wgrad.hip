// Weight / bias gradient of the implicit-GEMM contraction on v_mfma_f32_32x32x2_f32 (gfx950).
//
// dW[tap][k][n] = sum over GEMM rows r of X[gather(r,tap)][k] * dY[scatter(r,n)].
// The contraction runs over rows, two per MFMA (lane half h takes row r+h); both operands are
// read straight from global memory with the channel index on the lane, i.e. 128-byte coalesced
// segments (NHWC keeps the channel innermost).  Rows are split over `splits` waves; every wave
// keeps T*KT*NT accumulator tiles of 32(k) x 32(n) in registers for its whole row range and
// stores them once as a partial slab.  A second kernel adds the slabs in split order (fixed
// order => bitwise reproducible) into dW with the weight tensor's own strides.
#include "common.h"
#include "../../include/mgx.h"

bool mgx_twgrad_applicable(const mgx_conv_desc& d);
bool mgx_wconv_enabled();                                   // conv_gemm.hip
bool mgx_wwgrad_applicable(const mgx_conv_desc& d);         // wwgrad.hip
long mgx_wwgrad_splits(const mgx_conv_desc& d);
long mgx_wwgrad_du_floats(const mgx_conv_desc& d);
int mgx_wwgrad_launch(const mgx_conv_desc& d, const float* x, const float* dy, const float* dymul, float* partial,
                      float* bpartial, hipStream_t st);
int mgx_wwgrad_finish(const mgx_conv_desc& d, const float* du, float* dw, hipStream_t st);
long mgx_twgrad_splits(const mgx_conv_desc& d);
int mgx_twgrad_launch(const mgx_conv_desc& d, const float* x, const float* dy, const float* dymul, float* partial,
                      float* bpartial, hipStream_t st);

namespace {

struct WgradPlan {
    int T, KT, NT;          // tiles per wave: taps x k-tiles x n-tiles
    int ntaps, ktot, ntot;  // totals
    int tap_groups, k_groups, n_groups;
    long splits;            // multiple of 4
    long rows_per_split;    // even
    long tiles_total;
    long partial_floats, bias_floats;
    bool tiled;             // LDS-staged tile kernel (tconv.hip)
    bool wino;              // Winograd kernel (wwgrad.hip): 16 transform-domain tiles per 32-channel slice
    long du_floats;         // its transform-domain gradient between slab reduction and final transform
};

WgradPlan make_plan(const mgx_conv_desc& d) {
    WgradPlan p;
    p.ntaps = d.kh * d.kw;
    p.ktot = (d.cin + 31) / 32;
    p.ntot = (d.ngemm + 31) / 32;
    if (p.ntaps % 9 == 0) p.T = 9;
    else if (p.ntaps % 8 == 0) p.T = 8;
    else if (p.ntaps % 4 == 0) p.T = 4;
    else p.T = 1;
    if (p.T > 1) { p.KT = 1; p.NT = 1; }
    else {
        p.KT = p.ktot >= 2 ? 2 : 1;
        p.NT = p.ntot >= 4 ? 4 : (p.ntot >= 2 ? 2 : 1);
    }
    p.wino = mgx_wconv_enabled() && mgx_wwgrad_applicable(d);
    p.du_floats = 0;
    if (p.wino) {
        p.tiled = false;
        p.T = 16; p.KT = 1; p.NT = 1;
        p.tap_groups = 1; p.k_groups = p.ktot; p.n_groups = 1;
        p.splits = mgx_wwgrad_splits(d);
        p.rows_per_split = 0;
        p.tiles_total = 16L * p.ktot;
        p.partial_floats = p.splits * p.tiles_total * 1024;
        p.bias_floats = p.splits * 64;
        p.du_floats = mgx_wwgrad_du_floats(d);
        return p;
    }
    p.tiled = mgx_twgrad_applicable(d);
    if (p.tiled) {
        p.T = p.ntaps; p.KT = 1; p.NT = 1;
        p.tap_groups = 1; p.k_groups = p.ktot; p.n_groups = 1;
        p.splits = mgx_twgrad_splits(d);
        p.rows_per_split = 0;
        p.tiles_total = (long)p.ntaps * p.ktot * p.ntot;
        p.partial_floats = p.splits * p.tiles_total * 1024;
        p.bias_floats = p.splits * p.ntot * 64;
        return p;
    }
    p.tap_groups = p.ntaps / p.T;
    p.k_groups = (p.ktot + p.KT - 1) / p.KT;
    p.n_groups = (p.ntot + p.NT - 1) / p.NT;
    const long groups = (long)p.tap_groups * p.k_groups * p.n_groups;
    const long R = (long)d.nimg * d.hout * d.wout;
    long s = 2048 / groups;
    if (s < 4) s = 4;
    const long smax = (R + 63) / 64;  // at least 64 rows per split
    if (s > smax) s = smax;
    s = ((s + 3) / 4) * 4;
    p.splits = s;
    long rps = (R + s - 1) / s;
    rps += rps & 1;
    p.rows_per_split = rps;
    p.tiles_total = (long)p.ntaps * p.ktot * p.ntot;
    p.partial_floats = p.splits * p.tiles_total * 1024;
    p.bias_floats = p.splits * p.ntot * 64;
    return p;
}

// one tap, stride 1, no padding, no depth-to-space: rows of X and dY are plain offsets
bool dense_layer(const mgx_conv_desc& d) {
    return d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.pl == 0 && d.dkh == 1 && d.dkw == 1 &&
           d.hin == d.hout && d.win == d.wout && d.cout == d.ngemm;
}

template <int T, int KT, int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                    const float* __restrict__ dy,
                                                    const float* __restrict__ dymul, float* __restrict__ partial,
                                                    float* __restrict__ bpartial, int tap_groups, int k_groups,
                                                    int ktot, int ntot, long rows_per_split, long tiles_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long split = (long)blockIdx.x * 4 + wv;
    int g = blockIdx.y;
    const int tg = g % tap_groups; g /= tap_groups;
    const int kg = g % k_groups;
    const int ng = g / k_groups;
    const int tap0 = tg * T, kt0 = kg * KT, nt0 = ng * NT;
    const long R = (long)d.nimg * d.hout * d.wout;
    const long rbeg = split * rows_per_split;
    long rend = rbeg + rows_per_split;
    if (rend > R) rend = R;

    f32x16 acc[T][KT][NT];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][kt][nt][i] = 0.f;
    float bsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

    // column (n) part of the dY address, fixed per lane
    const long out_w = (long)d.wout * d.dkw;
    long npart[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (nt0 + nt) * 32 + r;
        nok[nt] = (nt0 + nt) < ntot && n < d.ngemm;
        const int nn = nok[nt] ? n : 0;
        const int ab = nn / d.cout, co = nn - ab * d.cout;
        const int a2 = ab / d.dkw, b2 = ab - a2 * d.dkw;
        npart[nt] = (a2 * out_w + b2) * d.cout + co;
    }
    int kcol[KT];
    bool kok[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        kcol[kt] = (kt0 + kt) * 32 + r;
        kok[kt] = (kt0 + kt) < ktot && kcol[kt] < d.cin;
    }

    if (rbeg < rend) {
        // position of this lane's first row (rbeg + h), then advance by 2 rows per step; the operands
        // of the next step are loaded before the MFMAs of the current one (the loads are per-lane
        // dwords straight from global memory: their latency must not sit between every two steps)
        long row = rbeg + h;
        const int hw = d.hout * d.wout;
        long img = row / hw;
        int rem = (int)(row - img * hw);
        int ho = rem / d.wout;
        int wo = rem - ho * d.wout;
        auto load = [&](float (&bv)[NT], float (&av)[T][KT]) {
            const bool rv = row < rend;
            const long ybase = ((img * d.hout + ho) * d.dkh * out_w + (long)wo * d.dkw) * d.cout;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v = 0.f;
                if (rv && nok[nt]) {
                    v = dy[ybase + npart[nt]];
                    if (dymul) v *= dymul[ybase + npart[nt]];
                }
                bv[nt] = v;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int tap = tap0 + t;
                const int a = tap / d.kw, b = tap - a * d.kw;
                const int hi = ho * d.sh + a - d.pt;
                const int wi = wo * d.sw + b - d.pl;
                const bool ok = rv && hi >= 0 && hi < d.hin && wi >= 0 && wi < d.win;
                const long xoff = ((img * d.hin + hi) * d.win + wi) * (long)d.cin;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) av[t][kt] = (ok && kok[kt]) ? x[xoff + kcol[kt]] : 0.f;
            }
            row += 2;
            wo += 2;
            while (wo >= d.wout) { wo -= d.wout; ++ho; }
            while (ho >= d.hout) { ho -= d.hout; ++img; }
        };
        auto mma = [&](const float (&bv)[NT], const float (&av)[T][KT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bsum[nt] += bv[nt];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][kt][nt] = mfma32(av[t][kt], bv[nt], acc[t][kt][nt]);
        };
        const long steps = (rend - rbeg + 1) / 2;          // row pairs (the same for both lane halves)
        float bv0[NT], av0[T][KT], bv1[NT], av1[T][KT];
        load(bv0, av0);
        for (long s2 = 0; s2 < steps; s2 += 2) {
            if (s2 + 1 < steps) load(bv1, av1);
            mma(bv0, av0);
            if (s2 + 1 < steps) {
                if (s2 + 2 < steps) load(bv0, av0);
                mma(bv1, av1);
            }
        }
    }

    // ---- store partial tiles: partial[((split*tiles_total + tile)*16 + reg)*64 + lane] ----
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if ((kt0 + kt) >= ktot || (nt0 + nt) >= ntot) continue;
                const long tile = ((long)(tap0 + t) * ktot + (kt0 + kt)) * ntot + (nt0 + nt);
                float* dst = partial + ((split * tiles_total + tile) * 16) * 64 + lane;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) dst[reg * 64] = acc[t][kt][nt][reg];
            }
    if (bpartial && tg == 0 && kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if ((nt0 + nt) < ntot) bpartial[(split * ntot + (nt0 + nt)) * 64 + lane] = bsum[nt];
    }
}

// The same for a Dense layer (one tap, stride 1, no padding, no scatter): row r of X and of dY are plain
// offsets r * cin and r * ngemm, so a step costs its loads and MFMAs only, and four steps of operands are in
// flight ahead of the MFMAs (per-lane dword loads straight from global memory: one step ahead leaves most of
// their latency exposed when a wave has ~50 steps in all).  Same slabs, same order of summation per wave.
template <int KT, int NT>
__global__ __launch_bounds__(256) void dense_wgrad_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ dymul, float* __restrict__ partial,
                                                          float* __restrict__ bpartial, int k_groups, int ktot, int ntot,
                                                          long rows_per_split, long tiles_total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long split = (long)blockIdx.x * 4 + wv;
    const int kg = blockIdx.y % k_groups, ng = blockIdx.y / k_groups;
    const int kt0 = kg * KT, nt0 = ng * NT;
    const long R = (long)d.nimg * d.hout * d.wout;
    const long rbeg = split * rows_per_split;
    long rend = rbeg + rows_per_split;
    if (rend > R) rend = R;

    f32x16 acc[KT][NT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kt][nt][i] = 0.f;
    float bsum[NT];
    int ncol[NT], kcol[KT];
    bool nok[NT], kok[KT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bsum[nt] = 0.f;
        ncol[nt] = (nt0 + nt) * 32 + r;
        nok[nt] = (nt0 + nt) < ntot && ncol[nt] < d.ngemm;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        kcol[kt] = (kt0 + kt) * 32 + r;
        kok[kt] = (kt0 + kt) < ktot && kcol[kt] < d.cin;
    }
    if (rbeg < rend) {
        constexpr int DEPTH = 4;
        long row = rbeg + h;
        const float* xp = x + row * d.cin;
        const float* yp = dy + row * d.ngemm;
        const float* mp = dymul ? dymul + row * d.ngemm : nullptr;
        const long xs = 2L * d.cin, ys = 2L * d.ngemm;
        float bv[DEPTH][NT], av[DEPTH][KT];
        auto load = [&](float (&b)[NT], float (&a)[KT]) {
            const bool rv = row < rend;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float v = 0.f;
                if (rv && nok[nt]) {
                    v = yp[ncol[nt]];
                    if (mp) v *= mp[ncol[nt]];
                }
                b[nt] = v;
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) a[kt] = (rv && kok[kt]) ? xp[kcol[kt]] : 0.f;
            row += 2;
            xp += xs;
            yp += ys;
            if (mp) mp += ys;
        };
        auto mma = [&](const float (&b)[NT], const float (&a)[KT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bsum[nt] += b[nt];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = mfma32(a[kt], b[nt], acc[kt][nt]);
        };
        const long steps = (rend - rbeg + 1) / 2;          // row pairs (the same for both lane halves)
#pragma unroll
        for (int q = 0; q < DEPTH - 1; ++q) load(bv[q], av[q]);       // rows past the end load zeros
        for (long s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                load(bv[(q + DEPTH - 1) % DEPTH], av[(q + DEPTH - 1) % DEPTH]);
                if (s0 + q < steps) mma(bv[q], av[q]);
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if ((kt0 + kt) >= ktot || (nt0 + nt) >= ntot) continue;
            const long tile = (long)(kt0 + kt) * ntot + (nt0 + nt);
            float* dst = partial + ((split * tiles_total + tile) * 16) * 64 + lane;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) dst[reg * 64] = acc[kt][nt][reg];
        }
    if (bpartial && kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if ((nt0 + nt) < ntot) bpartial[(split * ntot + (nt0 + nt)) * 64 + lane] = bsum[nt];
    }
}

// dW[tap*wtap + k*wsk + n*wsn] += sum_split partial[...].  A block of 1024 threads covers 64
// consecutive elements x 16 split groups (group g adds splits g, g+16, ... in order); the 16 group
// sums are then added in a fixed tree, so the result does not depend on scheduling.
// Blocks wblocks and up add the bias partials instead, so that a weight gradient ends in one reduction launch,
// not two: db[c] += sum over n == c (mod nbias), splits, lane halves; four bias elements a block, 256 threads each
// (per-thread sums over its splits in double, wave sums, then the four waves in a fixed tree).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(mgx_conv_desc d, const float* __restrict__ partial,
                                                            float* __restrict__ dw, long splits, int ktot, int ntot,
                                                            long tiles_total, const float* __restrict__ bpartial,
                                                            float* __restrict__ db, int wblocks, int b_ngemm,
                                                            int b_nbias, int b_ntot) {
    if ((int)blockIdx.x >= wblocks) {
        __shared__ double bred[16];
        const int q = threadIdx.x >> 8, t = threadIdx.x & 255;
        const int c = ((int)blockIdx.x - wblocks) * 4 + q;
        double s = 0.0;
        if (c < b_nbias)
            for (int n = c; n < b_ngemm; n += b_nbias) {
                const int nt = n >> 5, j = n & 31;
                for (long sp = t; sp < splits; sp += 256) {
                    const float* src = bpartial + (sp * b_ntot + nt) * 64;
                    s += (double)src[j] + (double)src[j + 32];
                }
            }
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) bred[threadIdx.x >> 6] = s;
        __syncthreads();
        if (t == 0 && c < b_nbias) db[c] += (float)((bred[4 * q] + bred[4 * q + 1]) + (bred[4 * q + 2] + bred[4 * q + 3]));
        return;
    }
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + e;
    const long per_tap = (long)d.cin * d.ngemm;
    const long total = (long)d.kh * d.kw * per_tap;
    float s = 0.f;
    int tap = 0, k = 0, n = 0;
    if (idx < total) {
        tap = (int)(idx / per_tap);
        const int rem = (int)(idx - tap * per_tap);
        k = rem / d.ngemm;
        n = rem - k * d.ngemm;  // consecutive threads -> consecutive n
        const long tile = ((long)tap * ktot + (k >> 5)) * ntot + (n >> 5);
        const int i = k & 31, j = n & 31;
        const int lane = j + 32 * ((i >> 2) & 1);
        const int reg = (i & 3) + 4 * (i >> 3);
        const float* src = partial + (tile * 16 + reg) * 64 + lane;
        const long stride = tiles_total * 1024;
        for (long sp = g; sp < splits; sp += 16) s += src[sp * stride];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && idx < total) {
        float t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = red[q][e];
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int q = 0; q < w; ++q) t[q] += t[q + w];
        dw[(long)tap * d.wtap + (long)k * d.wsk + (long)n * d.wsn] += t[0];
    }
}

template <int T, int KT, int NT>
void launch_wgrad(const mgx_conv_desc& d, const WgradPlan& p, const float* x, const float* dy, const float* dymul,
                  float* partial, float* bpartial, hipStream_t st) {
    dim3 grid((unsigned)(p.splits / 4), (unsigned)(p.tap_groups * p.k_groups * p.n_groups)), block(256);
    hipLaunchKernelGGL((wgrad_kernel<T, KT, NT>), grid, block, 0, st, d, x, dy, dymul, partial, bpartial,
                       p.tap_groups, p.k_groups, p.ktot, p.ntot, p.rows_per_split, p.tiles_total);
}

// ------------------------------------------------------------------------------------------
// Weight / bias gradient of a 1x1 convolution with a 2-channel input (from-audio layers): three
// sums per output channel over all positions, on the vector ALUs at the HBM read rate.  A thread
// owns 4 channels and a strided share of the block's positions; the block's partial sums go to the
// workspace and a second kernel adds the blocks in order (bitwise reproducible).
// ------------------------------------------------------------------------------------------
template <bool MUL>
__global__ __launch_bounds__(256) void pointwise2_wgrad_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy,
                                                               const float* __restrict__ dymul,
                                                               float* __restrict__ partial, long npos, int n,
                                                               int pos_per_block) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [3][pstep][n]
    const int q = n >> 2;
    const int c = (threadIdx.x % q) * 4, pl = threadIdx.x / q, pstep = 256 / q;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, ab = a0;
    const long p0 = (long)blockIdx.x * pos_per_block;
    const long p1 = p0 + pos_per_block < npos ? p0 + pos_per_block : npos;
#pragma unroll 4
    for (long p = p0 + pl; p < p1; p += pstep) {
        f32x4 g = __builtin_nontemporal_load((const f32x4*)(dy + p * n + c));
        if (MUL) g *= __builtin_nontemporal_load((const f32x4*)(dymul + p * n + c));
        const float2 xv = *(const float2*)(x + p * 2);
        a0 += g * xv.x;
        a1 += g * xv.y;
        ab += g;
    }
    *(f32x4*)(red + (0 * pstep + pl) * n + c) = a0;
    *(f32x4*)(red + (1 * pstep + pl) * n + c) = a1;
    *(f32x4*)(red + (2 * pstep + pl) * n + c) = ab;
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * n; e += 256) {
        const int k = e / n, col = e - k * n;
        float s = 0.f;
        for (int j = 0; j < pstep; ++j) s += red[(k * pstep + j) * n + col];
        partial[(long)blockIdx.x * 3 * n + e] = s;
    }
}

// A block of 1024 threads covers 16 consecutive sums x 64 groups of partial blocks (group g adds blocks g,
// g+64, ... in order); the 64 group sums are then added in a fixed order, so the result does not depend on
// scheduling.  (One thread per sum walks ~10^3 dependent loads: 230 us for a 128-channel layer.)
__global__ __launch_bounds__(1024) void pointwise2_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                       float* __restrict__ dw,
                                                                       float* __restrict__ dbias, int nblocks, int n) {
    __shared__ float red[64][16];
    const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float s = 0.f;
    if (e < 3 * n)
        for (int b = g; b < nblocks; b += 64) s += partial[(long)b * 3 * n + e];
    red[g][el] = s;
    __syncthreads();
    if (g == 0 && e < 3 * n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 64; ++q) t += red[q][el];
        if (e < 2 * n) dw[e] += t;
        else if (dbias) dbias[e - 2 * n] += t;
    }
}

bool pointwise2_wgrad_applicable(const mgx_conv_desc& d) {
    return d.cin == 2 && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.dkh == 1 && d.dkw == 1 &&
           d.pt == 0 && d.pl == 0 && d.hout == d.hin && d.wout == d.win && d.ngemm == d.cout && d.ngemm % 4 == 0 &&
           256 % (d.ngemm / 4) == 0 && d.ngemm <= 1024 && d.wsn == 1 && d.wsk == d.ngemm &&
           (d.nbias == d.ngemm || d.nbias == 0);
}

struct Pw2Plan {
    long npos, ppb;
    int blocks, pstep;
};

Pw2Plan pointwise2_wgrad_plan(const mgx_conv_desc& d) {
    Pw2Plan p;
    p.npos = (long)d.nimg * d.hin * d.win;
    p.pstep = 256 / (d.ngemm / 4);
    long ppb = (p.npos + 1023) / 1024;
    ppb = ((ppb + p.pstep - 1) / p.pstep) * p.pstep;
    if (ppb < p.pstep) ppb = p.pstep;
    p.ppb = ppb;
    p.blocks = (int)((p.npos + ppb - 1) / ppb);
    return p;
}

}  // namespace

extern "C" int64_t mgx_conv_wgrad_workspace(const mgx_conv_desc* dp) {
    if (!dp) return -1;
    if (pointwise2_wgrad_applicable(*dp))
        return (int64_t)pointwise2_wgrad_plan(*dp).blocks * 3 * dp->ngemm * (int64_t)sizeof(float);
    const WgradPlan p = make_plan(*dp);
    return (p.partial_floats + p.bias_floats + p.du_floats) * (int64_t)sizeof(float);
}

extern "C" int mgx_conv_wgrad_f32(const mgx_conv_desc* dp, const float* x, const float* dy, const float* dymul,
                                  float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dp || !x || !dy || !dw || !workspace) return MGX_ERR_ARG;
    const mgx_conv_desc d = *dp;
    if (d.nimg <= 0 || d.cin <= 0 || d.hout <= 0 || d.wout <= 0 || d.kh <= 0 || d.kw <= 0) return MGX_ERR_ARG;
    if (d.ngemm != d.dkh * d.dkw * d.cout) return MGX_ERR_ARG;
    if (dbias && d.nbias <= 0) return MGX_ERR_ARG;
    if (pointwise2_wgrad_applicable(d)) {
        const Pw2Plan pp = pointwise2_wgrad_plan(d);
        if ((int64_t)pp.blocks * 3 * d.ngemm * (int64_t)sizeof(float) > workspace_bytes) return MGX_ERR_WORKSPACE;
        hipStream_t s2 = (hipStream_t)stream;
        float* part = (float*)workspace;
        const int lds = 3 * pp.pstep * d.ngemm * (int)sizeof(float);
        if (dymul)
            hipLaunchKernelGGL((pointwise2_wgrad_kernel<true>), dim3(pp.blocks), dim3(256), lds, s2, x, dy, dymul, part,
                               pp.npos, d.ngemm, (int)pp.ppb);
        else
            hipLaunchKernelGGL((pointwise2_wgrad_kernel<false>), dim3(pp.blocks), dim3(256), lds, s2, x, dy, dymul,
                               part, pp.npos, d.ngemm, (int)pp.ppb);
        hipLaunchKernelGGL(pointwise2_wgrad_reduce_kernel, dim3((3 * d.ngemm + 15) / 16), dim3(1024), 0, s2, part, dw,
                           dbias, pp.blocks, d.ngemm);
        return mgx_launch_status();
    }
    const WgradPlan p = make_plan(d);
    if ((p.partial_floats + p.bias_floats + p.du_floats) * (int64_t)sizeof(float) > workspace_bytes)
        return MGX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* bpartial = dbias ? partial + p.partial_floats : nullptr;
    if (p.wino) {
        // slabs -> dU[16][cin][32] (the generic fixed-order slab reduction with the 16 transform
        // positions in the place of taps) -> dw += G^T dU G
        float* du = partial + p.partial_floats + p.bias_floats;
        if (mgx_wwgrad_launch(d, x, dy, dymul, partial, bpartial, st) != MGX_OK) return MGX_ERR_LAUNCH;
        if (hipMemsetAsync(du, 0, p.du_floats * sizeof(float), st) != hipSuccess) return MGX_ERR_LAUNCH;
        mgx_conv_desc du_desc = d;
        du_desc.kh = du_desc.kw = 4;
        du_desc.wtap = (int64_t)d.cin * 32;
        du_desc.wsk = 32;
        du_desc.wsn = 1;
        {
            const int wblocks = (int)((p.du_floats + 63) / 64), bblocks = dbias ? (d.nbias + 3) / 4 : 0;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(wblocks + bblocks)), dim3(1024), 0, st, du_desc,
                               partial, du, p.splits, p.ktot, p.ntot, p.tiles_total, bpartial, dbias, wblocks,
                               (int)d.ngemm, (int)d.nbias, p.ntot);
        }
        if (mgx_wwgrad_finish(d, du, dw, st) != MGX_OK) return MGX_ERR_LAUNCH;
        return mgx_launch_status();
    }
#define MGX_WG(TT, KK, NN) launch_wgrad<TT, KK, NN>(d, p, x, dy, dymul, partial, bpartial, st)
    if (p.tiled) {
        if (mgx_twgrad_launch(d, x, dy, dymul, partial, bpartial, st) != MGX_OK) return MGX_ERR_LAUNCH;
    } else if (dense_layer(d)) {
        dim3 grid((unsigned)(p.splits / 4), (unsigned)(p.k_groups * p.n_groups)), block(256);
#define MGX_DW(KK, NN)                                                                                              \
    hipLaunchKernelGGL((dense_wgrad_kernel<KK, NN>), grid, block, 0, st, d, x, dy, dymul, partial, bpartial, p.k_groups, \
                       p.ktot, p.ntot, p.rows_per_split, p.tiles_total)
        if (p.KT == 1 && p.NT == 1) MGX_DW(1, 1);
        else if (p.KT == 1 && p.NT == 2) MGX_DW(1, 2);
        else if (p.KT == 1 && p.NT == 4) MGX_DW(1, 4);
        else if (p.KT == 2 && p.NT == 1) MGX_DW(2, 1);
        else if (p.KT == 2 && p.NT == 2) MGX_DW(2, 2);
        else MGX_DW(2, 4);
#undef MGX_DW
    } else if (p.T == 9) MGX_WG(9, 1, 1);
    else if (p.T == 8) MGX_WG(8, 1, 1);
    else if (p.T == 4) MGX_WG(4, 1, 1);
    else if (p.KT == 1 && p.NT == 1) MGX_WG(1, 1, 1);
    else if (p.KT == 1 && p.NT == 2) MGX_WG(1, 1, 2);
    else if (p.KT == 1 && p.NT == 4) MGX_WG(1, 1, 4);
    else if (p.KT == 2 && p.NT == 1) MGX_WG(1, 2, 1);
    else if (p.KT == 2 && p.NT == 2) MGX_WG(1, 2, 2);
    else MGX_WG(1, 2, 4);
#undef MGX_WG
    if (hipGetLastError() != hipSuccess) return MGX_ERR_LAUNCH;
    const long total = (long)d.kh * d.kw * d.cin * d.ngemm;
    const int wblocks = (int)((total + 63) / 64), bblocks = dbias ? (d.nbias + 3) / 4 : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(wblocks + bblocks)), dim3(1024), 0, st, d, partial, dw,
                       p.splits, p.ktot, p.ntot, p.tiles_total, bpartial, dbias, wblocks, (int)d.ngemm, (int)d.nbias,
                       p.ntot);
    return mgx_launch_status();
}
