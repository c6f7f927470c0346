// LDS-staged tile kernels for the discriminator's stride-1 'same' convolutions (3x3 and 2x2,
// 32-channel chunks) on v_mfma_f32_32x32x2_f32: forward, data-gradient and weight-gradient.
//
// A workgroup (4 waves) walks output tiles of TH=8 rows x TW=32 columns of one image.  The input
// tile with its halo ((8+kh-1) x (32+kw-1) positions x 32 channels) is staged once into LDS with a
// 36-float position stride (conflict-free ds_read_b128 along channels for 32 consecutive
// positions); out-of-image positions are written as zeros, and the optional element-wise
// multiplier (dY (.) f'(pre) for gradients) is applied while staging.  Forward / data-gradient:
// every wave owns 2 rows (2 tiles of 32 positions), keeps the 32x32 weight slice of all taps in
// registers (taps*16 VGPRs, lane half h holds k = 16h..16h+15 for column l&31) and issues
// taps*16 MFMAs per tile from four 16-byte LDS reads per tap.  Inputs / outputs with 128 channels
// are processed as four 32-channel chunks (accumulating over input chunks, or looping over
// output chunks on the same staged tile).  The grid is persistent (<= 2 workgroups per CU).
#include "common.h"
#include "../../include/mgx.h"

#include <type_traits>

namespace {

constexpr int TH = 8, TW = 32, PSTR = 36;

// compile-time unrolled loop: f(integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int KH, int KW>
constexpr int tconv_lds_bytes() { return (TH + KH - 1) * (TW + KW - 1) * PSTR * 4; }

struct Tile {
    long img;
    int h0, w0;
};

__device__ __forceinline__ Tile decode_tile(long t, int tiles_h, int tiles_w) {
    Tile q;
    const int per_img = tiles_h * tiles_w;
    q.img = t / per_img;
    const int rem = (int)(t - q.img * per_img);
    const int th = rem / tiles_w;
    q.h0 = th * TH;
    q.w0 = (rem - th * tiles_w) * TW;
    return q;
}

// Stage channels [c0, c0+32) of the (IH x IW) input window of a tile into LDS, stride `pstr`.
// Every load is unconditional (out-of-image positions read a clamped, valid address and are then
// replaced by zeros) so that the loads of one batch are issued back to back instead of being
// branched around and waited for one by one.
template <int IH, int IW, bool MUL>
__device__ __forceinline__ void stage_tile(float* lds, int pstr, const float* __restrict__ x,
                                           const float* __restrict__ xmul, long img, int hbase, int wbase, int H,
                                           int W, int ctot, int c0) {
    constexpr int NPIX = IH * IW;
    constexpr int ITER = (NPIX + 31) / 32;
    constexpr int BATCH = 4;   // float4 loads in flight per thread (bounds the staging registers)
    const int c4 = threadIdx.x & 7;
    const int p0 = threadIdx.x >> 3;
    const long imgbase = img * H;
#pragma unroll
    for (int i0 = 0; i0 < ITER; i0 += BATCH) {
        f32x4 v[BATCH], m[BATCH];
        bool ok[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (i0 + j < ITER) {
                const int px = p0 + 32 * (i0 + j);
                const int ty = px / IW, tx = px - ty * IW;
                const int hi = hbase + ty, wi = wbase + tx;
                ok[j] = px < NPIX && hi >= 0 && hi < H && wi >= 0 && wi < W;
                const int hc = min(max(hi, 0), H - 1), wc = min(max(wi, 0), W - 1);
                const long off = ((imgbase + hc) * W + wc) * (long)ctot + c0 + c4 * 4;
                v[j] = *(const f32x4*)(x + off);
                if (MUL) m[j] = *(const f32x4*)(xmul + off);
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (i0 + j < ITER) {
                const int px = p0 + 32 * (i0 + j);
                f32x4 val = v[j];
                if (MUL) val *= m[j];
                if (!ok[j]) val = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (px < NPIX) *(f32x4*)(lds + px * pstr + c4 * 4) = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient
// ------------------------------------------------------------------------------------------
template <int KH, int KW, bool WT>   // WT: weights stored [n][k] (wsk == 1, wsn == 32); else [k][n] (wsk == 32, wsn == 1)
__global__ __launch_bounds__(256, 2) void tconv_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                       const float* __restrict__ xmul, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ dact, long ntiles, int tiles_h,
                                                       int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int IH = TH + KH - 1, IW = TW + KW - 1, TAPS = KH * KW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win;
    const int kchunks = d.cin / 32, nchunks = d.ngemm / 32;
    const bool reload_w = (kchunks * nchunks) > 1;
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    float wreg[TAPS][16];
    bool w_loaded = false;

    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const Tile q = decode_tile(t, tiles_h, tiles_w);
        for (int nc = 0; nc < nchunks; ++nc) {
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
            for (int kc = 0; kc < kchunks; ++kc) {
                __syncthreads();  // every wave is done reading the previous contents
                if (nc == 0 || kchunks > 1) {
                    if (xmul) stage_tile<IH, IW, true>(lds, PSTR, x, xmul, q.img, q.h0 - d.pt, q.w0 - d.pl, H, W, d.cin, kc * 32);
                    else stage_tile<IH, IW, false>(lds, PSTR, x, xmul, q.img, q.h0 - d.pt, q.w0 - d.pl, H, W, d.cin, kc * 32);
                }
                if (reload_w || !w_loaded) {
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        const int wt = d.flip ? (TAPS - 1 - tap) : tap;
                        constexpr int WSK = WT ? 1 : 32, WSN = WT ? 32 : 1;   // compile-time strides: one
                        const float* wp = w + (long)wt * d.wtap + (kc * 32 + 16 * h) * WSK +   // base address,
                                          (nc * 32 + r) * WSN;                                  // immediate offsets
#pragma unroll
                        for (int s = 0; s < 16; ++s) wreg[tap][s] = wp[s * WSK];
                    }
                    w_loaded = true;
                }
                __syncthreads();
                // (tile row, tap) steps in a fixed, fully unrolled order.  The scheduling barrier after
                // each step keeps the compiler from hoisting the LDS reads of later steps to the top
                // (the register file holds the weights); the second wave of the SIMD covers the wait.
                const float* base = lds + ((wv * 2) * IW + r) * PSTR + 16 * h;
                static_for<0, 2 * TAPS>([&](auto step_c) {
                    constexpr int step = decltype(step_c)::value;
                    constexpr int pt = step / TAPS, tap = step % TAPS;
                    const float* src = base + ((pt + tap / KW) * IW + (tap % KW)) * PSTR;
                    f32x4 cur[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) cur[j] = *(const f32x4*)(src + 4 * j);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[pt] = mfma32(cur[j][s], wreg[tap][4 * j + s], acc[pt]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ---- epilogue for output channels [nc*32, nc*32+32) ----
            const int n = nc * 32 + r;
            const float bv = d.nbias > 0 ? bias[n] : 0.f;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int hh = q.h0 + wv * 2 + pt;
                const long rowoff = ((q.img * H + hh) * (long)W) * d.ngemm + n;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int col = q.w0 + acc_row(reg, lane);
                    if (col >= W) continue;
                    const long off = rowoff + (long)col * d.ngemm;
                    float v = acc[pt][reg] + bv;
                    if (d.act_alpha > 0.f) {
                        float tt, s2;
                        tanh_sech2(v, tt, s2);
                        if (dact) dact[off] = s2 + inv_alpha;
                        v = tt + v / d.act_alpha;
                    }
                    if (d.accumulate) v += y[off];
                    y[off] = v;
                    // bound the live ranges of the unrolled epilogue (the weights stay in registers)
                    if ((reg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[tap][ci][co] = sum_px X[px+tap][ci] * dP[px][co], ci chunk = blockIdx.y
// ------------------------------------------------------------------------------------------
template <int KH, int KW>
constexpr int twgrad_lds_bytes() { return ((TH + KH - 1) * (TW + KW - 1) + TH * TW) * 32 * 4; }

template <int KH, int KW>
__global__ __launch_bounds__(256, 2) void twgrad_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        const float* __restrict__ dymul, float* __restrict__ partial,
                                                        float* __restrict__ bpartial, long ntiles, int tiles_h,
                                                        int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int IH = TH + KH - 1, IW = TW + KW - 1, TAPS = KH * KW;
    float* xl = lds;                    // [IH*IW][32]
    float* gl = lds + IH * IW * 32;     // [TH*TW][32]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win;
    const int kc = blockIdx.y;
    const int kchunks = d.cin / 32;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;

    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const Tile q = decode_tile(t, tiles_h, tiles_w);
        __syncthreads();
        stage_tile<IH, IW, false>(xl, 32, x, nullptr, q.img, q.h0 - d.pt, q.w0 - d.pl, H, W, d.cin, kc * 32);
        if (dymul) stage_tile<TH, TW, true>(gl, 32, dy, dymul, q.img, q.h0, q.w0, H, W, d.ngemm, 0);
        else stage_tile<TH, TW, false>(gl, 32, dy, dymul, q.img, q.h0, q.w0, H, W, d.ngemm, 0);
        __syncthreads();
#pragma unroll 4
        for (int pp = 0; pp < 32; ++pp) {
            const int px = pp * 2 + h;               // 0..63 within this wave's two rows
            const int row = wv * 2 + (px >> 5), col = px & 31;
            const float bv = gl[(row * TW + col) * 32 + r];
            bsum += bv;
#pragma unroll
            for (int a = 0; a < KH; ++a)
#pragma unroll
                for (int b = 0; b < KW; ++b) {
                    const float av = xl[((row + a) * IW + (col + b)) * 32 + r];
                    acc[a * KW + b] = mfma32(av, bv, acc[a * KW + b]);
                }
        }
    }
    // slabs: partial[((split*tiles_total + tile)*16 + reg)*64 + lane], split = (wg, wave),
    // tile = tap*kchunks + kc (one 32-column tile)
    const long split = (long)blockIdx.x * 4 + wv;
    const long tiles_total = (long)TAPS * kchunks;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        float* dst = partial + ((split * tiles_total + (long)t * kchunks + kc) * 16) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) dst[reg * 64] = acc[t][reg];
    }
    if (bpartial && kc == 0) bpartial[split * 64 + lane] = bsum;
}

int tile_grid(long ntiles) {
    long g = ntiles < 512 ? ntiles : 512;
    return (int)g;
}

}  // namespace

// Is this contraction a stride-1 'same' conv the tile kernels cover?
bool mgx_tconv_applicable(const mgx_conv_desc& d) {
    if (d.kh != d.kw || (d.kh != 2 && d.kh != 3)) return false;
    if (d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1) return false;
    if (d.hout != d.hin || d.wout != d.win) return false;
    if (d.pt < 0 || d.pt >= d.kh || d.pl < 0 || d.pl >= d.kw) return false;
    if (d.cin % 32 || d.ngemm % 32 || d.cout != d.ngemm) return false;
    if (d.cin != 32 && d.ngemm != 32) return false;
    if (d.cin > 128 || d.ngemm > 128) return false;
    if (d.hin % TH) return false;
    if (d.nbias != 0 && d.nbias != d.ngemm) return false;
    // weight strides the kernels hard-code: [k][n] rows of 32 (forward) or [n][k] rows of 32 (data-gradient)
    const bool kn = d.wsk == 32 && d.wsn == 1 && d.ngemm == 32;
    const bool nk = d.wsk == 1 && d.wsn == 32 && d.cin == 32;
    if (!kn && !nk) return false;
    if (d.wtap != (int64_t)d.cin * d.ngemm) return false;
    return true;
}

int mgx_tconv_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                     float* y, float* dact, hipStream_t st) {
    const int tiles_h = d.hin / TH, tiles_w = (d.win + TW - 1) / TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    dim3 grid(tile_grid(ntiles)), block(256);
    constexpr int lds3 = tconv_lds_bytes<3, 3>(), lds2 = tconv_lds_bytes<2, 2>();
    const bool wt = d.wsk == 1;
#define MGX_TC(KK, WW, LDS) \
    hipLaunchKernelGGL((tconv_kernel<KK, KK, WW>), grid, block, LDS, st, d, x, xmul, w, bias, y, dact, ntiles, tiles_h, tiles_w)
    if (d.kh == 3) { if (wt) MGX_TC(3, true, lds3); else MGX_TC(3, false, lds3); }
    else { if (wt) MGX_TC(2, true, lds2); else MGX_TC(2, false, lds2); }
#undef MGX_TC
    return mgx_launch_status();
}

// weight gradient: same applicability on the FORWARD descriptor, output channels must be 32
bool mgx_twgrad_applicable(const mgx_conv_desc& d) { return mgx_tconv_applicable(d) && d.ngemm == 32 && d.flip == 0; }

long mgx_twgrad_splits(const mgx_conv_desc& d) {
    const int tiles_h = d.hin / TH, tiles_w = (d.win + TW - 1) / TW;
    return (long)tile_grid((long)d.nimg * tiles_h * tiles_w) * 4;
}

int mgx_twgrad_launch(const mgx_conv_desc& d, const float* x, const float* dy, const float* dymul, float* partial,
                      float* bpartial, hipStream_t st) {
    const int tiles_h = d.hin / TH, tiles_w = (d.win + TW - 1) / TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    dim3 grid(tile_grid(ntiles), d.cin / 32), block(256);
    constexpr int lds3 = twgrad_lds_bytes<3, 3>(), lds2 = twgrad_lds_bytes<2, 2>();
    if (d.kh == 3)
        hipLaunchKernelGGL((twgrad_kernel<3, 3>), grid, block, lds3, st, d, x, dy, dymul, partial, bpartial, ntiles,
                           tiles_h, tiles_w);
    else
        hipLaunchKernelGGL((twgrad_kernel<2, 2>), grid, block, lds2, st, d, x, dy, dymul, partial, bpartial, ntiles,
                           tiles_h, tiles_w);
    return mgx_launch_status();
}
