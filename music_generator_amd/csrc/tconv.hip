// LDS-staged tile kernel for the WEIGHT gradient of the discriminator's stride-1 'same'
// convolutions (3x3 and 2x2, 32-channel chunks) on v_mfma_f32_32x32x2_f32, plus the dispatch
// predicates shared with the forward / data-gradient tile kernel (tconv3.hip).
//
// A workgroup (4 waves, two workgroups per CU) walks output tiles of TH=8 rows x TW=32 columns of
// one image.  The 32-channel input tile with its halo ((8+kh-1) x (32+kw-1) positions) and the
// 32-channel output-gradient tile (multiplied by the saved activation derivative while staging)
// go to LDS position-major; out-of-image positions are written as zeros.  Contraction runs over
// positions, two per MFMA: a wave owns two tile rows and keeps one 32x32 accumulator per tap for
// the whole launch; the grid is persistent and every workgroup writes one slab of partial sums (its
// four waves' tiles added in LDS in wave order) that conv_wgrad.hip adds in fixed order.  Input chunks of 128-channel convolutions are blockIdx.y.
#include "common.h"
#include "wino.h"
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

__device__ __forceinline__ Tile decode_tile(long t, int tiles_h, int tiles_w, int rows = TH) {
    Tile q;
    const int per_img = tiles_h * tiles_w;
    q.img = t / per_img;
    const int rem = (int)(t - q.img * per_img);
    const int th = rem / tiles_w;
    q.h0 = th * rows;
    q.w0 = (rem - th * tiles_w) * TW;
    return q;
}

// Stage channels [c0, c0+32) of the (IH x IW) window of a tile into LDS as [position][32], times
// `xmul` when MUL.  Thread (column tid/8, channel quad tid%8) moves its quad of every window row:
// one per-thread offset (out-of-image columns point outside the buffer range and read as zeros),
// the row is a scalar offset, rows outside the image use an empty buffer.  The columns beyond 32
// are one more load of the first (IW-32)*IH*8 threads.  No per-load address arithmetic: vector-ALU
// instructions take issue time from the MFMAs of the co-resident workgroup.
// `gather` (1x1 windows only): the staged image is the (a, b) phase of a 32-channel image twice as
// large, a = gather >> 1 & 1, b = gather & 1 with bit 2 set: position (i, j) reads (2i + a, 2j + b).
template <int IH, int IW, bool MUL>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ x, const float* __restrict__ xmul,
                                           long img, int hbase, int wbase, int H, int W, int ctot, int c0,
                                           int gather = 0) {
    constexpr int ROWB = 4;                               // rows in flight (bounds the staging registers)
    constexpr int XC = IW - 32;                           // columns beyond the 32 of the main pass
    constexpr int NRES = XC * IH * 8;
    const int tid = threadIdx.x, c4 = tid & 7, col = tid >> 3;
    const int cstr = gather ? 64 : ctot;                  // floats between columns / between rows of the window
    const int rstr = gather ? 4 * W * 32 : W * ctot;
    const long origin = gather ? ((img * 2 * H + 2 * hbase + ((gather >> 1) & 1)) * (2L * W) + 2 * wbase + (gather & 1)) * 32
                               : ((img * H + hbase) * (long)W + wbase) * ctot + c0;
    // the scalar row offset takes part in the range check: the range spans the whole window
    const unsigned range = (unsigned)((IH * rstr + IW * cstr) * 4);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + origin), 0, range, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(x + origin), 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mx =
        __builtin_amdgcn_make_buffer_rsrc((void*)((MUL ? xmul : x) + origin), 0, range, 0x00020000);
    const __amdgpu_buffer_rsrc_t mz =
        __builtin_amdgcn_make_buffer_rsrc((void*)((MUL ? xmul : x) + origin), 0, 0, 0x00020000);
    const unsigned vmain = (unsigned)(col + wbase) < (unsigned)W ? (unsigned)((col * cstr + c4 * 4) * 4) : 0x80000000u;
    const int row_bytes = rstr * 4;
    float* dst = lds + col * 32 + c4 * 4;
#pragma unroll
    for (int i0 = 0; i0 < IH; i0 += ROWB) {
        f32x4 v[ROWB], m[ROWB];
#pragma unroll
        for (int j = 0; j < ROWB; ++j) {
            if (i0 + j < IH) {
                const bool row_ok = (unsigned)(hbase + i0 + j) < (unsigned)H;          // uniform
                v[j] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(row_ok ? rx : rz, vmain, (i0 + j) * row_bytes, 0));
                if (MUL)
                    m[j] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(row_ok ? mx : mz, vmain, (i0 + j) * row_bytes, 0));
            }
        }
#pragma unroll
        for (int j = 0; j < ROWB; ++j)
            if (i0 + j < IH) *(f32x4*)(dst + (i0 + j) * IW * 32) = MUL ? v[j] * m[j] : v[j];
    }
    if (NRES > 0) {
        const int rrow = tid / (XC * 8), rem = tid - rrow * (XC * 8);
        const int rcol = 32 + (rem >> 3), rq = rem & 7;
        const bool ok = tid < NRES && (unsigned)(hbase + rrow) < (unsigned)H && (unsigned)(rcol + wbase) < (unsigned)W;
        const unsigned vres = ok ? (unsigned)((rrow * rstr + rcol * cstr + rq * 4) * 4) : 0x80000000u;
        f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vres, 0, 0));
        if (MUL) v *= __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mx, vres, 0, 0));
        if (tid < NRES) *(f32x4*)(lds + (rrow * IW + rcol) * 32 + rq * 4) = v;
    }
}

// RT = rows of a tile: 8, or 4 for images whose height is a multiple of 4 only (one row per wave)
template <int KH, int KW, int RT = TH>
constexpr int twgrad_lds_bytes() { return ((RT + KH - 1) * (TW + KW - 1) + RT * TW) * 32 * 4; }

// A 4x4 kernel keeps 16 accumulator tiles = all 256 accumulation registers of a wave: one wave per
// SIMD, MFMAs through the asm form that pins the accumulators there.
// NCH > 1: the output gradient has NCH 32-channel chunks (the 1x1 convolution 32 -> 128); they are
// staged one after the other against the same staged input tile, one accumulator tile each.
template <int KH, int KW, int RT, int NCH = 1>
__global__ __launch_bounds__(256, KH * KW == 16 ? 1 : 2) void twgrad_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        const float* __restrict__ dymul, float* __restrict__ partial,
                                                        float* __restrict__ bpartial, long ntiles, int tiles_h,
                                                        int tiles_w, int gather) {
    // gather (1x1 kernels): the weight gradient of a 2x2 stride-2 convolution 32 -> 32; input chunk
    // kc is the (kc >> 1, kc & 1) phase of the 32-channel input, twice the size of the output
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int IH = RT + KH - 1, IW = TW + KW - 1, TAPS = KH * KW;
    float* xl = lds;                    // [IH*IW][32]
    float* gl = lds + IH * IW * 32;     // [RT*TW][32]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hout, W = d.wout;       // tiles are laid over the output image (== input unless gathered)
    // the input chunks of a tile share its output-gradient tile: keep them on one XCD (workgroups go
    // to the XCDs round-robin by linear id); slabs are addressed by (chunk, split), sums unchanged
    unsigned bx = blockIdx.x;
    int kc = blockIdx.y;
    const int kchunks = gridDim.y;
    if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, q = lin >> 3;
        kc = (int)(q % gridDim.y);
        bx = (q / gridDim.y) * 8 + (lin & 7);
    }

    f32x16 acc[TAPS * NCH];
#pragma unroll
    for (int t = 0; t < TAPS * NCH; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum[NCH];
#pragma unroll
    for (int nc = 0; nc < NCH; ++nc) bsum[nc] = 0.f;

    for (long t = bx; t < ntiles; t += gridDim.x) {
        const Tile q = decode_tile(t, tiles_h, tiles_w, RT);
        __syncthreads();
        if (KH == 1 && gather) stage_tile<IH, IW, false>(xl, x, nullptr, q.img, q.h0, q.w0, H, W, 32, 0, 4 | kc);
        else stage_tile<IH, IW, false>(xl, x, nullptr, q.img, q.h0 - d.pt, q.w0 - d.pl, H, W, d.cin, kc * 32);
#pragma unroll
        for (int nc = 0; nc < NCH; ++nc) {
            if (nc > 0) __syncthreads();             // every wave is done with the previous chunk
            if (dymul) stage_tile<RT, TW, true>(gl, dy, dymul, q.img, q.h0, q.w0, H, W, d.ngemm, nc * 32);
            else stage_tile<RT, TW, false>(gl, dy, dymul, q.img, q.h0, q.w0, H, W, d.ngemm, nc * 32);
            __syncthreads();
            if constexpr (RT == 4 && KH > 1) {
                // 4-row tiles: wave wv takes 8 columns of all four rows, and a tap row that lies in the padding
                // above or below the image (zeros in the staged tile) is skipped - every wave skips the same ones
                const int hb = q.h0 - d.pt;
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    float bv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bv[j] = gl[(row * TW + wv * 8 + j * 2 + h) * 32 + r];
                        bsum[nc] += bv[j];
                    }
#pragma unroll
                    for (int a = 0; a < KH; ++a) {
                        if ((unsigned)(hb + row + a) < (unsigned)H) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int b = 0; b < KW; ++b) {
                                    const float av = xl[((row + a) * IW + (wv * 8 + j * 2 + h + b)) * 32 + r];
                                    if constexpr (TAPS == 16) mfma_acc(acc[(a * KW + b) * NCH + nc], av, bv[j]);
                                    else acc[(a * KW + b) * NCH + nc] = mfma32(av, bv[j], acc[(a * KW + b) * NCH + nc]);
                                }
                        }
                    }
                }
            } else
#pragma unroll 4
            for (int pp = 0; pp < RT * 4; ++pp) {
                const int px = pp * 2 + h;               // within this wave's RT/4 rows
                const int row = wv * (RT / 4) + (px >> 5), col = px & 31;
                const float bv = gl[(row * TW + col) * 32 + r];
                bsum[nc] += bv;
#pragma unroll
                for (int a = 0; a < KH; ++a)
#pragma unroll
                    for (int b = 0; b < KW; ++b) {
                        const float av = xl[((row + a) * IW + (col + b)) * 32 + r];
                        if constexpr (TAPS == 16) mfma_acc(acc[(a * KW + b) * NCH + nc], av, bv);
                        else acc[(a * KW + b) * NCH + nc] = mfma32(av, bv, acc[(a * KW + b) * NCH + nc]);
                    }
            }
        }
    }
    // One slab per WORKGROUP: partial[((split*tiles_total + tile)*16 + reg)*64 + lane], split = workgroup,
    // tile = (tap*kchunks + kc)*NCH + nc (32 x 32 each).  The four waves' accumulator tiles are added in LDS in
    // wave order, two tiles a round (waves 0 and 1 add one each): a slab per wave is four times the bytes the
    // fixed-order reduction kernel has to read, and for small images that kernel was as long as this one.
    const long split = bx;
    const long tiles_total = (long)TAPS * kchunks * NCH;
    constexpr int NTILE = TAPS * NCH;
    static_for<0, (NTILE + 1) / 2>([&](auto rr) {
        constexpr int t0 = decltype(rr)::value * 2;
        __syncthreads();                      // the tile buffers (or the previous round's tiles) are done with
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (t0 + j < NTILE) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) lds[((j * 4 + wv) * 16 + reg) * 64 + lane] = acc[t0 + j][reg];
            }
        __syncthreads();
        if (wv < 2 && t0 + wv < NTILE) {
            const int tt = t0 + wv, t = tt / NCH, nc = tt - t * NCH;
            const float* src = lds + (wv * 4) * 1024 + lane;
            float* dst = partial + ((split * tiles_total + ((long)t * kchunks + kc) * NCH + nc) * 16) * 64 + lane;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                dst[reg * 64] = ((src[reg * 64] + src[1024 + reg * 64]) + src[2048 + reg * 64]) + src[3072 + reg * 64];
        }
    });
    if (bpartial && kc == 0) {
        __syncthreads();
#pragma unroll
        for (int nc = 0; nc < NCH; ++nc) lds[(wv * NCH + nc) * 64 + lane] = bsum[nc];
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int nc = 0; nc < NCH; ++nc)
                bpartial[(split * NCH + nc) * 64 + lane] =
                    ((lds[nc * 64 + lane] + lds[(NCH + nc) * 64 + lane]) + lds[(2 * NCH + nc) * 64 + lane]) +
                    lds[(3 * NCH + nc) * 64 + lane];
        }
    }
}

int tile_grid(long ntiles) {
    long g = ntiles < 512 ? ntiles : 512;
    return (int)g;
}

}  // namespace

// Is this contraction a stride-1 'same' conv the tile kernels cover?
// the two strided forms the 1x1 tile kernel covers: 2x2 stride-2 'valid' convolution 32 -> 32 (input
// gathered as a 128-channel position) and its data-gradient (1x1, 32 -> 4 x 32 with depth-to-space)
static bool strided2_applicable(const mgx_conv_desc& d) {
    if (d.nbias != 0 && d.nbias != d.ngemm && !(d.dkh == 2 && d.nbias == d.cout)) return false;
    if ((long)2 * (TH + 2) * 2 * d.win * 32 * 4 >= (1L << 31)) return false;
    if ((long)d.nimg * (d.hout / TH) * ((d.wout + TW - 1) / TW) >= (1L << 31)) return false;
    if (d.kh == 2 && d.kw == 2 && d.sh == 2 && d.sw == 2 && d.dkh == 1 && d.dkw == 1 && d.pt == 0 && d.pl == 0 &&
        d.flip == 0 && d.cin == 32 && d.ngemm == 32 && d.cout == 32 && d.hin == 2 * d.hout && d.win == 2 * d.wout &&
        d.hout % TH == 0 && d.wsn == 1 && d.wsk == 32 && d.wtap == 1024)
        return true;
    if (d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.dkh == 2 && d.dkw == 2 && d.pt == 0 && d.pl == 0 &&
        d.cin == 32 && d.cout == 32 && d.ngemm == 128 && d.hout == d.hin && d.wout == d.win && d.hin % TH == 0 &&
        d.nbias == 0 && ((d.wsk == 1 && d.wsn == 32) || (d.wsn == 1 && d.wsk == 128)))
        return true;
    return false;
}

bool mgx_tconv_applicable(const mgx_conv_desc& d) {
    if (strided2_applicable(d)) return true;
    if (d.kh != d.kw || d.kh < 1 || d.kh > 4) return false;
    if (d.kh == 4 && (d.cin != 32 || d.ngemm != 32)) return false;   // 16 weight slices: one 32x32 slice per tap
    if (d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1) return false;
    if (d.hout != d.hin || d.wout != d.win) return false;
    if (d.pt < 0 || d.pt >= d.kh || d.pl < 0 || d.pl >= d.kw) return false;
    if (d.cin % 32 || d.ngemm % 32 || d.cout != d.ngemm) return false;
    if (d.cin != 32 && d.ngemm != 32) return false;
    if (d.cin > 128 || d.ngemm > 128) return false;
    if (d.hin % 4) return false;          // 8-row tiles, or 4-row tiles when the height is a multiple of 4 only
    // the tile kernels address one window of rows / one output row through 32-bit buffer offsets
    if ((long)(TH + 3) * d.win * (d.cin > d.ngemm ? d.cin : d.ngemm) * 4 >= (1L << 31)) return false;
    if ((long)d.nimg * (d.hin / 4) * ((d.win + TW - 1) / TW) >= (1L << 31)) return false;
    if (d.nbias != 0 && d.nbias != d.ngemm) return false;
    // dense weight slices per tap: [k][n] rows (forward) or [n][k] rows (data-gradient)
    const bool kn = d.wsn == 1 && d.wsk == d.ngemm;
    const bool nk = d.wsk == 1 && d.wsn == d.cin;
    if (!kn && !nk) return false;
    if (d.wtap != (int64_t)d.cin * d.ngemm) return false;
    return true;
}

int mgx_tconv3_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                      float* y, float* dact, hipStream_t st, const float* blend_old, const float* blend_alpha);

int mgx_tconv_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                     float* y, float* dact, hipStream_t st) {
    return mgx_tconv3_launch(d, x, xmul, w, bias, y, dact, st, nullptr, nullptr);   // tconv3.hip
}

// The 1x1 convolution + SoftRectifier that ends a discriminator block (trainer/models.py:280-281) and the
// WeightedSum behind it (models.py:297) in one launch, for passes that keep no tape.
extern "C" int mgx_conv1x1_blend_supported(const mgx_conv_desc* d) {
    return d && d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->dkh == 1 && d->dkw == 1 && !d->accumulate &&
           d->cin == 32 && d->ngemm == 128 && mgx_tconv_applicable(*d);
}

extern "C" int mgx_conv1x1_blend_f32(const mgx_conv_desc* d, const float* x, const float* w, const float* bias,
                                     const float* old, const float* alpha, float* y, void* stream) {
    if (!d || !x || !w || !old || !alpha || !y || !mgx_conv1x1_blend_supported(d)) return MGX_ERR_ARG;
    if (d->nbias > 0 && (!bias || d->nbias != d->ngemm)) return MGX_ERR_ARG;
    return mgx_tconv3_launch(*d, x, nullptr, w, bias, y, nullptr, (hipStream_t)stream, old, alpha);
}

// weight gradient: same applicability on the FORWARD descriptor; 32 output channels, or the 1x1
// convolution 32 -> 128
static bool twgrad_strided2_applicable(const mgx_conv_desc& d);

bool mgx_twgrad_applicable(const mgx_conv_desc& d) {
    if (twgrad_strided2_applicable(d)) return true;
    if (!mgx_tconv_applicable(d) || d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1 || d.flip != 0) return false;
    if (d.kh == 1) return d.cin == 32 && d.ngemm == 128;        // 1x1, four output chunks
    return d.ngemm == 32;
}

// the 2x2 stride-2 'valid' convolution 32 -> 32: as a 1x1 weight gradient over the four phases
static bool twgrad_strided2_applicable(const mgx_conv_desc& d) {
    return strided2_applicable(d) && d.sh == 2 && d.hout % 4 == 0 &&
           (long)4 * (TH + 1) * d.win * 32 * 4 < (1L << 31);
}

bool mgx_pointwise2_applicable(const mgx_conv_desc& d);   // conv_gemm.hip
bool mgx_pointwise_to2_applicable(const mgx_conv_desc& d);
bool mgx_wconv_enabled();                                  // conv_gemm.hip
bool mgx_wconv_applicable(const mgx_conv_desc& d, bool mul);   // wconv.hip
bool mgx_wwgrad_applicable(const mgx_conv_desc& d);            // wwgrad.hip
bool mgx_pgemm_applicable(const mgx_conv_desc& d, bool mul);   // pgemm.hip
bool mgx_tapsum_applicable(const mgx_conv_desc& d, bool mul);

extern "C" int mgx_conv_route(const mgx_conv_desc* d, int wgrad) {
    if (!d) return 0;
    if (wgrad == 1) {
        if (mgx_wconv_enabled() && mgx_wwgrad_applicable(*d)) return 3;
        return mgx_twgrad_applicable(*d) ? 1 : (mgx_pointwise2_applicable(*d) ? 2 : 0);
    }
    if (mgx_wconv_enabled() && mgx_wconv_applicable(*d, wgrad == 2)) return 3;
    if (mgx_tconv_applicable(*d)) return 1;
    if (mgx_pointwise_to2_applicable(*d)) return 2;
    if (mgx_pointwise2_applicable(*d) && wgrad == 0) return 2;
    if (mgx_pgemm_applicable(*d, wgrad == 2)) return 4;
    return mgx_tapsum_applicable(*d, wgrad == 2) ? 5 : 0;
}

long mgx_twgrad_splits(const mgx_conv_desc& d) {
    const int rt = (d.hout % TH || d.kh == 4) ? 4 : TH;
    const int tiles_h = d.hout / rt, tiles_w = (d.wout + TW - 1) / TW;
    return (long)tile_grid((long)d.nimg * tiles_h * tiles_w);      // one slab per workgroup
}

int mgx_twgrad_launch(const mgx_conv_desc& d, const float* x, const float* dy, const float* dymul, float* partial,
                      float* bpartial, hipStream_t st) {
    const int rt = (d.hout % TH || d.kh == 4) ? 4 : TH;
    const int tiles_h = d.hout / rt, tiles_w = (d.wout + TW - 1) / TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    const bool s2 = d.sh == 2;
    dim3 grid(tile_grid(ntiles), s2 ? 4 : d.cin / 32), block(256);
#define MGX_TWGRAD(KK, RR)                                                                                         \
    hipLaunchKernelGGL((twgrad_kernel<KK, KK, RR>), grid, block, (twgrad_lds_bytes<KK, KK, RR>()), st, d, x, dy,   \
                       dymul, partial, bpartial, ntiles, tiles_h, tiles_w, 0)
    if (s2 && rt == 8)
        hipLaunchKernelGGL((twgrad_kernel<1, 1, 8, 1>), grid, block, (twgrad_lds_bytes<1, 1, 8>()), st, d, x, dy, dymul,
                           partial, bpartial, ntiles, tiles_h, tiles_w, 1);
    else if (s2)
        hipLaunchKernelGGL((twgrad_kernel<1, 1, 4, 1>), grid, block, (twgrad_lds_bytes<1, 1, 4>()), st, d, x, dy, dymul,
                           partial, bpartial, ntiles, tiles_h, tiles_w, 1);
    else if (d.kh == 1 && rt == 8)
        hipLaunchKernelGGL((twgrad_kernel<1, 1, 8, 4>), grid, block, (twgrad_lds_bytes<1, 1, 8>()), st, d, x, dy, dymul,
                           partial, bpartial, ntiles, tiles_h, tiles_w, 0);
    else if (d.kh == 1)
        hipLaunchKernelGGL((twgrad_kernel<1, 1, 4, 4>), grid, block, (twgrad_lds_bytes<1, 1, 4>()), st, d, x, dy, dymul,
                           partial, bpartial, ntiles, tiles_h, tiles_w, 0);
    else if (d.kh == 4) MGX_TWGRAD(4, 4);
    else if (d.kh == 3 && rt == 8) MGX_TWGRAD(3, 8);
    else if (d.kh == 3) MGX_TWGRAD(3, 4);
    else if (rt == 8) MGX_TWGRAD(2, 8);
    else MGX_TWGRAD(2, 4);
#undef MGX_TWGRAD
    return mgx_launch_status();
}
