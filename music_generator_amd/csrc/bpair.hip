// Two consecutive convolution + activation layers of a discriminator block (trainer/models.py:259-276) in one
// launch, for forward passes that keep no tape: Conv2D(32, 3x3, 'same') + SoftRectifier followed by
// Conv2D(32, 3x3 or 2x2, 'same') + SoftRectifier, 32 -> 32 -> 32 channels, bf16 NHWC in and out.  The output of
// the first layer never reaches HBM: it is produced tile by tile into LDS (as bf16, exactly the values the
// layer-by-layer path would have stored) and consumed from there.
//
// A workgroup of 16 waves owns a 16 x 32 region of the FIRST layer's output (wave w = row w): that is what the
// second layer needs for (17 - KB) x (33 - KB) of its own outputs, and it needs an 18 x 34 input tile with halo.
// Per tile: MFMAs of layer 1 on the staged input tile -> bias (the accumulator's start value), activation,
// zeros where the position lies outside the image (the second layer's zero padding) -> LDS; barrier; MFMAs of
// layer 2 on the LDS tile -> activation -> stores of the valid outputs; barrier.  The next input tile is loaded
// during the first MFMA phase (two input buffers).  LDS layouts, operand fragments and the lane exchange before
// the stores are those of bconv.hip.  Results are bit-identical to two launches of mgx_conv_gemm_bf16.
#include "bf16.h"
#include "../../include/mgx.h"

namespace {

constexpr int PA_H = 16, PA_W = 32;                  // region of the first layer's output per tile
constexpr int PI_H = PA_H + 2, PI_W = PA_W + 2;      // input tile (3x3, pad 1)
constexpr int P_NPIX = PI_H * PI_W;
constexpr int P_NTHR = 64 * PA_H;

__device__ __forceinline__ int pxoff(int p, int q) { return p * 64 + ((q ^ ((p >> 2) & 3)) << 4); }

struct PairDesc {
    int nimg, h, w;
    float alpha_a, alpha_b;
};

template <int KB>
constexpr int pair_lds_bytes() {
    return (9 + KB * KB) * 32 * 32 * 2 + 2 * P_NPIX * 64 + PA_H * (PA_W + KB - 1) * 64 + 2 * 32 * 4;
}

// Keras kernel (kh, kw, 32, 32) float32 -> bf16 MFMA A fragments [tap][k-step][lane][8] in LDS
__device__ __forceinline__ void pair_weights(const float* __restrict__ w, int taps, unsigned char* wl, int tid) {
    for (int i = tid; i < taps * 1024; i += P_NTHR) {
        const int n = i & 31, k = (i >> 5) & 31, tap = i >> 10;
        const int f = tap * 2 + (k >> 4);
        const int ln = ((k >> 3) & 1) * 32 + n;
        *(bf16_t*)(wl + ((f * 64 + ln) * 8 + (k & 7)) * 2) = (bf16_t)w[i];
    }
}

template <int KB>
__global__ __launch_bounds__(P_NTHR) void bpair_kernel(PairDesc d, const bf16_t* __restrict__ x,
                                                       const float* __restrict__ wa, const float* __restrict__ ba,
                                                       const float* __restrict__ wb, const float* __restrict__ bb,
                                                       bf16_t* __restrict__ y, long ntiles, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TBH = PA_H - KB + 1, TBW = PA_W - KB + 1;      // valid outputs of the second layer per tile
    constexpr int PB = (KB - 1) / 2;                             // leading pad of the second layer ('same')
    constexpr int MW = PA_W + KB - 1;                            // columns of the LDS tile of layer-1 outputs
    constexpr int SITER = (P_NPIX * 4 + P_NTHR - 1) / P_NTHR;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.h, W = d.w;
    unsigned char* wla = lds;
    unsigned char* wlb = wla + 9 * 2048;
    unsigned char* xt = wlb + KB * KB * 2048;                    // [2][P_NPIX][64 B]
    unsigned char* mid = xt + 2 * P_NPIX * 64;                   // [PA_H][MW][64 B]
    float* bl = (float*)(mid + PA_H * MW * 64);                  // biases of both layers

    pair_weights(wa, 9, wla, tid);
    pair_weights(wb, KB * KB, wlb, tid);
    if (tid < 64) bl[tid] = tid < 32 ? ba[tid] : bb[tid - 32];

    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        const unsigned tu = (unsigned)t;
        const unsigned im = tu / per_img;
        const unsigned rem = tu - im * per_img;
        const unsigned th = rem / (unsigned)tiles_w;
        img = im;
        h0 = (int)th * TBH;
        w0 = (int)(rem - th * tiles_w) * TBW;
    };
    // ---- input tile staging: thread t moves 16-byte piece #(t + 1024 i) = (position, quad) ----
    const int q4 = tid & 3;
    const unsigned band_bytes = (unsigned)PI_H * W * 64u;
    int rel[SITER], tyx[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int px = (tid >> 2) + (P_NTHR / 4) * i;
        const int ty = px / PI_W, tx = px - ty * PI_W;
        rel[i] = ((ty * W + tx) * 32 + q4 * 8) * 2;
        tyx[i] = px < P_NPIX ? ((ty << 16) | tx) : (0x7000 << 16);
    }
    u32x4 sv[SITER];
    auto stage_load = [&](long t) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        const int hb = h0 - PB - 1, wb0 = w0 - PB - 1;
        const long band = ((img * H + hb) * (long)W) * 32;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + band), 0, band_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const unsigned hi = (unsigned)((tyx[i] >> 16) + hb), wi = (unsigned)((tyx[i] & 0xffff) + wb0);
            const unsigned off = (hi < (unsigned)H && wi < (unsigned)W) ? (unsigned)(wb0 * 64 + rel[i]) : 0xfffffff0u;
            sv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto stage_write = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int px = (tid >> 2) + (P_NTHR / 4) * i;
            if ((P_NTHR / 4) * (i + 1) <= P_NPIX || px < P_NPIX) *(u32x4*)(buf + pxoff(px, q4)) = sv[i];
        }
    };

    long t = blockIdx.x;
    if (t < ntiles) {
        stage_load(t);
        stage_write(xt);
    }
    __syncthreads();
    const float ia_a = 1.0f / d.alpha_a, k1_a = 1.0f + ia_a;
    const float ia_b = 1.0f / d.alpha_b, k1_b = 1.0f + ia_b;
    int cur = 0;
    // registers 4g .. 4g+3 of an accumulator tile are output channels 8g + 4h .. 8g + 4h + 3
    auto acc_init = [&](f32x16& acc, const float* b32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(b32 + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
        }
    };
    auto activate = [&](f32x16& a, float ia, float k1, u32x4& s0, u32x4& s1) {
        unsigned o[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            float y0, y1, d0, d1;
            act_bf16(a[i], ia, k1, y0, d0);
            act_bf16(a[i + 1], ia, k1, y1, d1);
            o[i >> 1] = bf_pack(y0, y1);
        }
        s0 = (u32x4){o[0], o[1], o[2], o[3]};
        s1 = (u32x4){o[4], o[5], o[6], o[7]};
    };

    while (t < ntiles) {
        const long tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        if (has_next) stage_load(tn);
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        // ---- layer 1: row wvs of the 16 x 32 region ----
        {
            const unsigned char* xb = xt + cur * (P_NPIX * 64);
            const unsigned char* wfr = wla + lane * 16;
            f32x16 acc;
            acc_init(acc, bl);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ta = tap / 3, tb = tap - ta * 3;
                const int p0 = (wvs + ta) * PI_W + tb + r;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = mfma_bf16(*(const u32x4*)(wfr + (tap * 2 + ks) * 1024), *(const u32x4*)(xb + pxoff(p0, 2 * ks + h)),
                                    acc);
            }
            if (has_next) stage_write(xt + (cur ^ 1) * (P_NPIX * 64));
            mfma_result_wait(acc);
            acc_rows_interleaved(acc);
            u32x4 s0, s1;
            activate(acc, ia_a, k1_a, s0, s1);
            // the second layer pads with zeros: positions of this region that lie outside the image hold 0
            const int row = h0 - PB + wvs, col = w0 - PB + r;
            if ((unsigned)row >= (unsigned)H || (unsigned)col >= (unsigned)W) {
                s0 = (u32x4){0u, 0u, 0u, 0u};
                s1 = s0;
            }
            const int p = wvs * MW + r;
            *(u32x4*)(mid + pxoff(p, h)) = s0;          // channels 8h .. 8h+7     = quad h
            *(u32x4*)(mid + pxoff(p, 2 + h)) = s1;      // channels 16+8h .. +7    = quad 2 + h
        }
        __syncthreads();
        // ---- layer 2: row wvs of the (17 - KB) x (33 - KB) outputs ----
        if (wvs < TBH) {
            const unsigned char* wfr = wlb + lane * 16;
            f32x16 acc;
            acc_init(acc, bl + 32);
#pragma unroll
            for (int tap = 0; tap < KB * KB; ++tap) {
                const int ta = tap / KB, tb = tap - ta * KB;
                const int p0 = (wvs + ta) * MW + tb + r;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = mfma_bf16(*(const u32x4*)(wfr + (tap * 2 + ks) * 1024), *(const u32x4*)(mid + pxoff(p0, 2 * ks + h)),
                                    acc);
            }
            mfma_result_wait(acc);
            acc_rows_interleaved(acc);
            u32x4 s0, s1;
            activate(acc, ia_b, k1_b, s0, s1);
            const int row = h0 + wvs;
            const long rowoff = ((img * H + row) * (long)W) * 32;
            const unsigned row_bytes = row < H ? (unsigned)W * 64u : 0u;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + rowoff), 0, row_bytes, 0x00020000);
            const unsigned voff = r < TBW ? (unsigned)((w0 + r) * 64 + 16 * h) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(s0, ry, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(s1, ry, voff + 32, 0, 0);
        }
        __syncthreads();
        if (has_next) cur ^= 1;
        t = tn;
    }
}

template <int KB>
int launch_pair(const PairDesc& d, const bf16_t* x, const float* wa, const float* ba, const float* wb, const float* bb,
                bf16_t* y, hipStream_t st) {
    constexpr int TBH = PA_H - KB + 1, TBW = PA_W - KB + 1;
    const int tiles_h = (d.h + TBH - 1) / TBH, tiles_w = (d.w + TBW - 1) / TBW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    if (ntiles <= 0 || ntiles >= (1L << 31)) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    constexpr int lds = pair_lds_bytes<KB>();
    static_assert(lds <= 160 * 1024, "the pair kernel's LDS image exceeds a CU");
    if (hipFuncSetAttribute((const void*)bpair_kernel<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return MGX_ERR_LAUNCH;
    hipLaunchKernelGGL(bpair_kernel<KB>, dim3(grid), dim3(P_NTHR), lds, st, d, x, wa, ba, wb, bb, y, ntiles, tiles_h,
                       tiles_w);
    return mgx_launch_status();
}

}  // namespace

// Shapes the pair kernel serves: at least 16 rows (a tile is 16 rows of the first layer), byte offsets of a row
// band within 32 bits.
extern "C" int mgx_conv_pair_bf16_supported(int64_t nimg, int h, int w, int kb) {
    if (nimg <= 0 || h < 16 || w < 8 || (kb != 2 && kb != 3)) return 0;
    if ((long)PI_H * w * 64 >= (1L << 31)) return 0;
    return 1;
}

extern "C" int mgx_conv_pair_bf16(int64_t nimg, int h, int w, const void* x, const float* wa, const float* ba,
                                  float alpha_a, const float* wb, const float* bb, float alpha_b, int kb, void* y,
                                  void* stream) {
    if (!x || !wa || !ba || !wb || !bb || !y || !mgx_conv_pair_bf16_supported(nimg, h, w, kb)) return MGX_ERR_ARG;
    if (!(alpha_a > 0.f) || !(alpha_b > 0.f)) return MGX_ERR_ARG;
    PairDesc d;
    d.nimg = (int)nimg; d.h = h; d.w = w; d.alpha_a = alpha_a; d.alpha_b = alpha_b;
    hipStream_t st = (hipStream_t)stream;
    if (kb == 3) return launch_pair<3>(d, (const bf16_t*)x, wa, ba, wb, bb, (bf16_t*)y, st);
    return launch_pair<2>(d, (const bf16_t*)x, wa, ba, wb, bb, (bf16_t*)y, st);
}
