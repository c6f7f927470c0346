// The first two layers of a discriminator block (trainer/models.py:256-260) in one launch, for forward passes that
// keep no tape: Conv2D(128, 1x1) + SoftRectifier on the 2-channel audio ("from-audio"), then Conv2D(32, 3x3, 'same')
// + SoftRectifier.  The 128-channel tensor between them is the largest of the network (256 bytes a position in
// bf16 at the block's full resolution: written once and read once, 512 of the pair's 584 bytes a position); here it
// is never stored: a workgroup re-computes it from the audio for the tile it convolves.
//
// Tiles, LDS layouts, operand fragments, accumulation order, epilogue and tile order are those of bconv.hip with
// 16 x 32 tiles (16 waves, wave w = tile row w) and 128 input channels as 4 steps over 32-channel slices.  Where
// bconv.hip stages a slice of the stored tensor, this kernel computes it: thread t owns 16-byte pieces (position,
// 8 channels) of the 18 x 34 halo tile, reads the position's two audio samples once per tile (prefetched one tile
// ahead), and per step evaluates tanh(v) + v / alpha of v = x0 w0 + x1 w1 + b for its 8 channels with the
// arithmetic of the stand-alone from-audio kernel (same fused multiply-adds, same bf16 rounding), zeros where the
// position lies outside the image (the convolution pads the ACTIVATION with zeros).  The slice of step s+1 is
// computed into the second tile buffer before the MFMAs of step s are issued; one barrier per step.
// Results are bit-identical to mgx_pointwise2_bf16_fwd followed by mgx_conv_gemm_bf16.
#include "bf16.h"
#include "../../include/mgx.h"

namespace {

constexpr int AU_TH = 16, AU_TW = 32, AU_IH = 18, AU_IW = 34, AU_NPIX = AU_IH * AU_IW, AU_NTHR = 64 * AU_TH;
constexpr int AU_SITER = (AU_NPIX * 4 + AU_NTHR - 1) / AU_NTHR;      // 16-byte pieces per thread and slice: 3
constexpr int AU_WL = 9 * 128 * 32 * 2;                               // the 3x3 weights as bf16 fragments
constexpr int AU_LDS = AU_WL + 2 * AU_NPIX * 64 + 3 * 128 * 4 + 32 * 4;

__device__ __forceinline__ int axoff(int p, int q) { return p * 64 + ((q ^ ((p >> 2) & 3)) << 4); }

struct AudioDesc {
    int nimg, h, w;
    float alpha_a, alpha_b;
};

__global__ __launch_bounds__(AU_NTHR) void baudio_kernel(AudioDesc d, const float* __restrict__ x,
                                                         const float* __restrict__ wa, const float* __restrict__ ba,
                                                         const float* __restrict__ wb, const float* __restrict__ bb,
                                                         bf16_t* __restrict__ y, long ntiles, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.h, W = d.w;
    unsigned char* wl = lds;                                   // [tap][k/16][64 lanes][16 B]
    unsigned char* xt = lds + AU_WL;                           // [2][NPIX][64 B]
    float* al = (float*)(xt + 2 * AU_NPIX * 64);               // from-audio kernel rows w0, w1 and bias: [3][128]
    float* bl = al + 3 * 128;                                  // bias of the 3x3 convolution

    // ---- weights of the 3x3 convolution: Keras (3, 3, 128, 32) float32 -> bf16 fragments (once per workgroup) ----
    for (int i = tid; i < 9 * 128 * 32; i += AU_NTHR) {
        const int n = i & 31, k = (i >> 5) & 127, tap = i >> 12;
        const int f = tap * 8 + (k >> 4), ln = ((k >> 3) & 1) * 32 + n;
        *(bf16_t*)(wl + ((f * 64 + ln) * 8 + (k & 7)) * 2) = (bf16_t)wb[i];
    }
    if (tid < 128) {
        al[tid] = wa[tid];
        al[128 + tid] = wa[128 + tid];
        al[256 + tid] = ba[tid];
    }
    if (tid < 32) bl[tid] = bb[tid];

    // tile order: as bconv.hip (the 32 workgroups of an XCD take 32 consecutive tiles, vertical neighbours first)
    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        unsigned tu = (unsigned)t;
        if (gridDim.x == 256) {
            const unsigned round = tu >> 8, b = tu & 255u;
            if ((long)(round + 1) * 256 <= ntiles) tu = (round << 8) + (b & 7u) * 32u + (b >> 3);
        }
        const unsigned im = tu / per_img;
        const unsigned rem = tu - im * per_img;
        const unsigned tw = rem / (unsigned)tiles_h;
        img = im;
        h0 = (int)(rem - tw * tiles_h) * AU_TH;
        w0 = (int)tw * AU_TW;
    };

    // ---- the thread's pieces: piece #(t + 1024 i) = (position t / 4 + 256 i, quad t & 3) ----
    const int q4 = tid & 3;
    int pty[AU_SITER], ptx[AU_SITER];
#pragma unroll
    for (int i = 0; i < AU_SITER; ++i) {
        const int px = (tid >> 2) + (AU_NTHR / 4) * i;
        pty[i] = px / AU_IW;
        ptx[i] = px - pty[i] * AU_IW;
    }
    // audio samples of the thread's positions of tile t_ (0 and "outside" where the position is not in the image)
    auto audio_load = [&](long t_, float2 (&xv)[AU_SITER], unsigned& inside) {
        long img; int h0, w0;
        tile_of(t_, img, h0, w0);
        inside = 0u;
#pragma unroll
        for (int i = 0; i < AU_SITER; ++i) {
            const int px = (tid >> 2) + (AU_NTHR / 4) * i;
            const int hi = h0 - 1 + pty[i], wi = w0 - 1 + ptx[i];
            const bool ok = px < AU_NPIX && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            xv[i] = make_float2(0.f, 0.f);
            if (ok) {
                xv[i] = *(const float2*)(x + ((img * H + hi) * (long)W + wi) * 2);
                inside |= 1u << i;
            }
        }
    };
    const float ia_a = 1.0f / d.alpha_a, k1_a = 1.0f + ia_a;
    // slice kc (channels 32 kc .. 32 kc + 31) of the activation of the tile whose samples are xv -> tile buffer
    auto slice_compute = [&](const float2 (&xv)[AU_SITER], unsigned inside, int kc, unsigned char* buf) {
        const int c0 = 32 * kc + 8 * q4;
        float w0[8], w1[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            const f32x4 a0 = *(const f32x4*)(al + c0 + e), a1 = *(const f32x4*)(al + 128 + c0 + e),
                        a2 = *(const f32x4*)(al + 256 + c0 + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w0[e + j] = a0[j]; w1[e + j] = a1[j]; b[e + j] = a2[j]; }
        }
#pragma unroll
        for (int i = 0; i < AU_SITER; ++i) {
            const int px = (tid >> 2) + (AU_NTHR / 4) * i;
            if ((AU_NTHR / 4) * (i + 1) > AU_NPIX && px >= AU_NPIX) continue;
            float v[8], dv;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = __builtin_fmaf(xv[i].y, w1[k], __builtin_fmaf(xv[i].x, w0[k], b[k]));
                act_bf16(v[k], ia_a, k1_a, v[k], dv);
            }
            u32x4 o = {bf_pack(v[0], v[1]), bf_pack(v[2], v[3]), bf_pack(v[4], v[5]), bf_pack(v[6], v[7])};
            if (!((inside >> i) & 1u)) o = (u32x4){0u, 0u, 0u, 0u};
            *(u32x4*)(buf + axoff(px, q4)) = o;
        }
    };

    const float ia_b = 1.0f / d.alpha_b, k1_b = 1.0f + ia_b;
    long t = blockIdx.x;
    float2 xc[AU_SITER], xn[AU_SITER];
    unsigned in_c = 0u, in_n = 0u;
    __syncthreads();                                   // al, bl, wl are in place
    if (t < ntiles) {
        audio_load(t, xc, in_c);
        slice_compute(xc, in_c, 0, xt);
    }
    __syncthreads();
    int cur = 0;
    while (t < ntiles) {
        const long tn = t + gridDim.x;
        const bool has_next = tn < ntiles;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                  // registers 4g .. 4g+3: output channels 8g + 4h .. + 3
            const f32x4 v = *(const f32x4*)(bl + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
        }
#pragma unroll 1
        for (int kc = 0; kc < 4; ++kc) {
            // the next slice (of this tile, or slice 0 of the next one) goes into the other buffer first
            if (kc == 2 && has_next) audio_load(tn, xn, in_n);
            if (kc < 3) slice_compute(xc, in_c, kc + 1, xt + (cur ^ 1) * (AU_NPIX * 64));
            else if (has_next) slice_compute(xn, in_n, 0, xt + (cur ^ 1) * (AU_NPIX * 64));
            const unsigned char* xb = xt + cur * (AU_NPIX * 64);
            const unsigned char* wfr = wl + (kc * 2 * 64 + lane) * 16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ta = tap / 3, tb = tap - ta * 3;
                const int p0 = (wvs + ta) * AU_IW + tb + r;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc = mfma_bf16(*(const u32x4*)(wfr + (tap * 8 + ks) * 1024), *(const u32x4*)(xb + axoff(p0, 2 * ks + h)),
                                    acc);
            }
            if (kc == 3) {
                // ---- epilogue of the wave's row: activation, lane exchange, two 16-byte stores per lane ----
                long img; int h0, w0;
                tile_of(t, img, h0, w0);
                const int row = h0 + wvs;
                const long rowoff = ((img * H + row) * (long)W) * 32;
                const unsigned row_bytes = row < H ? (unsigned)W * 64u : 0u;
                const __amdgpu_buffer_rsrc_t ry =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(y + rowoff), 0, row_bytes, 0x00020000);
                mfma_result_wait(acc);
                acc_rows_interleaved(acc);
                unsigned o[8];
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    float y0, y1, d0, d1;
                    act_bf16(acc[i], ia_b, k1_b, y0, d0);
                    act_bf16(acc[i + 1], ia_b, k1_b, y1, d1);
                    o[i >> 1] = bf_pack(y0, y1);
                }
                const u32x4 s0 = {o[0], o[1], o[2], o[3]}, s1 = {o[4], o[5], o[6], o[7]};
                const unsigned voff = (unsigned)((w0 + r) * 64 + 16 * h);
                __builtin_amdgcn_raw_buffer_store_b128(s0, ry, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(s1, ry, voff + 32, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
#pragma unroll
        for (int i = 0; i < AU_SITER; ++i) xc[i] = xn[i];
        in_c = in_n;
        t = tn;
    }
}

}  // namespace

// Shapes the kernel serves: rows a multiple of 16 (its tiles), 32-bit byte offsets inside an output row band.
extern "C" int mgx_audio_conv3_bf16_supported(int64_t nimg, int h, int w) {
    if (nimg <= 0 || h < 16 || h % 16 || w < 8) return 0;
    if ((long)w * 64 >= (1L << 31)) return 0;
    const long ntiles = (long)nimg * (h / AU_TH) * ((w + AU_TW - 1) / AU_TW);
    return ntiles < (1L << 31) ? 1 : 0;
}

extern "C" int mgx_audio_conv3_bf16(int64_t nimg, int h, int w, const float* x, const float* wa, const float* ba,
                                    float alpha_a, const float* wb, const float* bb, float alpha_b, void* y,
                                    void* stream) {
    if (!x || !wa || !ba || !wb || !bb || !y || !mgx_audio_conv3_bf16_supported(nimg, h, w)) return MGX_ERR_ARG;
    if (!(alpha_a > 0.f) || !(alpha_b > 0.f)) return MGX_ERR_ARG;
    AudioDesc d;
    d.nimg = (int)nimg; d.h = h; d.w = w; d.alpha_a = alpha_a; d.alpha_b = alpha_b;
    const int tiles_h = h / AU_TH, tiles_w = (w + AU_TW - 1) / AU_TW;
    const long ntiles = (long)nimg * tiles_h * tiles_w;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    static_assert(AU_LDS <= 160 * 1024, "the kernel's LDS image exceeds a CU");
    if (hipFuncSetAttribute((const void*)baudio_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AU_LDS) != hipSuccess)
        return MGX_ERR_LAUNCH;
    hipLaunchKernelGGL(baudio_kernel, dim3(grid), dim3(AU_NTHR), AU_LDS, (hipStream_t)stream, d, x, wa, ba, wb, bb,
                       (bf16_t*)y, ntiles, tiles_h, tiles_w);
    return mgx_launch_status();
}
