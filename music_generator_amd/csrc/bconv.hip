// bf16 convolutions of the discriminator's grow blocks (trainer/models.py:256-281): stride-1 'same'
// 3x3 / 2x2 / 1x1 convolutions between 32- and 128-channel NHWC bf16 tensors, the 2x2 stride-2 'valid'
// convolution, and the data-gradient of each, on v_mfma_f32_32x32x16_bf16 with float32 accumulation.
//
// One workgroup of 8 waves per CU walks tiles of 8 rows x 64 columns; wave w owns row w as two
// accumulator tiles D[output channel][position] (32 x 32 each): the output channels are the MFMA rows,
// so that after one exchange between the lane halves every lane holds 16 consecutive channels of one
// position and stores them as 32 contiguous bytes.
// LDS holds
//   * the whole weight tensor, converted once per workgroup from the float32 master copy (Keras
//     layout, read through the descriptor's strides) to bf16 in MFMA-fragment order
//     [chunk][tap][k-step][lane][8], so the A operand is one lane-linear ds_read_b128;
//   * two input tiles with halo ((8+kh-1) x (64+kw-1) positions x 32 channels = 64 B a position, the
//     four 16-byte quads of a position XOR-swizzled with bits 2-3 of the position index, which makes
//     the B-operand ds_read_b128 of 32 consecutive positions conflict-free): while the MFMAs run
//     on one, the next tile's global loads are in flight into registers and are written to the other
//     buffer afterwards; one barrier per step.
// 128 input (output) channels run as 4 steps per tile over 32-channel slices with the accumulator
// (the staged tile) kept.  Positions outside the image read as zeros through out-of-range buffer
// offsets; columns past the row end are dropped by the store's range check.
// Epilogues: bias + tanh(x)+x/alpha (+ optional store of the derivative for the backward pass);
// multiply by a bf16 tensor of the output's shape (the data-gradient of a convolution whose input came
// out of an activation: dY (.) f'), either the stored derivative itself or the activation's stored output, from
// which the derivative is recovered; plain; or (1x1 kernels, passes that keep no tape) bias + activation followed by
// the fade-in blend (1 - alpha) old + alpha y of trainer/models.py:30-35 with a bf16 tensor `old` of the output's
// shape, in place of a WeightedSum launch that would read the activation back; the same with `old` itself re-computed
// per element from the 2-channel audio it is a 1x1 convolution + activation of (8 bytes a position read instead of 256);
// or (128 output channels, backward passes that need no weight gradients) epilogue 4 followed by the data gradient of
// the from-audio 1x1 convolution in front: the 128-channel gradient is contracted with that layer's (2, 128) kernel
// in registers and only the two audio channels are written.
#include <cstdlib>
#include "bf16.h"
#include "../../include/mgx.h"

namespace {

// Tile shapes: TH rows (= waves of the workgroup) x NT * 32 columns.  8 x 64 (8 waves, two accumulator tiles
// a wave) and 16 x 32 (16 waves = 4 per SIMD, one accumulator tile a wave: twice the waves to cover the
// latencies of a step, 1.20 instead of 1.29 input rows and columns read per output with a 3x3 halo).
template <int KH, int KW, int TH, int NT>
constexpr int npix() { return (TH + KH - 1) * (NT * 32 + KW - 1); }

// byte offset of quad q (8 channels) of tile position p
__device__ __forceinline__ int xoff(int p, int q) { return p * 64 + ((q ^ ((p >> 2) & 3)) << 4); }

// The other input of the blend as the layer that produces it: Conv2D(128, 1x1) + SoftRectifier(alpha) on float32
// audio of the output's height and width (trainer/models.py:292-296), kernel (2, 128) and bias (128) float32.
struct BlendAudio {
    const float* audio;
    const float* w;
    const float* b;
    float inv_alpha;
};
// Epilogue 6: kernel (2, 128) of the from-audio layer whose output this data gradient arrives at, the gradient of
// its 2-channel float32 input (nimg, h, w, 2), and whether that is added to.
struct ToAudio {
    const float* w;
    float* dx;
    int accumulate;
};

template <int KH, int KW, int EPI, int TH, int NT>
__global__ __launch_bounds__(64 * TH) void bconv_kernel(mgx_conv_desc d, const bf16_t* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       bf16_t* __restrict__ y, bf16_t* __restrict__ dact,
                                                       const bf16_t* __restrict__ ymul, long ntiles, int tiles_h,
                                                       int tiles_w, int gin_w, int scat,
                                                       const float* __restrict__ blend_alpha, BlendAudio ba, ToAudio ta) {
    // EPI: 0 plain, 1 bias + activation (+ derivative), 2 multiply by ymul, 3 = 1 followed by the blend with ymul,
    // 4 multiply by the activation derivative recovered from ymul = the activation's stored OUTPUT (dact_from_y),
    // 5 = 3 with the blend's other input re-computed from the audio (ba) instead of read from ymul,
    // 6 = 4 contracted with the from-audio kernel (ta): nothing of the output's own shape is written
    constexpr bool ACT = EPI == 1 || EPI == 3 || EPI == 5, PRE = EPI == 2 || EPI == 3 || EPI == 4 || EPI == 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TW = NT * 32, NTHR = 64 * TH;
    constexpr int IH = TH + KH - 1, IW = TW + KW - 1, TAPS = KH * KW, NPIX = IH * IW;
    constexpr int SITER = (NPIX * 4 + NTHR - 1) / NTHR;      // 16-byte staging loads per thread
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.hin, W = d.win;
    const int kchunks = d.cin / 32, nchunks = d.ngemm / 32;   // one of them is 1
    const int S = kchunks * nchunks;
    const int K = d.cin, N = d.ngemm;
    unsigned char* wl = lds;                                   // [nchunk][tap][K/16][64 lanes][16 B]
    unsigned char* xt = lds + TAPS * K * N * 2;                // [2][NPIX][64 B]

    // ---- weights: float32 global -> bf16 fragments in LDS (once per workgroup) ----
    {
        const int total = TAPS * K * N;
        const bool n_fast = d.wsn == 1;
        for (int i = tid; i < total; i += NTHR) {
            int tap, k, n;
            if (n_fast) { n = i % N; const int t2 = i / N; k = t2 % K; tap = t2 / K; }
            else { k = i % K; const int t2 = i / K; n = t2 % N; tap = t2 / N; }
            const int wt = d.flip ? (TAPS - 1 - tap) : tap;
            const float v = w[(long)wt * d.wtap + (long)k * d.wsk + (long)n * d.wsn];
            const int nc = n >> 5, f = (nc * TAPS + tap) * (K >> 4) + (k >> 4);
            const int ln = ((k >> 3) & 1) * 32 + (n & 31);
            *(bf16_t*)(wl + ((f * 64 + ln) * 8 + (k & 7)) * 2) = (bf16_t)v;
        }
    }

    // ---- input tile staging: thread t moves 16-byte piece #(t + 512 i) = (position, quad) ----
    // Tile order.  Workgroup b runs on XCD b % 8 (each XCD has its own L2), and the halo of a tile is input its
    // neighbours read too: within every full round of 256 tiles the 32 workgroups of an XCD take 32 CONSECUTIVE
    // tiles, and consecutive tiles are vertical neighbours first (column-major inside an image), so that the
    // halo rows and most halo columns of a tile are found in the XCD's L2 instead of being fetched again.
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
        h0 = (int)(rem - tw * tiles_h) * TH;
        w0 = (int)tw * TW;
    };
    const int q4 = tid & 3;
    const unsigned band_bytes = gin_w > 0 ? (unsigned)(2 * IH) * gin_w * 64u : (unsigned)IH * W * d.cin * 2u;
    int rel[SITER], tyx[SITER];
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int px = (tid >> 2) + (NTHR / 4) * i;
        const int ty = px / IW, tx = px - ty * IW;
        rel[i] = gin_w > 0 ? ((ty * 2 * gin_w + 2 * tx) * 32 + q4 * 8) * 2 : ((ty * W + tx) * d.cin + q4 * 8) * 2;
        tyx[i] = px < NPIX ? ((ty << 16) | tx) : (0x7000 << 16);
    }
    u32x4 sv[SITER];
    auto stage_load = [&](long t, int kc) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        const int hb = h0 - d.pt, wb = w0 - d.pl;
        const int base = gin_w > 0 ? ((2 * w0 + (kc & 1)) * 32 + (kc >> 1) * gin_w * 32) * 2 : (wb * d.cin + kc * 32) * 2;
        const long band = gin_w > 0 ? ((img * 2 * H + 2 * h0) * (long)gin_w) * 32 : ((img * H + hb) * (long)W) * d.cin;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + band), 0, band_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const unsigned hi = (unsigned)((tyx[i] >> 16) + hb), wi = (unsigned)((tyx[i] & 0xffff) + wb);
            const unsigned off = (hi < (unsigned)H && wi < (unsigned)W) ? (unsigned)(base + rel[i]) : 0xfffffff0u;
            sv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto stage_write = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int px = (tid >> 2) + (NTHR / 4) * i;
            if ((NTHR / 4) * (i + 1) <= NPIX || px < NPIX) *(u32x4*)(buf + xoff(px, q4)) = sv[i];
        }
    };

    // biases in LDS: an accumulation starts from the bias of its output channels instead of from zero
    float* bl = (float*)(xt + 2 * NPIX * 64);
    if (ACT)
        for (int i = tid; i < N; i += NTHR) bl[i] = d.nbias > 0 ? bias[i] : 0.f;
    // the from-audio layer of the blend's other input: {w[0][c], w[1][c], bias[c], -} per output channel c
    f32x4* fa = (f32x4*)(bl + N);
    if (EPI == 5)
        for (int i = tid; i < 128; i += NTHR) fa[i] = (f32x4){ba.w[i], ba.w[128 + i], ba.b[i], 0.f};
    // epilogue 6: {w[0][c], w[1][c]} per channel c of the from-audio kernel
    f32x2* tw = (f32x2*)(bl + N);
    if (EPI == 6)
        for (int i = tid; i < 128; i += NTHR) tw[i] = (f32x2){ta.w[i], ta.w[128 + i]};
    float ta0[NT], ta1[NT];          // the lane's partial sums of the two audio channels over the chunks of a tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ta0[nt] = 0.f; ta1[nt] = 0.f; }

    long t = blockIdx.x;
    int oc = 0;
    if (t < ntiles) {
        stage_load(t, 0);
        stage_write(xt);
    }
    __syncthreads();
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    const float k1 = 1.0f + inv_alpha;
    int cur = 0;
    f32x16 acc[NT];
    // registers 4g .. 4g+3 of an accumulator tile are output channels 8g + 4h .. 8g + 4h + 3 of chunk nc_
    auto acc_init = [&](int nc_) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ACT) v = *(const f32x4*)(bl + nc_ * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][4 * g + j] = v[j];
        }
    };
    acc_init(0);

    // ---- epilogue of the wave's row of tile t_, output channel chunk nc_ ----
    // what the epilogue of this step needs from global memory (the lane's 2 x 32 bytes of `ymul`) is requested
    // before the MFMA phase, so that its latency is covered by it
    u32x4 pre_m[NT][2];
    f32x2 pre_a[NT];
    auto epilogue_addr = [&](long t_, int nc_, long& rowoff, unsigned& row_bytes, int& pstride, int& coff, int& w0) {
        long img; int h0;
        tile_of(t_, img, h0, w0);
        const int row = h0 + wvs;
        rowoff = scat ? ((img * 2 * H + 2 * row + (nc_ >> 1)) * (2L * W)) * 32 : ((img * H + row) * (long)W) * N;
        row_bytes = row < H ? (scat ? 2u * W * 64u : (unsigned)W * N * 2u) : 0u;
        pstride = scat ? 128 : N * 2;                   // bytes between consecutive columns
        coff = scat ? (nc_ & 1) * 64 + 16 * h : nc_ * 64 + 16 * h;    // the lane's first 16 bytes; the second 32 further
    };
    auto epilogue_prefetch = [&](long t_, int nc_) {
        if (EPI == 5) {
            long img; int h0, w0;
            tile_of(t_, img, h0, w0);
            const int row = h0 + wvs;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(ba.audio + ((img * H + row) * (long)W) * 2), 0, row < H ? (unsigned)W * 8u : 0u, 0x00020000);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                pre_a[nt] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, (unsigned)((w0 + nt * 32 + r) * 8), 0, 0));
        }
        if (PRE) {
            long rowoff; unsigned row_bytes; int pstride, coff, w0;
            epilogue_addr(t_, nc_, rowoff, row_bytes, pstride, coff, w0);
            const __amdgpu_buffer_rsrc_t rm =
                __builtin_amdgcn_make_buffer_rsrc((void*)(ymul + rowoff), 0, row_bytes, 0x00020000);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned voff = (unsigned)((w0 + nt * 32 + r) * pstride + coff);
                pre_m[nt][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, voff, 0, 0));
                pre_m[nt][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, voff + 32, 0, 0));
            }
        }
    };
    auto epilogue = [&](f32x16 (&av)[NT], long t_, int nc_) {
        long rowoff; unsigned row_bytes; int pstride, coff, w0;
        epilogue_addr(t_, nc_, rowoff, row_bytes, pstride, coff, w0);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + rowoff), 0, row_bytes, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) mfma_result_wait(av[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16& a = av[nt];
            acc_rows_interleaved(a);
            const unsigned voff = (unsigned)((w0 + nt * 32 + r) * pstride + coff);
            unsigned o[8], dd[8];
            if (ACT) {
                if (d.act_alpha > 0.f) {
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        float y0, y1, d0, d1;
                        act_bf16(a[i], inv_alpha, k1, y0, d0);
                        act_bf16(a[i + 1], inv_alpha, k1, y1, d1);
                        o[i >> 1] = bf_pack(y0, y1);
                        dd[i >> 1] = bf_pack(d0, d1);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i += 2) o[i >> 1] = bf_pack(a[i], a[i + 1]);
                }
                if (EPI == 3 || EPI == 5) {
                    const float al = blend_alpha[0], om = 1.0f - al;
                    u32x4 m0, m1;
                    if (EPI == 5) {
                        // the stand-alone from-audio kernel's arithmetic (two fused multiply-adds from the bias,
                        // activation, rounding to bf16) for the lane's channels 8h .. 8h+7 and 16+8h .. 16+8h+7 of chunk nc_
                        const float a0 = pre_a[nt][0], a1 = pre_a[nt][1], fk1 = 1.0f + ba.inv_alpha;
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            unsigned m[4];
#pragma unroll
                            for (int i = 0; i < 8; i += 2) {
                                const f32x4 c0 = fa[nc_ * 32 + 16 * half + 8 * h + i], c1 = fa[nc_ * 32 + 16 * half + 8 * h + i + 1];
                                float v0 = __builtin_fmaf(a1, c0[1], __builtin_fmaf(a0, c0[0], c0[2]));
                                float v1 = __builtin_fmaf(a1, c1[1], __builtin_fmaf(a0, c1[0], c1[2]));
                                float dv;
                                act_bf16(v0, ba.inv_alpha, fk1, v0, dv);
                                act_bf16(v1, ba.inv_alpha, fk1, v1, dv);
                                m[i >> 1] = bf_pack(v0, v1);
                            }
                            if (half == 0) m0 = (u32x4){m[0], m[1], m[2], m[3]};
                            else m1 = (u32x4){m[0], m[1], m[2], m[3]};
                        }
                    } else {
                        m0 = pre_m[nt][0];
                        m1 = pre_m[nt][1];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = bf_pack(mgx_blend(om, bf_lo(m0[i]), al, bf_lo(o[i])), mgx_blend(om, bf_hi(m0[i]), al, bf_hi(o[i])));
                        o[4 + i] = bf_pack(mgx_blend(om, bf_lo(m1[i]), al, bf_lo(o[4 + i])),
                                           mgx_blend(om, bf_hi(m1[i]), al, bf_hi(o[4 + i])));
                    }
                }
            } else if (EPI == 2) {
                const u32x4 m0 = pre_m[nt][0], m1 = pre_m[nt][1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = bf_pack(a[2 * i] * bf_lo(m0[i]), a[2 * i + 1] * bf_hi(m0[i]));
                    o[4 + i] = bf_pack(a[8 + 2 * i] * bf_lo(m1[i]), a[8 + 2 * i + 1] * bf_hi(m1[i]));
                }
            } else if (EPI == 4 || EPI == 6) {
                const u32x4 m0 = pre_m[nt][0], m1 = pre_m[nt][1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = bf_pack(a[2 * i] * dact_from_y(bf_lo(m0[i]), inv_alpha, k1),
                                   a[2 * i + 1] * dact_from_y(bf_hi(m0[i]), inv_alpha, k1));
                    o[4 + i] = bf_pack(a[8 + 2 * i] * dact_from_y(bf_lo(m1[i]), inv_alpha, k1),
                                       a[8 + 2 * i + 1] * dact_from_y(bf_hi(m1[i]), inv_alpha, k1));
                }
                if (EPI == 6) {
                    // the values a store would have rounded to bf16, times the from-audio kernel: channels
                    // nc_ * 32 + 8h .. + 7 (o[0..3]) and nc_ * 32 + 16 + 8h .. + 7 (o[4..7])
                    if (nc_ == 0) { ta0[nt] = 0.f; ta1[nt] = 0.f; }
#pragma unroll
                    for (int half = 0; half < 2; ++half)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c = nc_ * 32 + 16 * half + 8 * h + 2 * i;
                            const f32x2 wa = tw[c], wb = tw[c + 1];
                            const float lo = bf_lo(o[4 * half + i]), hi = bf_hi(o[4 * half + i]);
                            ta0[nt] = __builtin_fmaf(hi, wb[0], __builtin_fmaf(lo, wa[0], ta0[nt]));
                            ta1[nt] = __builtin_fmaf(hi, wb[1], __builtin_fmaf(lo, wa[1], ta1[nt]));
                        }
                    if (nc_ == nchunks - 1) {
                        // the two lanes of a position hold the two halves of its channels
                        const float s0 = ta0[nt] + __shfl_xor(ta0[nt], 32), s1 = ta1[nt] + __shfl_xor(ta1[nt], 32);
                        long img; int h0, w0_;
                        tile_of(t_, img, h0, w0_);
                        const int row = h0 + wvs, col = w0_ + nt * 32 + r;
                        if (h == 0 && row < H && col < W) {
                            float* dst = ta.dx + ((img * H + row) * (long)W + col) * 2;
                            f32x2 v = {s0, s1};
                            if (ta.accumulate) { const f32x2 old = *(const f32x2*)dst; v[0] += old[0]; v[1] += old[1]; }
                            *(f32x2*)dst = v;
                        }
                    }
                    continue;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i += 2) o[i >> 1] = bf_pack(a[i], a[i + 1]);
            }
            u32x4 s0 = {o[0], o[1], o[2], o[3]}, s1 = {o[4], o[5], o[6], o[7]};
            __builtin_amdgcn_raw_buffer_store_b128(s0, ry, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(s1, ry, voff + 32, 0, 0);
            if (EPI == 1 && dact != nullptr && d.act_alpha > 0.f) {
                const __amdgpu_buffer_rsrc_t rd =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(dact + rowoff), 0, row_bytes, 0x00020000);
                u32x4 e0 = {dd[0], dd[1], dd[2], dd[3]}, e1 = {dd[4], dd[5], dd[6], dd[7]};
                __builtin_amdgcn_raw_buffer_store_b128(e0, rd, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(e1, rd, voff + 32, 0, 0);
            }
        }
    };

    while (t < ntiles) {
        int ocn = oc + 1;
        long tn = t;
        if (ocn == S) { ocn = 0; tn = t + gridDim.x; }
        const bool has_next = tn < ntiles;
        const bool need_x = has_next && (kchunks > 1 || ocn == 0);
        if (need_x) stage_load(tn, kchunks > 1 ? ocn : 0);
        const bool ep_now = kchunks == 1 || oc == S - 1;
        if (ep_now && (PRE || EPI == 5)) epilogue_prefetch(t, nchunks > 1 ? oc : 0);
        const unsigned char* xb = xt + cur * (NPIX * 64);
        const int kc = kchunks > 1 ? oc : 0, nc = nchunks > 1 ? oc : 0;
        const unsigned char* wb = wl + ((nc * TAPS * (K >> 4) + kc * 2) * 64 + lane) * 16;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ta = tap / KW, tb = tap - ta * KW;
            const int p0 = (wvs + ta) * IW + tb + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const u32x4 a = *(const u32x4*)(wb + (tap * (K >> 4) + ks) * 1024);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = mfma_bf16(a, *(const u32x4*)(xb + xoff(p0 + 32 * nt, 2 * ks + h)), acc[nt]);
            }
        }
        if (need_x) stage_write(xt + (cur ^ 1) * (NPIX * 64));
        if (ep_now) {
            epilogue(acc, t, nc);
            acc_init(nchunks > 1 ? ocn : 0);
        }
        __syncthreads();
        if (need_x) cur ^= 1;
        t = tn;
        oc = ocn;
    }
}

template <int KH, int KW, int TH, int NT>
int launch_shape(const mgx_conv_desc& d, const bf16_t* x, const float* w, const float* bias, bf16_t* y, bf16_t* dact,
                 const bf16_t* ymul, hipStream_t st, int gin_w, int scat, const float* blend_alpha = nullptr,
                 const BlendAudio* audio = nullptr, const ToAudio* to_audio = nullptr) {
    constexpr int TW = NT * 32;
    const int tiles_h = (d.hin + TH - 1) / TH, tiles_w = (d.win + TW - 1) / TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    if (ntiles <= 0 || ntiles >= (1L << 31)) return MGX_ERR_ARG;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    const int lds = KH * KW * d.cin * d.ngemm * 2 + 2 * npix<KH, KW, TH, NT>() * 64 + d.ngemm * 4 +
                    (audio ? 128 * 16 : 0) + (to_audio ? 128 * 8 : 0);
    if (lds > 160 * 1024) return MGX_ERR_ARG;
    const BlendAudio ba = audio ? *audio : BlendAudio{nullptr, nullptr, nullptr, 0.f};
    const ToAudio ta = to_audio ? *to_audio : ToAudio{nullptr, nullptr, 0};
    const int epi = to_audio ? 6 : blend_alpha != nullptr ? (audio ? 5 : 3)
                    : (ymul != nullptr ? (d.act_alpha > 0.f ? 4 : 2) : ((d.nbias > 0 || d.act_alpha > 0.f) ? 1 : 0));
#define MGX_B(E)                                                                                                     \
    do {                                                                                                             \
        if (hipFuncSetAttribute((const void*)bconv_kernel<KH, KW, E, TH, NT>,                                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)                      \
            return MGX_ERR_LAUNCH;                                                                                   \
        hipLaunchKernelGGL((bconv_kernel<KH, KW, E, TH, NT>), dim3(grid), dim3(64 * TH), lds, st, d, x, w, bias, y,  \
                           dact, ymul, ntiles, tiles_h, tiles_w, gin_w, scat, blend_alpha, ba, ta);                  \
    } while (0)
    if (epi == 3) {
        if constexpr (KH == 1 && KW == 1) MGX_B(3);
        else return MGX_ERR_ARG;
    } else if (epi == 5) {
        if constexpr (KH == 1 && KW == 1) MGX_B(5);
        else return MGX_ERR_ARG;
    } else if (epi == 6) {
        if constexpr (KH == 3 && KW == 3) MGX_B(6);
        else return MGX_ERR_ARG;
    } else if (epi == 4) MGX_B(4);
    else if (epi == 2) MGX_B(2);
    else if (epi == 1) MGX_B(1);
    else MGX_B(0);
#undef MGX_B
    return mgx_launch_status();
}

// 16 x 32 tiles (16 waves) where the image has the rows to fill them and both tile buffers fit beside the
// weights; 8 x 64 tiles otherwise.  MGX_BCONV_TILE=8 / 16 forces a shape (measurements).
template <int KH, int KW>
int launch_b(const mgx_conv_desc& d, const bf16_t* x, const float* w, const float* bias, bf16_t* y, bf16_t* dact,
             const bf16_t* ymul, hipStream_t st, int gin_w, int scat, const float* blend_alpha = nullptr,
             const BlendAudio* audio = nullptr, const ToAudio* to_audio = nullptr) {
    const int lds16 = KH * KW * d.cin * d.ngemm * 2 + 2 * npix<KH, KW, 16, 1>() * 64 + d.ngemm * 4;
    bool tall = d.hin >= 16 && d.hin % 16 == 0 && lds16 <= 160 * 1024;
    if (const char* e = getenv("MGX_BCONV_TILE")) tall = e[0] == '1' && lds16 <= 160 * 1024;
    if (tall) return launch_shape<KH, KW, 16, 1>(d, x, w, bias, y, dact, ymul, st, gin_w, scat, blend_alpha, audio, to_audio);
    return launch_shape<KH, KW, 8, 2>(d, x, w, bias, y, dact, ymul, st, gin_w, scat, blend_alpha, audio, to_audio);
}

}  // namespace

// Which contractions the bf16 tile kernel serves: stride-1 'same' convolutions with kernel 3x3, 2x2 or 1x1
// (either tap order) and the 2x2 stride-2 'valid' convolution, between 32 and 32 or 128 channels (not 128
// on both sides), and the depth-to-space data-gradient of the strided one.
extern "C" int mgx_conv_bf16_supported(const mgx_conv_desc* d) {
    if (!d) return 0;
    const bool chans = (d->cin == 32 && (d->ngemm == 32 || d->ngemm == 128)) || (d->cin == 128 && d->ngemm == 32);
    if (d->sh == 2 && d->sw == 2 && d->kh == 2 && d->kw == 2 && d->dkh == 1 && d->dkw == 1)
        return d->cin == 32 && d->ngemm == 32 && d->pt == 0 && d->pl == 0 && d->hin == 2 * d->hout &&
               d->win == 2 * d->wout && d->cout == 32;
    if (d->dkh == 2 && d->dkw == 2)
        return d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->cin == 32 && d->ngemm == 128 &&
               d->cout == 32 && d->hin == d->hout && d->win == d->wout;
    if (d->dkh != 1 || d->dkw != 1 || d->sh != 1 || d->sw != 1 || !chans || d->cout != d->ngemm) return 0;
    if (d->hin != d->hout || d->win != d->wout) return 0;
    if (!((d->kh == 3 && d->kw == 3) || (d->kh == 2 && d->kw == 2) || (d->kh == 1 && d->kw == 1))) return 0;
    if (d->pt < 0 || d->pt >= d->kh || d->pl < 0 || d->pl >= d->kw) return 0;
    return (d->wsk == 1 || d->wsn == 1) ? 1 : 0;
}

extern "C" int mgx_conv_gemm_bf16(const mgx_conv_desc* dp, const void* x, const float* w, const float* bias, void* y,
                                  void* dact, const void* ymul, void* stream) {
    if (!dp || !x || !w || !y || !mgx_conv_bf16_supported(dp)) return MGX_ERR_ARG;
    if (dp->accumulate) return MGX_ERR_ARG;
    if (dp->nbias > 0 && (!bias || dp->nbias != dp->ngemm)) return MGX_ERR_ARG;
    if (ymul && (dp->nbias > 0 || dact)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    mgx_conv_desc d = *dp;
    const bf16_t* xb = (const bf16_t*)x;
    bf16_t* yb = (bf16_t*)y;
    bf16_t* db = (bf16_t*)dact;
    const bf16_t* mb = (const bf16_t*)ymul;
    if (d.sh == 2) {
        // 2x2 stride-2 convolution = 1x1 convolution over the 128 channels of the gathered window
        d.hin = dp->hout; d.win = dp->wout; d.cin = 128; d.kh = d.kw = 1; d.sh = d.sw = 1;
        return launch_b<1, 1>(d, xb, w, bias, yb, db, mb, st, dp->win, 0);
    }
    if (d.dkh == 2) return launch_b<1, 1>(d, xb, w, bias, yb, db, mb, st, 0, 1);
    if (d.kh == 3) return launch_b<3, 3>(d, xb, w, bias, yb, db, mb, st, 0, 0);
    if (d.kh == 2) return launch_b<2, 2>(d, xb, w, bias, yb, db, mb, st, 0, 0);
    return launch_b<1, 1>(d, xb, w, bias, yb, db, mb, st, 0, 0);
}

// The 1x1 convolution + SoftRectifier that ends a discriminator block (trainer/models.py:280-281) and the
// WeightedSum behind it (models.py:297) in one launch, for passes that keep no tape:
// y = (1 - alpha) old + alpha act(conv(x)), every intermediate rounded to bf16 where the two launches round it.
extern "C" int mgx_conv1x1_blend_bf16(const mgx_conv_desc* dp, const void* x, const float* w, const float* bias,
                                      const void* old, const float* alpha, void* y, void* stream) {
    if (!dp || !x || !w || !old || !alpha || !y || !mgx_conv_bf16_supported(dp)) return MGX_ERR_ARG;
    if (dp->kh != 1 || dp->kw != 1 || dp->sh != 1 || dp->dkh != 1 || dp->accumulate) return MGX_ERR_ARG;
    if (dp->nbias > 0 && (!bias || dp->nbias != dp->ngemm)) return MGX_ERR_ARG;
    return launch_b<1, 1>(*dp, (const bf16_t*)x, w, bias, (bf16_t*)y, nullptr, (const bf16_t*)old, (hipStream_t)stream, 0, 0,
                          alpha);
}

// The same launch with `old` given as the layer that produces it in the fade-in graph (models.py:292-296): the
// Conv2D(128, 1x1) + SoftRectifier(audio_alpha) on the pooled 2-channel float32 audio (nimg, h, w, 2), kernel
// audio_w (2, 128), bias audio_b (128).  Each element of `old` is re-computed in the epilogue with the arithmetic of
// mgx_pointwise2_bf16_fwd and rounded to bf16 as that kernel stores it: bit-identical to the three launches.
extern "C" int mgx_conv1x1_blend_audio_bf16(const mgx_conv_desc* dp, const void* x, const float* w, const float* bias,
                                            const float* audio, const float* audio_w, const float* audio_b,
                                            float audio_alpha, const float* alpha, void* y, void* stream) {
    if (!dp || !x || !w || !audio || !audio_w || !audio_b || !alpha || !y || !mgx_conv_bf16_supported(dp)) return MGX_ERR_ARG;
    if (dp->kh != 1 || dp->kw != 1 || dp->sh != 1 || dp->dkh != 1 || dp->accumulate || dp->ngemm != 128) return MGX_ERR_ARG;
    if (dp->nbias > 0 && (!bias || dp->nbias != dp->ngemm)) return MGX_ERR_ARG;
    if (!(audio_alpha > 0.f)) return MGX_ERR_ARG;
    const BlendAudio ba = {audio, audio_w, audio_b, 1.0f / audio_alpha};
    return launch_b<1, 1>(*dp, (const bf16_t*)x, w, bias, (bf16_t*)y, nullptr, nullptr, (hipStream_t)stream, 0, 0, alpha, &ba);
}

// Data gradient of the block's first 3x3 convolution (128 -> 32 channels) AND of the from-audio 1x1 convolution +
// SoftRectifier in front of it (trainer/models.py:256-259), for backward passes that need no weight gradients (the
// generator step's sweep through the discriminator, models.py:144): `d` is the 3x3 layer's data-gradient descriptor
// (32 -> 128 channels), g the gradient with respect to its pre-activation, `y128` the from-audio layer's stored
// OUTPUT (the derivative is recovered from it, d->act_alpha that layer's alpha), audio_w its kernel (2, 128); dx
// (nimg, h, w, 2) float32 receives (accumulate: is added) the gradient of the audio.  What mgx_conv_gemm_bf16 (ymul)
// would store as bf16 is contracted with audio_w in registers instead: the 128-channel gradient is never written.
extern "C" int mgx_conv_dgrad_to_audio_bf16(const mgx_conv_desc* dp, const void* g, const float* w, const void* y128,
                                            const float* audio_w, float* dx, int accumulate, void* stream) {
    if (!dp || !g || !w || !y128 || !audio_w || !dx || !mgx_conv_bf16_supported(dp)) return MGX_ERR_ARG;
    if (dp->kh != 3 || dp->kw != 3 || dp->sh != 1 || dp->dkh != 1 || dp->accumulate || dp->cin != 32 || dp->ngemm != 128 ||
        dp->nbias > 0 || !(dp->act_alpha > 0.f))
        return MGX_ERR_ARG;
    const ToAudio ta = {audio_w, dx, accumulate ? 1 : 0};
    return launch_b<3, 3>(*dp, (const bf16_t*)g, w, nullptr, nullptr, nullptr, (const bf16_t*)y128, (hipStream_t)stream, 0, 0,
                          nullptr, nullptr, &ta);
}
