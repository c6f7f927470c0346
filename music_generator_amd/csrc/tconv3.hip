// Stride-1 'same' convolution (4x4 / 3x3 / 2x2 / 1x1) of one 32-channel input slice into one 32-channel
// output slice on v_mfma_f32_32x32x2_f32 — forward and data-gradient of the discriminator convs.
//
// One workgroup of 8 waves per CU walks tiles of 8 rows x 32 columns; wave w owns row w (one
// 32x32 accumulator tile).  LDS holds
//   * the 32x32 weight slice of every tap, once per launch, as [tap][k/4][n][4] so that a lane
//     (column n, half h) fetches four consecutive k with one conflict-free ds_read_b128;
//   * two input tiles with halo ((8+kh-1) x (32+kw-1) positions x 32 channels, 36-float stride):
//     while the MFMAs run on one, the next tile's global loads are already in flight into
//     registers (issued before the compute phase) and are written to the other buffer after it,
//     so HBM/L2 latency hides under the matrix work; one barrier per tile.
// The two waves that share a SIMD drift apart between barriers, which overlaps one wave's
// epilogue (bias, tanh(x)+x/alpha, stores) with the other's MFMAs.  Register use stays far below
// the 256 a 2-waves-per-SIMD kernel may use: no spills, no scratch traffic.
// Convs with 128 input (output) channels take 4 steps per tile over 32-channel slices: the
// accumulator persists over input slices (output slices re-use the staged tile); the next
// slice's weights travel through registers like the next tile and replace the LDS copy between
// two barriers.
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int PSTR = 36;
// A tile is 256 output positions: TH = 8 rows x 32 columns (wave w owns row w) or, for images whose
// height is a multiple of 4 only (the 4-row base block), TH = 4 rows x 64 columns (wave w owns row
// (w & 1) + 2 (w >> 2) of column half (w >> 1) & 1).  In an image of exactly 4 rows the tap rows that read
// nothing but padding are skipped: 1 of 3 for the outer output rows of a 3x3 kernel, 4 of the 16 of a 4x4.

// The 16 weight slices of a 4x4 kernel leave room for one input tile only: its next tile waits in
// registers until every wave is done with the current one (one more barrier per tile).
template <int KH>
constexpr int input_buffers() { return KH == 4 ? 1 : 2; }

template <int KH, int KW, int TH>
constexpr int lds_floats() {
    return KH * KW * 1024 + input_buffers<KH>() * (TH + KH - 1) * (256 / TH + KW - 1) * PSTR;
}

template <int KH, int KW, bool WT, bool MUL, int TH>
__global__ __launch_bounds__(512, 2) void tconv3_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                        const float* __restrict__ xmul,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ dact, long ntiles, int tiles_h,
                                                        int tiles_w, int gin_w, int scat,
                                                        const float* __restrict__ blend_old,
                                                        const float* __restrict__ blend_alpha) {
    // blend_alpha != nullptr (1x1 kernels, passes that keep no tape): the output is the fade-in blend
    // (1 - alpha) blend_old + alpha y of trainer/models.py:30-35 instead of y, blend_old of the output's shape.
    // gin_w > 0 (1x1 kernels only): the 128 input channels of position (i, j) are the 2x2 window at
    // (2i, 2j) of a 32-channel image gin_w columns wide, slice kc = tap (kc / 2, kc % 2): a 2x2
    // stride-2 convolution.  scat (1x1 kernels, 128 outputs): output channel block nc goes to
    // position (2i + nc / 2, 2j + nc % 2) of a 32-channel image twice as large: its data-gradient.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TW = 256 / TH;
    constexpr int IH = TH + KH - 1, IW = TW + KW - 1, TAPS = KH * KW, NPIX = IH * IW;
    constexpr int NBUF = input_buffers<KH>();
    constexpr int SITER = (NPIX * 8 + 511) / 512;      // float4 staging loads per thread (input tile)
    constexpr int WITER = (TAPS * 256 + 511) / 512;    // float4 staging stores per thread (weights)
    float* wl = lds;                                   // [TAPS][8][32][4]
    float* xt = lds + TAPS * 1024;                     // [NBUF][NPIX][36]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win;
    const int kchunks = d.cin / 32, nchunks = d.ngemm / 32;     // one of them is 1
    const int S = kchunks * nchunks;                            // steps per tile

    // ---- weight slice (kc, nc) of every tap: global -> registers -> LDS ----
    f32x4 wreg[WITER];
    auto w_load = [&](int oc) {
        const int kc = kchunks > 1 ? oc : 0, nc = nchunks > 1 ? oc : 0;
        const long woff = (long)kc * 32 * d.wsk + (long)nc * 32 * d.wsn;
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = min(tid + 512 * it, TAPS * 256 - 1);
            const int tap = i >> 8, kq = (i >> 5) & 7, n = i & 31;
            const int wt = d.flip ? (TAPS - 1 - tap) : tap;
            const float* wp = w + (long)wt * d.wtap + woff;
            if (WT) wreg[it] = *(const f32x4*)(wp + (long)n * d.wsn + kq * 4);          // [n][k] rows
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) wreg[it][j] = wp[(long)(kq * 4 + j) * d.wsk + n];   // [k][n] rows
            }
        }
    };
    auto w_write = [&]() {
#pragma unroll
        for (int it = 0; it < WITER; ++it) {
            const int i = tid + 512 * it;
            if (i < TAPS * 256) *(f32x4*)(wl + i * 4) = wreg[it];
        }
    };

    // ---- input tile: thread t moves float4 #(t + 512 i), i < SITER ----
    // Addressing goes through a buffer resource per tile that spans the IH image rows of the tile
    // window (< 4 GiB): a 32-bit byte offset per load, and positions outside the image get an
    // out-of-range offset, for which the hardware returns zeros - no clamping, no selects, no
    // 64-bit address arithmetic per element.
    const int c4 = tid & 7;
    // Tile order: workgroup b runs on XCD b % 8 (own L2); within every full round of 256 tiles the 32
    // workgroups of an XCD take 32 consecutive tiles, vertical neighbours first (column-major inside an
    // image), so that the halo of a tile is found in the XCD's L2 instead of being fetched from HBM again.
    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        unsigned tu = (unsigned)t;                          // ntiles < 2^31 (checked by the launcher)
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
    const unsigned band_bytes = gin_w > 0 ? (unsigned)(2 * IH) * gin_w * 32u * 4u : (unsigned)IH * W * d.cin * 4u;
    int rel[SITER], tyx[SITER];          // byte offset of the element inside the tile window; (ty << 16) | tx
#pragma unroll
    for (int i = 0; i < SITER; ++i) {
        const int px = (tid >> 3) + 64 * i;
        const int ty = px / IW, tx = px - ty * IW;
        rel[i] = gin_w > 0 ? ((ty * 2 * gin_w + 2 * tx) * 32 + c4 * 4) * 4 : ((ty * W + tx) * d.cin + c4 * 4) * 4;
        tyx[i] = px < NPIX ? ((ty << 16) | tx) : (0x7000 << 16);      // never inside an image
    }
    f32x4 sv[SITER], sm[SITER];
    auto stage_load = [&](long t, int kc) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        const int hb = h0 - d.pt, wb = w0 - d.pl;
        const int base = gin_w > 0 ? ((2 * w0 + (kc & 1)) * 32 + (kc >> 1) * gin_w * 32) * 4 : (wb * d.cin + kc * 32) * 4;
        // first row of the window (may lie above the image)
        const long band = gin_w > 0 ? ((img * 2 * H + 2 * h0) * (long)gin_w) * 32 : ((img * H + hb) * (long)W) * d.cin;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + band), 0, band_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rm = rx;
        if (MUL) rm = __builtin_amdgcn_make_buffer_rsrc((void*)(xmul + band), 0, band_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const unsigned hi = (unsigned)((tyx[i] >> 16) + hb), wi = (unsigned)((tyx[i] & 0xffff) + wb);
            const unsigned off = (hi < (unsigned)H && wi < (unsigned)W) ? (unsigned)(base + rel[i]) : 0xfffffff0u;
            sv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            if (MUL) sm[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, off, 0, 0));
        }
    };
    auto stage_write = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < SITER; ++i) {
            const int px = (tid >> 3) + 64 * i;
            f32x4 v = sv[i];
            if (MUL) v *= sm[i];
            if (64 * (i + 1) <= NPIX || px < NPIX) *(f32x4*)(buf + px * PSTR + c4 * 4) = v;
        }
    };

    long t = blockIdx.x;
    int oc = 0;
    w_load(0);
    w_write();
    if (t < ntiles) {
        stage_load(t, 0);
        stage_write(xt);
    }
    __syncthreads();
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    int cur = 0;
    // An accumulation starts from the bias of the lane's output channel (column r of chunk nc) instead of from
    // zero: the up to four values are loaded once, before the tile loop (a load inside the epilogue stands in
    // front of it with its whole latency, a quarter of the tile time of the 2x2 and 1x1 convolutions).
    float bl4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bl4[j] = (d.nbias > 0 && j < nchunks) ? bias[j * 32 + r] : 0.f;
    auto bias_of = [&](int nc_) { return nc_ == 0 ? bl4[0] : (nc_ == 1 ? bl4[1] : (nc_ == 2 ? bl4[2] : bl4[3])); };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bl4[0];

    // ---- epilogue of row wv of tile t_, output channels [32 nc, 32 nc + 32) ----
    // loads (accumulate), arithmetic and stores run in separate sweeps: 16 independent memory
    // operations in flight instead of one round trip per element
    // Stores go through a buffer resource that spans exactly the output row of this wave: columns
    // past the row end are dropped by the hardware range check, the per-register part of the
    // offset is a scalar, so a store costs no vector ALU work.
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    // the wave's row and first column in the tile.  On 4-row tiles the two waves of a SIMD (w and w + 4) take
    // rows p and p + 2 of one column half: in an image of exactly 4 rows every output row has tap rows that
    // fall into the padding (skipped below), and with this pairing each SIMD is left the same number of them.
    const int wrow = TH == 8 ? wvs : ((wvs & 1) + 2 * (wvs >> 2)), wcol = TH == 8 ? 0 : ((wvs >> 1) & 1) * 32;
    // tap rows [ta_lo, ta_hi) of this wave's output row read image rows; the others read zeros only
    int ta_lo = 0, ta_hi = KH;
    if (TH == 4 && H == 4 && gin_w == 0 && !scat) {
        ta_lo = max(0, d.pt - wrow);
        ta_hi = min(KH, 4 + d.pt - wrow);
    }
    auto epilogue = [&](const f32x16& a, long t_, int oc_) {
        const int n = (nchunks > 1 ? oc_ : 0) * 32 + r;
        long img; int h0, w0;
        tile_of(t_, img, h0, w0);
        const int nc_ = nchunks > 1 ? oc_ : 0;
        const long rowoff = scat ? ((img * 2 * H + 2 * (h0 + wrow) + (nc_ >> 1)) * (2L * W)) * 32
                                 : ((img * H + h0 + wrow) * (long)W) * d.ngemm;
        const unsigned row_bytes = scat ? 2u * W * 32u * 4u : (unsigned)W * d.ngemm * 4u;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + rowoff), 0, row_bytes, 0x00020000);
        const unsigned voff = scat ? (unsigned)(((2 * (w0 + wcol + 4 * h) + (nc_ & 1)) * 32 + r) * 4)
                                   : (unsigned)(((w0 + wcol + 4 * h) * d.ngemm + n) * 4);
        const int rstep = scat ? 2 * 32 * 4 : d.ngemm * 4;        // bytes per output position
        const bool act = d.act_alpha > 0.f;
        float v[16], der[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) v[reg] = a[reg];
        if (act) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float tt, s2;
                tanh_sech2(v[reg], tt, s2);
                der[reg] = s2 + inv_alpha;
                v[reg] = tt + v[reg] * inv_alpha;
            }
        }
        if constexpr (KH == 1) {
            if (blend_alpha != nullptr) {
                const float al = blend_alpha[0], om = 1.0f - al;
                const __amdgpu_buffer_rsrc_t ro =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(blend_old + rowoff), 0, row_bytes, 0x00020000);
                float old[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    old[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             ro, voff, ((reg & 3) + 8 * (reg >> 2)) * rstep, 0));
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] = mgx_blend(om, old[reg], al, v[reg]);
            }
        }
        if (d.accumulate) {
            float old[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                old[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                         ry, voff, ((reg & 3) + 8 * (reg >> 2)) * rstep, 0));
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) v[reg] += old[reg];
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[reg]), ry, voff,
                                                  ((reg & 3) + 8 * (reg >> 2)) * rstep, 0);
        if (act && dact) {
            const __amdgpu_buffer_rsrc_t rd =
                __builtin_amdgcn_make_buffer_rsrc((void*)(dact + rowoff), 0, row_bytes, 0x00020000);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, der[reg]), rd, voff,
                                                      ((reg & 3) + 8 * (reg >> 2)) * rstep, 0);
        }
    };
    // The two waves of a SIMD (w and w + 4) are kept half a phase apart: the upper four waves carry
    // their finished accumulator over the barrier and run its epilogue at the start of the next
    // step, while the lower four are already issuing MFMAs; at the end of the step the roles swap.
    const bool late = wvs >= 4;
    f32x16 pend;
    long pend_t = 0;
    int pend_oc = 0;
    bool pend_valid = false;
    while (t < ntiles) {
        if (pend_valid) {
            epilogue(pend, pend_t, pend_oc);
            pend_valid = false;
        }
        // the step after this one: next output/input chunk of the same tile, or the next tile
        int ocn = oc + 1;
        long tn = t;
        if (ocn == S) { ocn = 0; tn = t + gridDim.x; }
        const bool has_next = tn < ntiles;
        const bool need_x = has_next && (kchunks > 1 || ocn == 0);
        const bool need_w = has_next && S > 1;
        if (need_x) stage_load(tn, kchunks > 1 ? ocn : 0);     // in flight during the MFMA phase
        if (need_w) w_load(ocn);
        const float* xb = xt + cur * (NPIX * PSTR);
        // 4*TAPS quads of (4 input channels x 32 outputs); the LDS reads of quad q+2 are issued
        // before the MFMAs of quad q, so their latency is covered by 8 MFMAs of this wave
        {
            constexpr int NQ = 4 * TAPS;
            const float* abase = xb + (wrow * IW + wcol + r) * PSTR + 16 * h;
            const float* wbase = wl + (4 * h) * 128 + r * 4;
            f32x4 aq[3], wq[3];
            auto rd = [&](int q, int slot) {
                const int tap = q >> 2, j = q & 3;
                const int ta = tap / KW, tb = tap - ta * KW;
                aq[slot] = *(const f32x4*)(abase + (ta * IW + tb) * PSTR + 4 * j);
                wq[slot] = *(const f32x4*)(wbase + tap * 1024 + j * 128);
            };
            if (TH == 4 && KH > 1) {
                // one software pipeline per tap row, so that a row of padding is skipped as a whole
                constexpr int RQ = 4 * KW;
#pragma unroll
                for (int ta = 0; ta < KH; ++ta) {
                    if (ta >= ta_lo && ta < ta_hi) {
                        rd(ta * RQ, 0);
                        rd(ta * RQ + 1, 1);
#pragma unroll
                        for (int q = 0; q < RQ; ++q) {
                            if (q + 2 < RQ) rd(ta * RQ + q + 2, (q + 2) % 3);
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc = mfma32(aq[q % 3][s], wq[q % 3][s], acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
                rd(0, 0);
                rd(1, 1);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if (q + 2 < NQ) rd(q + 2, (q + 2) % 3);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = mfma32(aq[q % 3][s], wq[q % 3][s], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (S > 1 || NBUF == 1) __syncthreads();        // every wave is done with the weight slice (the input tile)
        if (need_x) stage_write(xt + (NBUF == 1 ? 0 : (cur ^ 1)) * (NPIX * PSTR));
        if (need_w) w_write();
        // the stores of the epilogue come after the staging writes: the wait for the staged loads
        // must not also wait for stores that were issued a moment ago
        if (kchunks == 1 || oc == S - 1) {
            if (late) {
                pend = acc;
                pend_t = t;
                pend_oc = oc;
                pend_valid = true;
            } else {
                epilogue(acc, t, oc);
            }
            const float b_next = bias_of(nchunks > 1 ? ocn : 0);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[reg] = b_next;
        }
        __syncthreads();
        if (need_x && NBUF == 2) cur ^= 1;
        t = tn;
        oc = ocn;
    }
    if (pend_valid) epilogue(pend, pend_t, pend_oc);
}

template <int KH, int KW, int TH = 8>
void launch3(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias, float* y,
             float* dact, hipStream_t st, int gin_w = 0, int scat = 0, const float* blend_old = nullptr,
             const float* blend_alpha = nullptr) {
    constexpr int TW = 256 / TH;
    const int tiles_h = d.hin / TH, tiles_w = (d.win + TW - 1) / TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
    constexpr int lds = lds_floats<KH, KW, TH>() * 4;
    const bool wt = d.wsk == 1, mul = xmul != nullptr;
    // more than 64 KiB of dynamic LDS must be allowed per kernel function
#define MGX_T3(WW, MM)                                                                                               \
    do {                                                                                                             \
        static std::atomic<uint64_t> allowed{0};                                                                     \
        mgx_allow_dynamic_lds(tconv3_kernel<KH, KW, WW, MM, TH>, lds, allowed);                                      \
        hipLaunchKernelGGL((tconv3_kernel<KH, KW, WW, MM, TH>), dim3(grid), dim3(512), lds, st, d, x, xmul, w, bias, y,   \
                           dact, ntiles, tiles_h, tiles_w, gin_w, scat, blend_old, blend_alpha);                     \
    } while (0)
    if (wt) { if (mul) MGX_T3(true, true); else MGX_T3(true, false); }
    else { if (mul) MGX_T3(false, true); else MGX_T3(false, false); }
#undef MGX_T3
}

}  // namespace

// Runs the convolution described by d; applicability is checked by mgx_tconv_applicable (tconv.hip).
int mgx_tconv3_launch(const mgx_conv_desc& d0, const float* x, const float* xmul, const float* w, const float* bias,
                      float* y, float* dact, hipStream_t st, const float* blend_old, const float* blend_alpha) {
    mgx_conv_desc d = d0;
    if (blend_alpha != nullptr) {          // plain 1x1 convolutions only
        if (!blend_old || d.kh != 1 || d.kw != 1 || d.sh != 1 || d.dkh != 1 || d.accumulate) return MGX_ERR_ARG;
        if (d.hin % 8) launch3<1, 1, 4>(d, x, xmul, w, bias, y, dact, st, 0, 0, blend_old, blend_alpha);
        else launch3<1, 1>(d, x, xmul, w, bias, y, dact, st, 0, 0, blend_old, blend_alpha);
        return mgx_launch_status();
    }
    if (d.sh == 2) {
        // 2x2 stride-2 convolution = 1x1 convolution over the 128 channels of the gathered window
        d.hin = d0.hout; d.win = d0.wout; d.cin = 128; d.kh = d.kw = 1; d.sh = d.sw = 1;
        launch3<1, 1>(d, x, xmul, w, bias, y, dact, st, d0.win, 0);
        return mgx_launch_status();
    }
    if (d.dkh == 2) {
        launch3<1, 1>(d, x, xmul, w, bias, y, dact, st, 0, 1);
        return mgx_launch_status();
    }
    if (d.kh == 4) {                      // 32 -> 32 only (mgx_tconv_applicable), always on 4-row tiles
        launch3<4, 4, 4>(d, x, xmul, w, bias, y, dact, st);
        return mgx_launch_status();
    }
    if (d.hin % 8) {                      // 4-row tiles: the 4 x 750 base block
        if (d.kh == 3) launch3<3, 3, 4>(d, x, xmul, w, bias, y, dact, st);
        else if (d.kh == 2) launch3<2, 2, 4>(d, x, xmul, w, bias, y, dact, st);
        else launch3<1, 1, 4>(d, x, xmul, w, bias, y, dact, st);
        return mgx_launch_status();
    }
    if (d.kh == 3) launch3<3, 3>(d, x, xmul, w, bias, y, dact, st);
    else if (d.kh == 2) launch3<2, 2>(d, x, xmul, w, bias, y, dact, st);
    else launch3<1, 1>(d, x, xmul, w, bias, y, dact, st);
    return mgx_launch_status();
}
