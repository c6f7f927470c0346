// Weight / bias gradient of the bf16 discriminator convolutions (tf.GradientTape, trainer/models.py:118)
// on v_mfma_f32_32x32x16_bf16:
//     dW[tap][ci][co] += sum_positions X[position + tap][ci] * G[position][co],   dB[co] += sum G
// with X the layer input and G the gradient with respect to the pre-activation, both NHWC bf16.
//
// The contraction index is the position, so both MFMA operands are needed channel-major while the
// tensors are position-major: tiles are staged to LDS as they lie in memory ([position][32 channels],
// 64 B a position) and read with ds_read_b64_tr_b16, which hands a lane four consecutive positions of
// its channel.  A workgroup of 4 waves (one per SIMD, up to 512 registers each) walks tiles of 8 rows x
// 64 columns; a wave owns two rows and keeps one 32x32 accumulator tile per tap (plus one for the bias,
// an MFMA against a matrix of ones) for the whole launch.  The gradient fragment of 16 positions is read
// once and used for every tap.  Two LDS buffers: the next tile's global loads are in flight during the
// MFMAs of the current one.
// Tensors with 128 channels on one side are split over workgroups in 32-channel chunks: the four
// chunks of a tile slot are given to blocks b, b+8, b+16, b+24, which share an XCD under round-robin
// placement, so the operand they have in common is fetched from HBM once (speed only).
// At the end the four waves are added in LDS in fixed order, every workgroup leaves one slab in the
// workspace, and a second kernel adds the slabs in fixed order into dW / dB (bitwise reproducible).
#include "bf16.h"
#include <type_traits>
#include "../../include/mgx.h"

namespace {

constexpr int WTH = 8;

template <int KH, int KW, bool S2>
struct WGeo {
    static constexpr int TWG = S2 ? 32 : 64;
    static constexpr int XH = S2 ? 2 * WTH : WTH + KH - 1;
    static constexpr int XW = S2 ? 2 * TWG : TWG + KW - 1;
    static constexpr int XPIX = XH * XW, GPIX = WTH * TWG, TAPS = KH * KW;
    static constexpr int LDS = 2 * (XPIX + GPIX) * 64;
};

template <int KH, int KW, bool S2>
__global__ __launch_bounds__(256, 1) void bwgrad_kernel(mgx_conv_desc d, const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ g, float* __restrict__ partial,
                                                        long ntiles, int tiles_h, int tiles_w, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using G = WGeo<KH, KW, S2>;
    constexpr int TWG = G::TWG, XW = G::XW, XPIX = G::XPIX, GPIX = G::GPIX, TAPS = G::TAPS;
    constexpr int XITER = (XPIX * 4 + 255) / 256, GITER = (GPIX * 4 + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hx = d.hin, Wx = d.win, Hg = d.hout, Wg = d.wout;
    const int chunk = (blockIdx.x >> 3) % chunks;
    const int slot = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * chunks));
    const int nslots = gridDim.x / chunks;
    const int xc0 = d.cin > 32 ? chunk * 32 : 0, gc0 = d.ngemm > 32 ? chunk * 32 : 0;
    unsigned char* xbuf = lds;                       // [2][XPIX][64 B]
    unsigned char* gbuf = lds + 2 * XPIX * 64;       // [2][GPIX][64 B]

    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        const unsigned tu = (unsigned)t;
        const unsigned im = tu / per_img;
        const unsigned rem = tu - im * per_img;
        const unsigned th = rem / (unsigned)tiles_w;
        img = im;
        h0 = (int)th * WTH;
        w0 = (int)(rem - th * tiles_w) * TWG;
    };
    const int q4 = tid & 3;
    u32x4 sx[XITER], sg[GITER];
    auto stage_load = [&](long t) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        // x window
        const int hb = S2 ? 2 * h0 : h0 - d.pt, wb = S2 ? 2 * w0 : w0 - d.pl;
        const long xband = ((img * Hx + hb) * (long)Wx) * d.cin;
        const unsigned xbytes = (unsigned)G::XH * Wx * d.cin * 2u;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + xband), 0, xbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < XITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            const int ty = px / XW, tx = px - ty * XW;
            const unsigned hi = (unsigned)(ty + hb), wi = (unsigned)(tx + wb);
            const bool ok = px < XPIX && hi < (unsigned)Hx && wi < (unsigned)Wx;
            const unsigned off = ok ? (unsigned)(((ty * Wx + (tx + wb)) * d.cin + xc0 + q4 * 8) * 2) : 0xfffffff0u;
            sx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
        // gradient tile
        const long gband = ((img * Hg + h0) * (long)Wg) * d.ngemm;
        const unsigned gbytes = (unsigned)WTH * Wg * d.ngemm * 2u;
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(g + gband), 0, gbytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < GITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            const int ty = px / TWG, tx = px - ty * TWG;
            const bool ok = px < GPIX && (unsigned)(ty + h0) < (unsigned)Hg && (unsigned)(tx + w0) < (unsigned)Wg;
            const unsigned off = ok ? (unsigned)(((ty * Wg + (tx + w0)) * d.ngemm + gc0 + q4 * 8) * 2) : 0xfffffff0u;
            sg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
        }
    };
    auto stage_write = [&](int b) {
#pragma unroll
        for (int i = 0; i < XITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            if (64 * (i + 1) <= XPIX || px < XPIX) *(u32x4*)(xbuf + (b * XPIX + px) * 64 + q4 * 16) = sx[i];
        }
#pragma unroll
        for (int i = 0; i < GITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            if (64 * (i + 1) <= GPIX || px < GPIX) *(u32x4*)(gbuf + (b * GPIX + px) * 64 + q4 * 16) = sg[i];
        }
    };

    f32x16 acc[TAPS + 1];
#pragma unroll
    for (int a = 0; a <= TAPS; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // transposed-read lane geometry: 16-lane group gq = lane >> 4 serves channel block gq & 1 and k half
    // gq >> 1; lane 4q + p of a group supplies the address of position q, channels 4p .. 4p+3
    const int hk = lane >> 5, cb = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
    const int col_l = 8 * hk + qq;                       // + 16 ks + 4 s
    const int ch_off = (16 * cb + 4 * pp) * 2;

    long t = slot;
    int cur = 0;
    if (t < ntiles) {
        stage_load(t);
        stage_write(0);
    }
    __syncthreads();
    while (t < ntiles) {
        const long tn = t + nslots;
        const bool has_next = tn < ntiles;
        if (has_next) stage_load(tn);
        const unsigned char* xb = xbuf + cur * XPIX * 64 + ch_off;
        const unsigned char* gb = gbuf + cur * GPIX * 64 + ch_off;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int wr = 2 * wvs + rr;
#pragma unroll
            for (int ks = 0; ks < TWG / 16; ++ks) {
                const int col = 16 * ks + col_l;
                const unsigned char* gp = gb + (wr * TWG + col) * 64;
                const u32x2 b0 = tr_read(gp), b1 = tr_read(gp + 4 * 64);
                const u32x4 bfrag = {b0[0], b0[1], b1[0], b1[1]};
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int ta = tap / KW, tb = tap - ta * KW;
                    const unsigned char* xp = S2 ? xb + ((2 * wr + ta) * XW + 2 * col + tb) * 64
                                                 : xb + ((wr + ta) * XW + col + tb) * 64;
                    const u32x2 a0 = tr_read(xp), a1 = tr_read(xp + (S2 ? 8 : 4) * 64);
                    const u32x4 afrag = {a0[0], a0[1], a1[0], a1[1]};
                    acc[tap] = mfma_bf16(afrag, bfrag, acc[tap]);
                }
                acc[TAPS] = mfma_bf16(ones, bfrag, acc[TAPS]);
            }
        }
        if (has_next) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        t = tn;
    }

    // ---- add the four waves in fixed order: (w0 + w2) + (w1 + w3), through LDS ----
    float* red = (float*)lds;                           // [2][TAPS+1][16][64] floats <= 80 KB
    __syncthreads();
    if (wvs >= 2) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wvs - 2) * (TAPS + 1) + a) * 16 + i) * 64 + lane] = acc[a][i];
    }
    __syncthreads();
    if (wvs < 2) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] += red[((wvs * (TAPS + 1) + a) * 16 + i) * 64 + lane];
    }
    __syncthreads();
    if (wvs == 1) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(a * 16 + i) * 64 + lane] = acc[a][i];
    }
    __syncthreads();
    if (wvs == 0) {
        float* dst = partial + ((long)(slot * chunks + chunk) * (TAPS + 1)) * 1024;
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(a * 16 + i) * 64 + lane] = acc[a][i] + red[(a * 16 + i) * 64 + lane];
    }
}

// dW[tap*wtap + k*wsk + n*wsn] += sum over slots of the slab element; block = 64 elements x 16 slot
// groups (group q adds slots q, q+16, ... in order), then a fixed tree over the groups.  Elements past
// the weights are the bias gradient (row 0 of the bias tile of chunk 0, or of every chunk when the
// chunks split the output channels).
__global__ __launch_bounds__(1024) void bwgrad_reduce_kernel(mgx_conv_desc d, const float* __restrict__ partial,
                                                             float* __restrict__ dw, float* __restrict__ db,
                                                             int nslots, int chunks, int taps) {
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + e;
    const long wtotal = (long)chunks * taps * 1024;
    const int nb = d.ngemm;                              // bias elements
    const long total = wtotal + (db ? nb : 0);
    float s = 0.f;
    long dst = -1;
    bool is_bias = false;
    if (idx < total) {
        int chunk, tile, i, j;
        if (idx < wtotal) {
            chunk = (int)(idx / (taps * 1024));
            const int rem = (int)(idx - (long)chunk * taps * 1024);
            tile = rem >> 10;
            i = (rem >> 5) & 31;
            j = rem & 31;
            const int k = (d.cin > 32 ? chunk * 32 : 0) + i, n = (d.ngemm > 32 ? chunk * 32 : 0) + j;
            dst = (long)tile * d.wtap + (long)k * d.wsk + (long)n * d.wsn;
        } else {
            const int n = (int)(idx - wtotal);
            chunk = d.ngemm > 32 ? n >> 5 : 0;
            tile = taps;
            i = 0;
            j = n & 31;
            dst = n;
            is_bias = true;
        }
        const int lane = j + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3);
        const float* src = partial + ((long)chunk * (taps + 1) + tile) * 1024 + reg * 64 + lane;
        const long stride = (long)chunks * (taps + 1) * 1024;
        for (int sp = q; sp < nslots; sp += 16) s += src[sp * stride];
    }
    red[q][e] = s;
    __syncthreads();
    if (q == 0 && idx < total) {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = red[k][e];
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; ++k) t[k] += t[k + w];
        if (is_bias) db[dst] += t[0];
        else dw[dst] += t[0];
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient AND weight gradient of one 32 -> 32 'same' convolution in one pass (the two read the same two
// tensors: the layer input X, which is also the stored output of the layer before it, and the gradient G with
// respect to this layer's pre-activation):
//     dW[tap][ci][co] += sum_q X[q][ci] G[q - tap + pad][co]          dB[co] += sum_q G[q][co]
//     dX[q][ci]        = sum_tap sum_co W[tap][ci][co] G[q - tap + pad][co]     (x f'(X[q][ci]) when X is the output of a
//                                                                               SoftRectifier layer: the gradient the
//                                                                               layer before wants)
// Both contract G shifted by the tap, so the G window (tile + halo) is staged once and serves both, and the X tile
// serves the weight gradient (channel-major through ds_read_b64_tr_b16) and the activation derivative of the
// epilogue: 128 B read + 64 B written per position instead of 192 + 128 + 64 for the two separate kernels.
// Geometry of bwgrad_kernel: 4 waves, tiles of 8 rows x 64 columns, a wave owns two rows, one accumulator tile per
// tap for the whole launch, plus one 32-position data-gradient tile at a time (D[ci][position]: the weights are the
// A operand, a register fragment per tap and 16-channel step; the B operand is 16 bytes of a staged position).
// Staged positions are 64 B with their four 16-byte chunks XOR-swizzled by bits 2-3 of the position index, which
// keeps both kinds of read conflict-free.
// ------------------------------------------------------------------------------------------
// compile-time unrolled loop: f(integral_constant<int, I>) for I in [I0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// CIN = 128 (the block's first 3x3 convolution, 128 -> 32): the tile is 4 rows x 32 columns, X is staged as four
// planes of 32 channels, and wave c owns input-channel chunk c for the WHOLE tile: its 32 of the 128 data-gradient
// channels and its own weight-gradient tiles (no sum over the waves at the end), the G window shared by all four.
template <int KH, int KW, int CIN>
struct FGeo {
    static constexpr int NCH = CIN / 32;
    static constexpr int TW = CIN == 128 ? 32 : 64, TH = CIN == 128 ? 4 : WTH;
    static constexpr int GH = TH + KH - 1, GW = TW + KW - 1;
    static constexpr int GPIX = GH * GW, XPIX = TH * TW, TAPS = KH * KW;
    static constexpr int LDS = 2 * (GPIX + NCH * XPIX) * 64;
};

__device__ __forceinline__ int sw_off(int p, int c16) { return p * 64 + ((c16 ^ ((p >> 2) & 3)) << 4); }

template <int KH, int KW, bool ACT, int CIN>
__global__ __launch_bounds__(256, 1) void bbwd_kernel(mgx_conv_desc d, const bf16_t* __restrict__ x,
                                                      const bf16_t* __restrict__ g, const float* __restrict__ w,
                                                      bf16_t* __restrict__ dx, float* __restrict__ partial, long ntiles,
                                                      int tiles_h, int tiles_w, float act_alpha) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using G = FGeo<KH, KW, CIN>;
    constexpr int TW = G::TW, TH = G::TH, NCH = G::NCH, GW = G::GW, GPIX = G::GPIX, XPIX = G::XPIX, TAPS = G::TAPS;
    constexpr int GITER = (GPIX * 4 + 255) / 256, XITER = (XPIX * NCH * 4 + 255) / 256;
    // A staged row of the G window is GW positions of 64 B; the swizzle of a position's four 16-byte chunks is keyed by
    // its COLUMN (bits 2-3), so a read's address is a per-lane column term (computed once for the launch) plus a
    // term that is uniform over the wave (buffer, row, 32-column block) plus a constant (kernel row, k-step).
    constexpr int ROWB = GW * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win;
    const int slot = blockIdx.x, nslots = gridDim.x;
    const int cch = CIN == 128 ? wvs : 0;               // the wave's input-channel chunk
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;       // 'same' padding in front (Keras: the extra row of an even kernel goes behind)
    constexpr int gy0 = KH - 1 - PT, gx0 = KW - 1 - PL;       // rows / columns of halo in front of the tile
    unsigned char* gbuf = lds;                       // [2][GPIX][64 B]
    unsigned char* xbuf = lds + 2 * GPIX * 64;       // [2][NCH planes][XPIX][64 B]

    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        const unsigned tu = (unsigned)t;
        const unsigned im = tu / per_img;
        const unsigned rem = tu - im * per_img;
        const unsigned th = rem / (unsigned)tiles_w;
        img = im;
        h0 = (int)th * TH;
        w0 = (int)(rem - th * tiles_w) * TW;
    };
    const int q4 = tid & 3;
    u32x4 sg[GITER], sx[XITER];
    auto stage_load = [&](long t) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        const int hb = h0 - gy0, wb = w0 - gx0;
        const long gband = ((img * H + hb) * (long)W) * 32;
        const __amdgpu_buffer_rsrc_t rg =
            __builtin_amdgcn_make_buffer_rsrc((void*)(g + gband), 0, (unsigned)G::GH * W * 64u, 0x00020000);
#pragma unroll
        for (int i = 0; i < GITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            const int ty = px / GW, tx = px - ty * GW;
            const unsigned hi = (unsigned)(ty + hb), wi = (unsigned)(tx + wb);
            const bool ok = px < GPIX && hi < (unsigned)H && wi < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((ty * W + (tx + wb)) * 32 + q4 * 8) * 2) : 0xfffffff0u;
            sg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
        }
        // X tile: 4 NCH 16-byte pieces a position, consecutive threads on consecutive pieces
        const long xband = ((img * H + h0) * (long)W) * CIN;
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + xband), 0, (unsigned)TH * W * (CIN * 2u), 0x00020000);
#pragma unroll
        for (int i = 0; i < XITER; ++i) {
            const int pc = tid + 256 * i, px = pc / (4 * NCH), sub = pc - px * (4 * NCH);
            const int ty = px / TW, tx = px - ty * TW;
            const bool ok = px < XPIX && (unsigned)(ty + h0) < (unsigned)H && (unsigned)(tx + w0) < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((ty * W + (tx + w0)) * CIN + sub * 8) * 2) : 0xfffffff0u;
            sx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto stage_write = [&](int b) {
#pragma unroll
        for (int i = 0; i < GITER; ++i) {
            const int px = (tid >> 2) + 64 * i;
            const int ty = px / GW, tx = px - ty * GW;
            if (64 * (i + 1) <= GPIX || px < GPIX) *(u32x4*)(gbuf + b * GPIX * 64 + ty * ROWB + sw_off(tx, q4)) = sg[i];
        }
#pragma unroll
        for (int i = 0; i < XITER; ++i) {
            const int pc = tid + 256 * i, px = pc / (4 * NCH), sub = pc - px * (4 * NCH);
            if (256 * (i + 1) <= XPIX * NCH * 4 || px < XPIX)
                *(u32x4*)(xbuf + (b * NCH + (sub >> 2)) * XPIX * 64 + sw_off(px, sub & 3)) = sx[i];
        }
    };

    // weights W[tap][ci][co] (Keras layout, co fastest) as A fragments of the data gradient: lane (ci, h), k = co
    u32x4 wf[TAPS][2];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* p = w + tap * (CIN * 32) + (cch * 32 + r) * 32 + 16 * ks + 8 * h;
            const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
            wf[tap][ks] = (u32x4){bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
        }
    const float inv_alpha = act_alpha > 0.f ? 1.0f / act_alpha : 0.f, k1 = 1.0f + inv_alpha;

    f32x16 acc[TAPS + 1];
#pragma unroll
    for (int a = 0; a <= TAPS; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // transposed-read lane geometry (bwgrad_kernel): 16-lane group gq = lane >> 4 serves channel block gq & 1 and k
    // half gq >> 1; lane 4q + p of a group supplies the address of position q, channels 4p .. 4p+3
    const int hk = lane >> 5, cb = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
    const int col_l = 8 * hk + qq;
    const int tr_k = 2 * cb + (pp >> 1), tr_h = 8 * (pp & 1);
    int dB[KW][2], wB[KW][2], xB[2], yB[2];          // per-lane column terms: data-gradient operand, transposed G / X reads, epilogue
#pragma unroll
    for (int tb = 0; tb < KW; ++tb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dB[tb][ks] = sw_off(r + KW - 1 - tb, 2 * ks + h);
#pragma unroll
        for (int j = 0; j < 2; ++j) wB[tb][j] = sw_off(col_l + KW - 1 - tb + 4 * j, tr_k) + tr_h;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        xB[j] = sw_off(col_l + 4 * j, tr_k) + tr_h;
        yB[j] = sw_off(r, 2 * j + h);
    }
    auto tr_two = [&](const unsigned char* p0, const unsigned char* p1, int off) {     // positions c .. c+3 and c+4 .. c+7 of the lane's channels
        const u32x2 a0 = tr_read(p0 + off), a1 = tr_read(p1 + off);
        return (u32x4){a0[0], a0[1], a1[0], a1[1]};
    };

    long t = slot;
    int cur = 0;
    if (t < ntiles) {
        stage_load(t);
        stage_write(0);
    }
    __syncthreads();
    while (t < ntiles) {
        const long tn = t + nslots;
        const bool has_next = tn < ntiles;
        if (has_next) stage_load(tn);
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        // One 32-position block at a time (row wr, columns col0 .. col0 + 31), its MFMAs in stages of one kernel row:
        // KH stages of the data gradient, then 2 KH of the weight gradient.  One wave per SIMD, so nothing but this
        // wave's own instruction order overlaps LDS latency, vector ALU work and the matrix unit: the LDS reads that
        // prepare stage s + 1 are issued between the MFMAs of stage s (the first stage of the next block between
        // the last MFMAs of this one), and the epilogue of the block's data gradient (lane exchange, activation
        // derivative, rounding, stores) in pieces between the MFMAs of its weight gradient.  The scheduling barriers
        // pin that order.
        const int ugb = cur * GPIX * 64, uxb = (2 * GPIX + (cur * NCH + cch) * XPIX) * 64;     // uniform terms of the current buffers
        constexpr int NJ = 2 * KW, NSTAGE = 3 * KH, NWM = 2 * (TAPS + 1), NPIECE = 19, WM0 = 2;
        u32x4 dq[2][NJ], wq[2][KW], af[2], yv[2], o[2];
        f32x16 dacc;
        int ug = ugb + (CIN == 128 ? 0 : 2 * wvs) * ROWB, ux = 0, ugn = ug, oh = 0, ow = 0;
        // item I of the LDS reads that stage S works on (S = NSTAGE: stage 0 of the next block)
        auto prep = [&](auto S_, auto I_) {
            constexpr int S = decltype(S_)::value, i = decltype(I_)::value;
            if constexpr (S < KH || S >= NSTAGE) {
                constexpr int ta = S < KH ? S : 0, tb = i >> 1, ks = i & 1;
                dq[ta & 1][i] = *(const u32x4*)(lds + (dB[tb][ks] + (S < KH ? ug : ugn)) + (KH - 1 - ta) * ROWB);
            } else {
                constexpr int st = S - KH, ks = st / KH, ta = st - ks * KH, first = ta == 0 ? 1 : 0;
                if constexpr (first == 1 && i == 0) {
                    af[ks & 1] = tr_two(lds + (xB[0] + ux), lds + (xB[1] + ux), 16 * ks * 64);
                } else {
                    constexpr int tb = i - first;
                    wq[st & 1][tb] = tr_two(lds + (wB[tb][0] + ug), lds + (wB[tb][1] + ug), (KH - 1 - ta) * ROWB + 16 * ks * 64);
                }
            }
        };
        // piece I of the epilogue of the block's data gradient
        auto piece = [&](auto I_) {
            constexpr int i = decltype(I_)::value;
            if constexpr (i < 8) {
                // lane exchange (acc_rows_interleaved, one swap a piece): lane (position r, h) ends up with channels
                // 8h .. 8h+7 in dacc[0..7] and 16+8h .. 16+8h+7 in dacc[8..15]
                constexpr int g = i >> 2, e = i & 3;
                float lo = dacc[8 * g + e], hi = dacc[8 * g + 4 + e];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                dacc[8 * g + e] = lo;
                dacc[8 * g + 4 + e] = hi;
            } else if constexpr (i == 8) {
                yv[0] = *(const u32x4*)(lds + (yB[0] + ux));
                yv[1] = *(const u32x4*)(lds + (yB[1] + ux));
            } else {
                constexpr int half = (i - 9) / 5, q = (i - 9) - 5 * half;
                if constexpr (q < 4) {
                    float v0 = dacc[8 * half + 2 * q], v1 = dacc[8 * half + 2 * q + 1];
                    if constexpr (ACT) {     // x f'(X): the layer before is a SoftRectifier layer and X its stored output
                        v0 *= dact_from_y(bf_lo(yv[half][q]), inv_alpha, k1);
                        v1 *= dact_from_y(bf_hi(yv[half][q]), inv_alpha, k1);
                    }
                    o[half][q] = bf_pack(v0, v1);
                } else if (oh < H && ow < W) {
                    bf16_t* dst = dx + (((img * H + oh) * (long)W + ow) * CIN) + cch * 32;
                    *(u32x4*)(dst + 16 * half + 8 * h) = o[half];
                }
            }
        };
        // MFMA M of stage S (the centre tap's second one sums the unshifted window for the bias gradient), with the
        // reads and epilogue pieces that go around it
        auto step = [&](auto S_, auto M_) {
            constexpr int S = decltype(S_)::value, m = decltype(M_)::value;
            constexpr bool dstage = S < KH;
            constexpr int st = dstage ? 0 : S - KH, ks = st / KH, ta = dstage ? S : st - ks * KH;
            constexpr int nm = dstage ? NJ : KW + (ta == PT ? 1 : 0);
            constexpr int nl = (S + 1 < KH || S + 1 >= NSTAGE) ? NJ : KW + ((S + 1 - KH) % KH == 0 ? 1 : 0);
            constexpr int l0 = (m * nl + nm - 1) / nm, l1 = ((m + 1) * nl + nm - 1) / nm;
            static_for<l0, l1>([&](auto I_) { prep(std::integral_constant<int, S + 1>{}, I_); });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (dstage) {
                dacc = mfma_bf16(wf[ta * KW + (m >> 1)][m & 1], dq[ta & 1][m], dacc);
            } else {
                constexpr bool bias = ta == PT && m == PL + 1;
                constexpr int tb = (ta == PT && m > PL) ? m - 1 : m;
                if constexpr (bias) acc[TAPS] = mfma_bf16(ones, wq[st & 1][tb], acc[TAPS]);
                else acc[ta * KW + tb] = mfma_bf16(af[ks & 1], wq[st & 1][tb], acc[ta * KW + tb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!dstage) {
                constexpr int wm = ks * (TAPS + 1) + ta * KW + (ta > PT ? 1 : 0) + m;     // weight-gradient MFMAs of the block so far
                if constexpr (wm >= WM0) {
                    constexpr int p0 = ((wm - WM0) * NPIECE) / (NWM - WM0), p1 = ((wm - WM0 + 1) * NPIECE) / (NWM - WM0);
                    static_for<p0, p1>(piece);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        static_for<0, NJ>([&](auto I_) { prep(std::integral_constant<int, NSTAGE>{}, I_); });
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int wr = CIN == 128 ? it : 2 * wvs + (it >> 1), col0 = CIN == 128 ? 0 : 32 * (it & 1);
            const int itn = (it + 1) & 3;                                      // the last block prepares the first again: unused, in range
            ug = ugb + wr * ROWB + col0 * 64;
            ux = uxb + (wr * TW + col0) * 64;
            ugn = CIN == 128 ? ugb + itn * ROWB : ugb + (2 * wvs + (itn >> 1)) * ROWB + 32 * (itn & 1) * 64;
            oh = h0 + wr;
            ow = w0 + col0 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) dacc[i] = 0.f;
            static_for<0, NSTAGE>([&](auto S_) {
                constexpr int S = decltype(S_)::value;
                constexpr int nm = S < KH ? NJ : KW + ((S - KH) % KH == PT ? 1 : 0);
                static_for<0, nm>([&](auto M_) { step(S_, M_); });
            });
        }
        if (has_next) stage_write(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        t = tn;
    }

    if constexpr (CIN == 128) {
        // every wave holds the tiles of its own input-channel chunk: slab (slot, chunk) of the workspace
        float* dst = partial + ((long)(slot * NCH + cch) * (TAPS + 1)) * 1024;
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(a * 16 + i) * 64 + lane] = acc[a][i];
        return;
    }
    // ---- add the four waves in fixed order: (w0 + w2) + (w1 + w3), through LDS ----
    float* red = (float*)lds;                           // [2][TAPS+1][16][64] floats <= 80 KB
    __syncthreads();
    if (wvs >= 2) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wvs - 2) * (TAPS + 1) + a) * 16 + i) * 64 + lane] = acc[a][i];
    }
    __syncthreads();
    if (wvs < 2) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][i] += red[((wvs * (TAPS + 1) + a) * 16 + i) * 64 + lane];
    }
    __syncthreads();
    if (wvs == 1) {
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(a * 16 + i) * 64 + lane] = acc[a][i];
    }
    __syncthreads();
    if (wvs == 0) {
        float* dst = partial + ((long)slot * (TAPS + 1)) * 1024;
#pragma unroll
        for (int a = 0; a <= TAPS; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(a * 16 + i) * 64 + lane] = acc[a][i] + red[(a * 16 + i) * 64 + lane];
    }
}

constexpr int WGRID = 256;

template <int KH, int KW, bool S2>
int launch_w(const mgx_conv_desc& d, const bf16_t* x, const bf16_t* g, float* dw, float* db, float* partial,
             hipStream_t st) {
    using G = WGeo<KH, KW, S2>;
    const int chunks = (d.cin > 32 ? d.cin / 32 : 1) * (d.ngemm > 32 ? d.ngemm / 32 : 1);
    const int tiles_h = (d.hout + WTH - 1) / WTH, tiles_w = (d.wout + G::TWG - 1) / G::TWG;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    if (ntiles <= 0 || ntiles >= (1L << 31)) return MGX_ERR_ARG;
    if (hipFuncSetAttribute((const void*)bwgrad_kernel<KH, KW, S2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            G::LDS) != hipSuccess)
        return MGX_ERR_LAUNCH;
    hipLaunchKernelGGL((bwgrad_kernel<KH, KW, S2>), dim3(WGRID), dim3(256), G::LDS, st, d, x, g, partial, ntiles,
                       tiles_h, tiles_w, chunks);
    const int nslots = WGRID / chunks;
    const long total = (long)chunks * G::TAPS * 1024 + (db ? d.ngemm : 0);
    hipLaunchKernelGGL(bwgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, st, d, partial, dw, db,
                       nslots, chunks, G::TAPS);
    return mgx_launch_status();
}

int wgrad_taps(const mgx_conv_desc& d) { return d.kh * d.kw; }

template <int KH, int KW, int CIN>
int launch_f(const mgx_conv_desc& d, const bf16_t* x, const bf16_t* g, const float* w, float* dw, float* db, bf16_t* dx,
             float* partial, float act_alpha, hipStream_t st) {
    using G = FGeo<KH, KW, CIN>;
    const int tiles_h = (d.hout + G::TH - 1) / G::TH, tiles_w = (d.wout + G::TW - 1) / G::TW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    if (ntiles <= 0 || ntiles >= (1L << 31)) return MGX_ERR_ARG;
    const void* fn = act_alpha > 0.f ? (const void*)bbwd_kernel<KH, KW, true, CIN> : (const void*)bbwd_kernel<KH, KW, false, CIN>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return MGX_ERR_LAUNCH;
    if (act_alpha > 0.f)
        hipLaunchKernelGGL((bbwd_kernel<KH, KW, true, CIN>), dim3(WGRID), dim3(256), G::LDS, st, d, x, g, w, dx, partial,
                           ntiles, tiles_h, tiles_w, act_alpha);
    else
        hipLaunchKernelGGL((bbwd_kernel<KH, KW, false, CIN>), dim3(WGRID), dim3(256), G::LDS, st, d, x, g, w, dx, partial,
                           ntiles, tiles_h, tiles_w, act_alpha);
    const long total = (long)G::NCH * G::TAPS * 1024 + (db ? d.ngemm : 0);
    hipLaunchKernelGGL(bwgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, st, d, partial, dw, db,
                       WGRID, G::NCH, G::TAPS);
    return mgx_launch_status();
}

}  // namespace

extern "C" int mgx_conv_bf16_supported(const mgx_conv_desc* d);

// Bytes of workspace mgx_conv_wgrad_bf16 needs for d (0: unsupported descriptor).
extern "C" int64_t mgx_conv_wgrad_bf16_workspace(const mgx_conv_desc* d) {
    if (!d || !mgx_conv_bf16_supported(d) || d->dkh != 1 || d->flip) return 0;
    return (int64_t)WGRID * (wgrad_taps(*d) + 1) * 1024 * 4;
}

extern "C" int mgx_conv_wgrad_bf16(const mgx_conv_desc* dp, const void* x, const void* g, float* dw, float* dbias,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dp || !x || !g || !dw || !workspace) return MGX_ERR_ARG;
    const int64_t need = mgx_conv_wgrad_bf16_workspace(dp);
    if (need == 0) return MGX_ERR_ARG;
    if (workspace_bytes < need) return MGX_ERR_WORKSPACE;
    if (dbias && dp->nbias != dp->ngemm) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bf16_t* xb = (const bf16_t*)x;
    const bf16_t* gb = (const bf16_t*)g;
    float* part = (float*)workspace;
    const mgx_conv_desc& d = *dp;
    if (d.sh == 2) return launch_w<2, 2, true>(d, xb, gb, dw, dbias, part, st);
    if (d.kh == 3) return launch_w<3, 3, false>(d, xb, gb, dw, dbias, part, st);
    if (d.kh == 2) return launch_w<2, 2, false>(d, xb, gb, dw, dbias, part, st);
    return launch_w<1, 1, false>(d, xb, gb, dw, dbias, part, st);
}

// The fused backward serves the 3x3 and 2x2 'same' convolutions between 32 and 32 channels and the 3x3 one from 128
// to 32 channels, stride 1, weights in the Keras layout of the forward descriptor (tap, input channel, output
// channel; output channel fastest).
extern "C" int mgx_conv_bwd_bf16_supported(const mgx_conv_desc* d) {
    if (!d || !mgx_conv_bf16_supported(d)) return 0;
    if ((d->cin != 32 && d->cin != 128) || d->ngemm != 32 || d->cout != 32 || d->sh != 1 || d->sw != 1 || d->dkh != 1 ||
        d->dkw != 1 || d->flip)
        return 0;
    if (!((d->kh == 3 && d->kw == 3) || (d->kh == 2 && d->kw == 2 && d->cin == 32))) return 0;
    if (d->hin != d->hout || d->win != d->wout || d->pt != (d->kh - 1) / 2 || d->pl != (d->kw - 1) / 2) return 0;
    if ((long)(WTH + 2) * d->win * d->cin * 2 >= (1L << 31)) return 0;       // byte offsets inside a tile's band of rows stay in 32 bits
    return d->wsn == 1 && d->wsk == 32 && d->wtap == 32 * d->cin;
}

extern "C" int64_t mgx_conv_bwd_bf16_workspace(const mgx_conv_desc* d) {
    return mgx_conv_bwd_bf16_supported(d) ? (int64_t)(d->cin / 32) * mgx_conv_wgrad_bf16_workspace(d) : 0;
}

extern "C" int mgx_conv_bwd_bf16(const mgx_conv_desc* dp, const void* x, const void* g, const float* w, float* dw,
                                 float* dbias, void* dx, float act_alpha, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
    if (!dp || !x || !g || !w || !dw || !dx || !workspace || !mgx_conv_bwd_bf16_supported(dp)) return MGX_ERR_ARG;
    if (workspace_bytes < mgx_conv_bwd_bf16_workspace(dp)) return MGX_ERR_WORKSPACE;
    if (dbias && dp->nbias != dp->ngemm) return MGX_ERR_ARG;
    if (!(act_alpha >= 0.f)) return MGX_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dp->cin == 128)
        return launch_f<3, 3, 128>(*dp, (const bf16_t*)x, (const bf16_t*)g, w, dw, dbias, (bf16_t*)dx, (float*)workspace, act_alpha, st);
    if (dp->kh == 3)
        return launch_f<3, 3, 32>(*dp, (const bf16_t*)x, (const bf16_t*)g, w, dw, dbias, (bf16_t*)dx, (float*)workspace, act_alpha, st);
    return launch_f<2, 2, 32>(*dp, (const bf16_t*)x, (const bf16_t*)g, w, dw, dbias, (bf16_t*)dx, (float*)workspace, act_alpha, st);
}
