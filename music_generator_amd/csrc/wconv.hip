// Forward 3x3 stride-1 'same' convolution (32 output channels, 32..128 input channels) of the
// discriminator blocks by the Winograd minimal-filtering algorithm F(2x2, 3x3) on
// v_mfma_f32_32x32x2_f32: 16 multiplications per 2x2 output patch and channel pair instead of 36.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// A workgroup of 4 waves (one per SIMD, 16 accumulator tiles each) owns a tile of 16 output rows x
// 32 columns; wave w owns rows 4w..4w+3 = 2 x 16 patches, which are the 32 rows of its MFMA tiles.
// For each of the 16 transform positions (xi, nu) the patch-by-channel matrix V[xi,nu] (32 x 32)
// times the transformed filter U[xi,nu] (32 x 32) accumulates into M[xi,nu]; input channels beyond
// 32 are further accumulation steps in the transform domain.  LDS holds U of the current channel
// slice ([xi nu][k/4][n][4], 64 KiB) and ONE input tile with halo (18 x 34 positions x 32
// channels, 36-float stride, even and odd columns stored apart so that the stride-2 patch reads of
// 16 lanes touch consecutive positions).  V is never stored: a lane builds its fragment from 8
// 16-byte LDS reads per 4 positions (the row combination first, then the 4 column combinations),
// the vector work sits between the MFMAs of 16 independent accumulators.  The next tile travels
// through registers while the current one is computed.  Epilogue: output transform in registers
// (an accumulator register of the 16 tiles is one patch x output channel), bias, tanh(x)+x/alpha,
// optional derivative store.
#include "common.h"
#include "../../include/mgx.h"

namespace {

constexpr int WTH = 16, WTW = 32, WIH = 18, WIW = 34, WPS = 36;
constexpr int W_NPIX = WIH * WIW;                    // 612
constexpr int W_U_FLOATS = 16 * 1024;
constexpr int W_LDS_BYTES = (W_U_FLOATS + W_NPIX * WPS) * 4;      // 153 664
constexpr int W_SITER = (W_NPIX * 8 + 255) / 256;    // float4 staging loads per thread: 20
#ifndef W_LOADB
#define W_LOADB 12
#endif

// column of the input tile -> position slot of its row (even columns first)
__host__ __device__ constexpr int wslot(int c) { return (c & 1) * 17 + (c >> 1); }

template <bool DACT, int S>
__global__ __launch_bounds__(256, 1) void wino3_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ dact, long ntiles, int tiles_h,
                                                       int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ul = lds;                      // [16][8][32][4]
    float* xt = lds + W_U_FLOATS;         // [WIH][34 slots][36]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win;
    // S = cin / 32 channel slices = accumulation steps per tile (unrolled: the 16 accumulator tiles
    // must not be carried around a rolled loop)

    // ---- transformed filter of slice kc: thread (kq, n) owns 4 input channels x 1 output channel ----
    auto u_fill = [&](int kc) {
        int tq = tid;
        asm volatile("" : "+v"(tq));       // keeps the 36 load addresses from being hoisted out of the tile loop
        const int kq = tq >> 5, n = tq & 31;
        const float* wp = w + (long)(kc * 32 + kq * 4) * d.wsk + n;
        float g[9][4];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 4; ++j) g[tap][j] = wp[(long)tap * d.wtap + (long)j * d.wsk];
        f32x4 u[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gg[4][3];               // G g: rows xi
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float g0 = g[b][j], g1 = g[3 + b][j], g2 = g[6 + b][j];
                gg[0][b] = g0;
                gg[1][b] = 0.5f * (g0 + g1 + g2);
                gg[2][b] = 0.5f * (g0 - g1 + g2);
                gg[3][b] = g2;
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                const float a0 = gg[xi][0], a1 = gg[xi][1], a2 = gg[xi][2];
                u[xi * 4 + 0][j] = a0;
                u[xi * 4 + 1][j] = 0.5f * (a0 + a1 + a2);
                u[xi * 4 + 2][j] = 0.5f * (a0 - a1 + a2);
                u[xi * 4 + 3][j] = a2;
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) *(f32x4*)(ul + (p * 8 + kq) * 128 + n * 4) = u[p];
    };

    // ---- input tile staging through registers; out-of-image positions read as zeros ----
    auto tile_of = [&](long t, long& img, int& h0, int& w0) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        const unsigned tu = (unsigned)t;
        const unsigned im = tu / per_img;
        const unsigned rem = tu - im * per_img;
        const unsigned th = rem / (unsigned)tiles_w;
        img = im;
        h0 = (int)th * WTH;
        w0 = (int)(rem - th * tiles_w) * WTW;
    };
    const int c4 = tid & 7;
    const unsigned band_bytes = (unsigned)WIH * W * d.cin * 4u;
    f32x4 sv[W_SITER];
    // The 20 (row, column) pairs of a thread are re-derived at every call from one opaque register:
    // left to itself the compiler hoists 60 loop-invariant offsets out of the tile loop and
    // spills them.
    auto stage_load = [&](long t, int kc) {
        long img; int h0, w0;
        tile_of(t, img, h0, w0);
        const int hb = h0 - 1, wb = w0 - 1;
        const long band = ((img * H + hb) * (long)W) * d.cin;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + band), 0, band_bytes, 0x00020000);
        int p0 = tid >> 3;
        asm volatile("" : "+v"(p0));
        int ty = 0, tx = p0;                               // p0 < 32 < WIW
        const int cbase = (wb * d.cin + kc * 32 + c4 * 4) * 4;
#pragma unroll
        for (int i = 0; i < W_SITER; ++i) {
            const unsigned hi = (unsigned)(ty + hb), wi = (unsigned)(tx + wb);
            const bool ok = ty < WIH && hi < (unsigned)H && wi < (unsigned)W;
            const unsigned off = ok ? (unsigned)((ty * W + tx) * d.cin * 4 + cbase) : 0x80000000u;
            sv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            tx += 32;
            const int wrap = tx >= WIW ? 1 : 0;
            tx -= wrap * WIW;
            ty += wrap;
        }
    };
    auto stage_write = [&]() {
        int p0 = tid >> 3;
        asm volatile("" : "+v"(p0));
        int ty = 0, tx = p0;
#pragma unroll
        for (int i = 0; i < W_SITER; ++i) {
            if (ty < WIH) *(f32x4*)(xt + (ty * WIW + (tx & 1) * 17 + (tx >> 1)) * WPS + c4 * 4) = sv[i];
            tx += 32;
            const int wrap = tx >= WIW ? 1 : 0;
            tx -= wrap * WIW;
            ty += wrap;
        }
    };

    // ---- per-lane constants of the MFMA phase ----
    const int tr = r >> 4, tc = r & 15;
    const float* abase = xt + ((4 * wv + 2 * tr) * WIW + tc) * WPS + 16 * h;
    const float* ubase = ul + (4 * h) * 128 + r * 4;
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    const unsigned row_bytes = (unsigned)W * d.ngemm * 4u;

    long t = blockIdx.x;
    u_fill(0);
    if (t < ntiles) {
        stage_load(t, 0);
        stage_write();
    }
    __syncthreads();
    // The accumulators are defined inside one tile iteration and only ever written by MFMAs inside
    // the step loop: a conditional reset carried around the tile loop makes them pass through 256
    // vector registers at the back edge.
    while (t < ntiles) {
        f32x16 acc[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < S; ++kc) {
        const int kcn = kc + 1 == S ? 0 : kc + 1;
        const long tn = kc + 1 == S ? t + gridDim.x : t;
        const bool has_next = tn < ntiles;
        // ---- 16 blocks (4 channel quads x 4 row combinations) of 16 MFMAs, software-pipelined: while
        //      block b multiplies, the fragment of block b+1 is transformed in four slices of 8
        //      vector instructions (one after every 4 MFMAs), the 8 LDS reads of block b+2 go out as
        //      soon as the row combinations of b+1 have consumed their registers, and the filter
        //      fragment of block b+2 is read when the last MFMA of block b has been issued.
        {
            constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};      // rows of row combination xi
            constexpr float SG[4] = {-1.f, 1.f, -1.f, -1.f};               // sign of the second row
            f32x4 da[4], db[4], uu[2][4], V[2][4], T[4];
            auto rd = [&](int b) {
                const int jq = b >> 2, xi = b & 3;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int so = ((c & 1) * 17 + (c >> 1)) * WPS + 4 * jq;
                    da[c] = *(const f32x4*)(abase + RA[xi] * WIW * WPS + so);
                    db[c] = *(const f32x4*)(abase + RB[xi] * WIW * WPS + so);
                }
            };
            auto ru = [&](int b, int slot) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    uu[slot][c] = *(const f32x4*)(ubase + ((b & 3) * 4 + c) * 1024 + (b >> 2) * 128);
            };
            auto rows = [&](int b, int c0) {                               // row combination of columns c0, c0+1
                T[c0] = da[c0] + SG[b & 3] * db[c0];
                T[c0 + 1] = da[c0 + 1] + SG[b & 3] * db[c0 + 1];
            };
            auto mm = [&](int b, int s) {
                const int xi = b & 3, cs = b & 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[xi * 4 + c] = mfma32(V[cs][c][s], uu[cs][c][s], acc[xi * 4 + c]);
            };
            rd(0);
            ru(0, 0);
            rows(0, 0);
            rows(0, 2);
            V[0][0] = T[0] - T[2], V[0][1] = T[1] + T[2], V[0][2] = T[2] - T[1], V[0][3] = T[1] - T[3];
            rd(1);
            ru(1, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int cn = (b & 1) ^ 1;
                const bool more = b + 1 < 16, more2 = b + 2 < 16;
                if (b == W_LOADB && has_next) stage_load(tn, kcn);
                mm(b, 0);
                if (more) rows(b + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mm(b, 1);
                if (more) rows(b + 1, 2);
                __builtin_amdgcn_sched_barrier(0);
                if (more2) rd(b + 2);
                mm(b, 2);
                if (more) V[cn][0] = T[0] - T[2], V[cn][1] = T[1] + T[2];
                __builtin_amdgcn_sched_barrier(0);
                mm(b, 3);
                if (more) V[cn][2] = T[2] - T[1], V[cn][3] = T[1] - T[3];
                __builtin_amdgcn_sched_barrier(0);
                if (more2) ru(b + 2, b & 1);
            }
        }
        __syncthreads();                                   // every wave is done with the tile and with U
        if (has_next) stage_write();
        if (has_next && S > 1) u_fill(kcn);
        if (kc + 1 < S) __syncthreads();
        }
        {
            // ---- epilogue: output transform, bias, activation, stores ----
            long img; int h0, w0;
            tile_of(t, img, h0, w0);
            const int rows_left = H - h0 < WTH ? H - h0 : WTH;
            if (4 * wv < rows_left) {                      // H is a multiple of 4: a wave is all in or all out
                const long bandoff = ((img * H + h0) * (long)W) * d.ngemm;
                const unsigned bytes = (unsigned)rows_left * row_bytes;
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + bandoff), 0, bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rd =
                    __builtin_amdgcn_make_buffer_rsrc((void*)((DACT ? dact : y) + bandoff), 0, bytes, 0x00020000);
                const float bv = d.nbias > 0 ? bias[r] : 0.f;
                const bool act = d.act_alpha > 0.f;
                const int colmax = W - w0 - 8 * h;         // columns of this lane half left in the row
                const unsigned vbase = (unsigned)((8 * h * d.ngemm + r) * 4);
                const int pstep = d.ngemm * 4;             // bytes per output position
                // output transform on whole accumulator tiles (register s of all 16 tiles is one
                // patch x output channel): Z = A^T M per column nu, then Y = Z A; one output row
                // parity a at a time keeps the live set small
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x16 Y[2];
                    {
                        f32x16 Z[4];
#pragma unroll
                        for (int nu = 0; nu < 4; ++nu) {
                            Z[nu] = a == 0 ? acc[nu] + acc[4 + nu] : acc[4 + nu] - acc[8 + nu];
                            __builtin_amdgcn_sched_barrier(0);
                            Z[nu] = a == 0 ? Z[nu] + acc[8 + nu] : Z[nu] - acc[12 + nu];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        Y[0] = Z[0] + Z[1] + Z[2];
                        Y[1] = Z[1] - Z[2] - Z[3];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int trs = s >> 3;                                     // patch row of register s
                        const int cu = 2 * ((s & 3) + 8 * ((s >> 2) & 1));           // column of its patch (lane half 0)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            float v = Y[b][s] + bv;
                            float der = 0.f;
                            if (act) {
                                float tt, s2;
                                tanh_sech2(v, tt, s2);
                                der = s2 + inv_alpha;
                                v = tt + v * inv_alpha;
                            }
                            const int soff = ((4 * wv + 2 * trs + a) * W + w0 + cu + b) * pstep;
                            const unsigned voff = (cu + b < colmax) ? vbase : 0x80000000u;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, voff, soff, 0);
                            if (DACT && act)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, der), rd, voff, soff, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __syncthreads();
        t += gridDim.x;
    }
}

}  // namespace

// Forward 3x3 'same' convolutions with 32 output channels that the Winograd kernel covers.
bool mgx_wconv_applicable(const mgx_conv_desc& d) {
    if (d.kh != 3 || d.kw != 3 || d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1) return false;
    if (d.pt != 1 || d.pl != 1 || d.hout != d.hin || d.wout != d.win || d.flip || d.accumulate) return false;
    // with a single 32-channel slice the per-tile epilogue outweighs the saved MFMAs: the direct tile
    // kernel is as fast there
    if (d.hin % 4 || d.cin % 32 || d.cin < 64 || d.cin > 128 || d.ngemm != 32 || d.cout != 32) return false;
    if (d.wsn != 1 || d.wsk != 32 || d.wtap != (int64_t)d.cin * 32) return false;
    if (d.nbias != 0 && d.nbias != 32) return false;
    if ((long)WIH * d.win * d.cin * 4 >= (1L << 31) || (long)WTH * d.win * d.ngemm * 4 >= (1L << 31)) return false;
    const long ntiles = (long)d.nimg * ((d.hin + WTH - 1) / WTH) * ((d.win + WTW - 1) / WTW);
    return ntiles < (1L << 31);
}

int mgx_wconv_launch(const mgx_conv_desc& d, const float* x, const float* w, const float* bias, float* y, float* dact,
                     hipStream_t st) {
    const int tiles_h = (d.hin + WTH - 1) / WTH, tiles_w = (d.win + WTW - 1) / WTW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
#define MGX_W3(DD, SS)                                                                                              \
    do {                                                                                                            \
        static bool allowed = false;                                                                                \
        if (!allowed) {                                                                                             \
            (void)hipFuncSetAttribute((const void*)wino3_kernel<DD, SS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      W_LDS_BYTES);                                                                 \
            allowed = true;                                                                                         \
        }                                                                                                           \
        hipLaunchKernelGGL((wino3_kernel<DD, SS>), dim3(grid), dim3(256), W_LDS_BYTES, st, d, x, w, bias, y, dact,   \
                           ntiles, tiles_h, tiles_w);                                                               \
    } while (0)
    const int S = d.cin / 32;
    if (dact) { if (S == 2) MGX_W3(true, 2); else if (S == 3) MGX_W3(true, 3); else MGX_W3(true, 4); }
    else { if (S == 2) MGX_W3(false, 2); else if (S == 3) MGX_W3(false, 3); else MGX_W3(false, 4); }
#undef MGX_W3
    return mgx_launch_status();
}
