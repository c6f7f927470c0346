// Forward 3x3 stride-1 'same' convolution (32 output channels, 32..128 input channels) of the
// discriminator blocks by the Winograd minimal-filtering algorithm F(2x2, 3x3) on
// v_mfma_f32_32x32x2_f32: 16 multiplications per 2x2 output patch and channel pair instead of 36.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A       d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// A workgroup of 4 waves (one per SIMD, 16 accumulator tiles each) owns a tile of 16 output rows x
// 32 columns; wave w owns rows 4w..4w+3 = 2 x 16 patches, which are the 32 rows of its MFMA tiles.
// For each of the 16 transform positions (xi, nu) the patch-by-channel matrix V[xi,nu] (32 x k)
// times the transformed filter U[xi,nu] (k x 32) accumulates into M[xi,nu].  The input channels are
// consumed in slices of 16: one accumulation step = 8 blocks of 16 MFMAs.
//
// On this chip every vector-ALU and vector-memory instruction takes issue cycles away from the fp32
// MFMAs of its SIMD (measured: a packed add ~8 cycles, a 16-byte buffer load ~70, an LDS store ~40
// of a 64-cycle MFMA slot), so the kernel is built to execute as few of them as possible:
//  * the input tile of the NEXT step (18 x 34 positions x 16 channels) is loaded while the current
//    step multiplies and written to the second of two LDS tile buffers in the same phase: 15
//    16-byte loads and 15 LDS stores per wave, one per group of 4 MFMAs, all addresses a per-lane
//    constant plus scalars, the halo outside the image zero-filled by the buffer range check.
//    (Loads with the LDS as destination need no registers but stall the issuing wave ~120 cycles
//    each.)  A position is 64 bytes; its four 16-byte channel quads are stored XOR-swizzled with
//    bits 2..3 of the position slot, the even and odd columns of a row apart, so that the
//    stride-2 patch reads of a 16-lane group cover all 64 banks;
//  * V is never stored: a lane builds its fragment from 8 16-byte LDS reads per 4 channels with 16
//    packed (two floats per lane) additions, slotted between the MFMAs of the previous block;
//  * the MFMAs are asm statements that pin the 16 accumulator tiles to the accumulation registers
//    and the epilogue copies one tile at a time out of them: the compiler's own placement puts all
//    256 copies behind the last MFMA and spills everything else.
// U of a 32-channel convolution stays in LDS for the whole kernel; wider inputs stream it one
// slice ahead (18 dword loads and 35 packed operations per thread and step).  Epilogue: output
// transform (an accumulator register of the 16 tiles is one patch x output channel), bias,
// tanh(x)+x/alpha on register pairs, optional derivative store.
#include "common.h"
#include "wino.h"
#include "../../include/mgx.h"

namespace {


constexpr int WTH = 16, WTW = 32, WIH = 18;
constexpr int W_RS = 40;                              // position slots per tile row (34 used)
constexpr int W_TILE = WIH * W_RS * 16;               // floats per tile buffer (16 channels per slot)
constexpr int W_USLICE = 16 * 4 * 128;                // floats of U per 16-channel slice: [xi nu][k/4][n][4]
constexpr int W_LDS_BYTES = (2 * W_USLICE + 2 * W_TILE) * 4;      // 157 696

// column of the input tile -> position slot of its row (even columns first)
__host__ __device__ constexpr int wslot(int c) { return (c & 1) * 17 + (c >> 1); }

struct WTile {
    int img, th, tw;
};

// MODE 0: linear output, 1: tanh(x)+x/alpha, 2: the same and the derivative stored to dact.
// S16 = cin / 16 accumulation steps per tile (unrolled: the 16 accumulator tiles must not be
// carried around a rolled loop; S16 is even, so the buffer of a step is a compile-time constant).
// MUL: the input is x * xmul elementwise (data-gradients: dy times the saved activation derivative).
// OSTR: output channels = floats between output positions.  With more than 32 the workgroup computes
// them 32 at a time from the one staged input tile (both 16-channel slices stay in their LDS
// buffers), the filter streaming through LDS one (channel block, slice) ahead.
template <int MODE, int S16, bool MUL, int OSTR>
__global__ __launch_bounds__(256, 1) void wino3_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                       const float* __restrict__ xmul,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ dact, long ntiles, int tiles_h,
                                                       int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ul = lds;                      // [2][16][4][32][4]
    float* xt = lds + 2 * W_USLICE;       // [2][WIH][W_RS][16]
    constexpr int NB = OSTR / 32;         // output channel blocks
    constexpr bool kResidentU = S16 == 2 && NB == 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win, cin = d.cin;

    // ---- transformed filter of a 16-channel slice: thread (kq, n, half) owns 2 input channels x 1
    //      output channel; the 18 addresses are one per-thread offset plus a scalar ----
    //      Weight of (tap, input channel k, output channel n) at w[tap' * wtap + k * wsk + n * wsn],
    //      tap' = 8 - tap for a flipped (data-gradient) filter.
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (unsigned)(9 * d.wtap * 4), 0x00020000);
    const int ukq = tid >> 6, un = (tid >> 1) & 31, uhalf = tid & 1;
    const unsigned uvoff = (unsigned)(((ukq * 4 + 2 * uhalf) * d.wsk + un * d.wsn) * 4);
    const unsigned uvoff1 = uvoff + (unsigned)(d.wsk * 4);
    float* const ustore = ul + ukq * 128 + un * 4 + 2 * uhalf;
    // filter slice u = output channel block u / S16, input channel slice u % S16
    auto u_load_tap = [&](int slice, int tap, f32x2 (&g)[9]) {
        const int soff = (((d.flip ? 8 - tap : tap) * (int)d.wtap) + (slice % S16) * 16 * d.wsk +
                          (slice / S16) * 32 * d.wsn) * 4;
        g[tap][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, uvoff, soff, 0));
        g[tap][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, uvoff1, soff, 0));
    };
    auto u_load = [&](int slice, f32x2 (&g)[9]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) u_load_tap(slice, tap, g);
    };
    auto u_store = [&](int ubuf, const f32x2 (&g)[9]) {
        const f32x2 half2 = {0.5f, 0.5f};
        f32x2 gg[4][3];                    // G g: rows xi
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const f32x2 t = pk_add(g[b], g[6 + b]);
            gg[0][b] = g[b];
            gg[1][b] = pk_mul(pk_add(t, g[3 + b]), half2);
            gg[2][b] = pk_mul(pk_sub(t, g[3 + b]), half2);
            gg[3][b] = g[6 + b];
        }
        float* up = ustore + ubuf * W_USLICE;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const f32x2 t = pk_add(gg[xi][0], gg[xi][2]);
            *(f32x2*)(up + (xi * 4 + 0) * 512) = gg[xi][0];
            *(f32x2*)(up + (xi * 4 + 1) * 512) = pk_mul(pk_add(t, gg[xi][1]), half2);
            *(f32x2*)(up + (xi * 4 + 2) * 512) = pk_mul(pk_sub(t, gg[xi][1]), half2);
            *(f32x2*)(up + (xi * 4 + 3) * 512) = gg[xi][2];
        }
    };

    // ---- tile order, in scalar registers.  Workgroup b runs on XCD b % 8 (each XCD has its own L2) and the
    //      18 x 34 halo tile of a 16 x 32 output tile overlaps its neighbours': within every full round of 256
    //      tiles the 32 workgroups of an XCD take 32 CONSECUTIVE tiles, and consecutive tiles are vertical
    //      neighbours first (column-major inside an image), so that halo rows and columns are found in the
    //      XCD's L2 instead of being fetched from HBM again (measured before: 1.43x the input bytes fetched) ----
    auto tile_at = [&](long t_) {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        unsigned tu = (unsigned)t_;
        if (gridDim.x == 256) {
            const unsigned round = tu >> 8, b = tu & 255u;
            if ((long)(round + 1) * 256 <= ntiles) tu = (round << 8) + (b & 7u) * 32u + (b >> 3);
        }
        WTile w_;
        w_.img = (int)(tu / per_img);
        const unsigned rem = tu - (unsigned)w_.img * per_img;
        w_.tw = (int)(rem / (unsigned)tiles_h);
        w_.th = (int)(rem - (unsigned)w_.tw * tiles_h);
        return w_;
    };
    WTile cur = tile_at(blockIdx.x);

    // ---- input tile global -> registers -> LDS.  Wave w moves tile rows w, w+4, ...; a row is three
    //      instructions of 16 position slots (the third half-masked).  Lane l of an instruction
    //      writes chunk l&3 of slot l>>2 and therefore fetches channel quad (l&3) ^ (slot>>2 & 3)
    //      of the column that slot stands for. ----
    // range of the tile's buffer resource, counted from the tile origin (row hb, column wb): the scalar
    // row offset takes part in the range check, and the last row's positions lie up to 34 columns
    // behind 18 whole rows (positions outside the image are masked explicitly, not by this range)
    const unsigned band_bytes = (unsigned)(WIH * W + WTW + 2) * cin * 4u;
    const int row_bytes_in = W * cin * 4;
    // stage_prepare sets up one tile slice (resources, per-lane source offsets); stage_load(j) and
    // stage_store(j) are its j-th load and LDS store: issued back to back they stall the wave on the
    // memory pipeline, so the main loop hands them out one per group of 4 MFMAs.
    __amdgpu_buffer_rsrc_t ds_rx, ds_rz, ds_mx, ds_mz;
    unsigned ds_vo[3];
    int ds_hb = 0, ds_buf = 0;
    bool ds_on = false;
    auto stage_prepare = [&](WTile t, int k16, int buf, bool on) {
        const int hb = t.th * WTH - 1, wb = t.tw * WTW - 1;
        const float* origin = x + (((long)t.img * H + hb) * (long)W + wb) * cin + k16 * 16;
        ds_rx = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, band_bytes, 0x00020000);
        ds_rz = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, 0, 0x00020000);
        if (MUL) {
            const float* morigin = xmul + (origin - x);
            ds_mx = __builtin_amdgcn_make_buffer_rsrc((void*)morigin, 0, band_bytes, 0x00020000);
            ds_mz = __builtin_amdgcn_make_buffer_rsrc((void*)morigin, 0, 0, 0x00020000);
        }
        ds_hb = hb;
        ds_buf = buf;
        ds_on = on;
        // The per-lane source offsets are re-derived for every slice from one opaque register: kept
        // live across the tile loop they are the first values the register allocator spills.
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int sl = 16 * k + ((ln >> 2) & 15);
            const int c = k == 0 ? 2 * sl : (sl < 17 ? 2 * sl : 2 * sl - 33);
            const int q = (ln & 3) ^ ((sl >> 2) & 3);
            const bool ok = (k < 2 || sl < 34) && (unsigned)(c + wb) < (unsigned)W;
            ds_vo[k] = ok ? (unsigned)((c * cin + q * 4) * 4) : 0x80000000u;
        }
    };
    // Instruction j moves row wv + 4 (j / 3), slot group j % 3, through sv[j].  (Loads with the LDS as
    // destination would need no registers, but each one stalls the issuing wave for ~120 cycles.)
    f32x4 sv[15], sm[MUL ? 15 : 1];
    float* const xs_lane = xt + wv * (W_RS * 16) + lane * 4;
    auto stage_load = [&](int j) {
        const int row = wv + 4 * (j / 3), k = j % 3;
        if (ds_on && row < WIH) {
            const bool row_ok = (unsigned)(ds_hb + row) < (unsigned)H;
            const __amdgpu_buffer_rsrc_t rs = row_ok ? ds_rx : ds_rz;
            sv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ds_vo[k], row * row_bytes_in, 0));
            if (MUL)
                sm[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(row_ok ? ds_mx : ds_mz, ds_vo[k],
                                                                                         row * row_bytes_in, 0));
        }
    };
    auto stage_store = [&](int j) {
        const int i = j / 3, k = j % 3;
        if (ds_on && wv + 4 * i < WIH) {
            float* dst = xs_lane + ds_buf * W_TILE + i * (4 * W_RS * 16) + k * 256;
            if (k < 2 || lane < 32) *(f32x4*)dst = MUL ? sv[j] * sm[j] : sv[j];
        }
    };

    // ---- per-lane constants of the MFMA phase ----
    const int tr = r >> 4, tc = r & 15;
    const float* aptr[2][4];              // [channel block jb][patch column c]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int sl = wslot(c) + tc;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
            aptr[jb][c] = xt + ((4 * wv + 2 * tr) * W_RS + sl) * 16 + (((2 * jb + h) ^ ((sl >> 2) & 3)) * 4);
    }
    const float* ubase = ul + h * 128 + r * 4;
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    // the lane's bias: with one output channel block it does not depend on the tile (a load inside the epilogue
    // has its whole latency exposed: one wave per SIMD, nothing else to run)
    const float bias_lane = (NB == 1 && d.nbias > 0) ? bias[r] : 0.f;
    const unsigned out_row_bytes = (unsigned)W * (unsigned)OSTR * 4u;

    long t = blockIdx.x;
    {
        f32x2 g[9];
        u_load(0, g);
        stage_prepare(cur, 0, 0, t < ntiles);
#pragma unroll
        for (int j = 0; j < 15; ++j) stage_load(j);
#pragma unroll
        for (int j = 0; j < 15; ++j) stage_store(j);
        u_store(0, g);
        if (kResidentU) {
            u_load(1, g);
            u_store(1, g);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    while (t < ntiles) {
        const WTile nxt = tile_at(t + gridDim.x < ntiles ? t + gridDim.x : t);
        const bool next_tile = t + gridDim.x < ntiles;
        // The MFMAs are asm statements with the accumulator tied to the accumulation registers: the
        // compiler's own choice moves a 256-register accumulator set into vector registers as soon
        // as a loop carries it or the unrolled code grows.  The first MFMA of a tile takes 0 as C.
        f32x16 acc[16];
        auto step = [&](const int k16, const int k2, const bool first, const bool last, const int nb)
                        __attribute__((always_inline)) {
            const int buf = k2, ubuf = k2;
            const int kn = NB == 1 ? (last ? 0 : k16 + 1) : (nb * S16 + k16 + 1) % (NB * S16);   // next filter slice
            // the tile and filter slice of the next step go into the buffers the previous step used
            // with several output blocks the input tile is staged during the first block only, the
            // next tile's first slice during the last step of the last block
            stage_prepare(last ? nxt : cur, last ? 0 : k16 + 1, buf ^ 1,
                          NB == 1 ? (!last || next_tile) : (last ? (nb == NB - 1 && next_tile) : nb == 0));
            f32x2 g[9];
            // ---- 8 blocks (2 channel blocks x 4 row combinations) of 16 MFMAs, software-pipelined:
            //      while block b multiplies, the fragment of block b+1 is transformed in four slices
            //      of 4 packed instructions (one after every 4 MFMAs), the 8 LDS reads of block b+2
            //      go out as soon as the row combinations of b+1 have consumed their registers, and
            //      the filter fragment of block b+2 is read when the last MFMA of block b is issued.
            {
                constexpr int RA[4] = {0, 1, 2, 1}, RB[4] = {2, 2, 1, 3};      // rows of row combination xi
                constexpr bool PLUS[4] = {false, true, false, false};         // sign of the second row
                f32x4 da[4], db[4], uu[2][4];
                f32x2 V[2][4][2], T[4][2];
                auto rd = [&](int b) {
                    const int jb = b >> 2, xi = b & 3;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        da[c] = *(const f32x4*)(aptr[jb][c] + buf * W_TILE + RA[xi] * (W_RS * 16));
                        db[c] = *(const f32x4*)(aptr[jb][c] + buf * W_TILE + RB[xi] * (W_RS * 16));
                    }
                };
                auto ru = [&](int b, int slot) {
                    const int jb = b >> 2, xi = b & 3;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        uu[slot][c] = *(const f32x4*)(ubase + ubuf * W_USLICE + ((xi * 4 + c) * 4 + 2 * jb) * 128);
                };
                auto rows = [&](int b, int c0) {                           // row combination of columns c0, c0+1
#pragma unroll
                    for (int c = c0; c < c0 + 2; ++c) {
                        T[c][0] = PLUS[b & 3] ? pk_add(lo2(da[c]), lo2(db[c])) : pk_sub(lo2(da[c]), lo2(db[c]));
                        T[c][1] = PLUS[b & 3] ? pk_add(hi2(da[c]), hi2(db[c])) : pk_sub(hi2(da[c]), hi2(db[c]));
                    }
                };
                auto cols = [&](int slot, int half) {                      // column combinations 2*half, 2*half+1
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (half == 0) {
                            V[slot][0][q] = pk_sub(T[0][q], T[2][q]);
                            V[slot][1][q] = pk_add(T[1][q], T[2][q]);
                        } else {
                            V[slot][2][q] = pk_sub(T[2][q], T[1][q]);
                            V[slot][3][q] = pk_sub(T[1][q], T[3][q]);
                        }
                    }
                };
                auto mm = [&](int b, int s) {
                    const int xi = b & 3, cs = b & 1;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (first && b < 4 && s == 0) mfma_first(acc[xi * 4 + c], V[cs][c][0][0], uu[cs][c][0]);
                        else mfma_acc(acc[xi * 4 + c], V[cs][c][s >> 1][s & 1], uu[cs][c][s]);
                };
                rd(0);
                ru(0, 0);
                rows(0, 0);
                rows(0, 2);
                cols(0, 0);
                cols(0, 1);
                rd(1);
                ru(1, 1);
                __builtin_amdgcn_sched_barrier(0);
                // memory instruction slot j = 0..31 (one per group of 4 MFMAs): the 15 tile loads of
                // the next step first, then the 9 filter taps of the next slice; from slot 17 on the
                // tile registers go to the LDS buffer the previous step read
                auto vmem = [&](int j) {
                    if (j < 15) stage_load(j);
                    else if (!kResidentU && j < 24) u_load_tap(kn, j - 15, g);
                    if (j >= 17) stage_store(j - 17);
                };
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int cn = (b & 1) ^ 1;
                    const bool more = b + 1 < 8, more2 = b + 2 < 8;
                    mm(b, 0);
                    if (more) rows(b + 1, 0);
                    vmem(4 * b);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(b, 1);
                    if (more) rows(b + 1, 2);
                    vmem(4 * b + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more2) rd(b + 2);
                    mm(b, 2);
                    if (more) cols(cn, 0);
                    vmem(4 * b + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(b, 3);
                    if (more) cols(cn, 1);
                    vmem(4 * b + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more2) ru(b + 2, b & 1);
                    if (!kResidentU && b == 6) {
                        u_store(ubuf ^ 1, g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (last) {
                // ---- epilogue: output transform, bias, activation, stores.  One buffer resource
                //      per output row clips the columns beyond the image by its range check. ----
                const int h0 = cur.th * WTH, w0 = cur.tw * WTW;
                // the compiler does not see the MFMAs: keep the reads of their results clear of the
                // last one by more than its 16 passes
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
                if (h0 + 4 * wv < H) {                     // H is a multiple of 4: a wave is all in or all out
                    const unsigned vbase = (unsigned)(((w0 + 8 * h) * OSTR + r) * 4);
                    const float bv = NB == 1 ? bias_lane : (d.nbias > 0 ? bias[nb * 32 + r] : 0.f);
                    const f32x2 bv2 = {bv, bv}, ia2 = {inv_alpha, inv_alpha};
                    // Output transform Y[a][b] = sum over (xi, nu) of At[a][xi] At[b][nu] M[xi][nu] with
                    // At = [1 1 1 0; 0 1 -1 -1]: every accumulator tile is copied out of the accumulation
                    // registers once and added to or subtracted from the (at most four) outputs it
                    // belongs to.  36 tile operations instead of the 24 of the two-stage form, but only
                    // the four outputs and one tile are live: the two-stage form runs out of registers.
                    f32x16 Y[2][2];
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
                        for (int nu = 0; nu < 4; ++nu) {
                            const f32x16 m = acc_read16(acc[xi * 4 + nu]);
#pragma unroll
                            for (int a = 0; a < 2; ++a) {
                                const int ca = a == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1));
#pragma unroll
                                for (int b = 0; b < 2; ++b) {
                                    const int cb = b == 0 ? (nu < 3 ? 1 : 0) : (nu == 0 ? 0 : (nu == 1 ? 1 : -1));
                                    if (ca * cb == 0) continue;
                                    if (xi == a && nu == b) Y[a][b] = m;           // first contribution, always +
                                    else Y[a][b] = ca * cb > 0 ? add16(Y[a][b], m) : sub16(Y[a][b], m);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
#pragma unroll
                        for (int trs = 0; trs < 2; ++trs) {                         // patch row of registers 8 trs ..
                            const long rowoff = (((long)cur.img * H + h0 + 4 * wv + 2 * trs + a) * (long)W) * OSTR + nb * 32;
                            const __amdgpu_buffer_rsrc_t ry =
                                __builtin_amdgcn_make_buffer_rsrc((void*)(y + rowoff), 0, out_row_bytes, 0x00020000);
                            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                                (void*)((MODE == 2 ? dact : y) + rowoff), 0, out_row_bytes, 0x00020000);
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
#pragma unroll
                                for (int i = 4 * trs; i < 4 * trs + 4; ++i) {       // accumulator registers 2i, 2i+1
                                    f32x2 v = pk_add(pair16(Y[a][b], i), bv2);
                                    f32x2 der = {0.f, 0.f};
                                    if (MODE >= 1) {
                                        // tanh and sech^2 of both values: e = exp(-2|v|), tanh = sign (1-e)/(1+e),
                                        // sech^2 = 4e/(1+e)^2 (hardware exp2 and reciprocal, as tanh_sech2)
                                        const f32x2 c2 = {2.885390081777927f, 2.885390081777927f};
                                        const f32x2 one = {1.f, 1.f}, four = {4.f, 4.f};
                                        const f32x2 tt = pk_mul(v, c2);
                                        f32x2 e;
                                        e[0] = __builtin_amdgcn_exp2f(-fabsf(tt[0]));
                                        e[1] = __builtin_amdgcn_exp2f(-fabsf(tt[1]));
                                        trans_settle(e);
                                        const f32x2 den = pk_add(one, e);
                                        f32x2 inv;
                                        inv[0] = __builtin_amdgcn_rcpf(den[0]);
                                        inv[1] = __builtin_amdgcn_rcpf(den[1]);
                                        trans_settle(inv);
                                        f32x2 mg = pk_mul(pk_sub(one, e), inv);
                                        mg[0] = copysignf(mg[0], v[0]);
                                        mg[1] = copysignf(mg[1], v[1]);
                                        if (MODE == 2) der = pk_fma(pk_mul(pk_mul(four, e), inv), inv, ia2);
                                        v = pk_fma(v, ia2, mg);
                                    }
#pragma unroll
                                    for (int e2 = 0; e2 < 2; ++e2) {
                                        const int s = 2 * i + e2;
                                        const int cu = 2 * ((s & 3) + 8 * ((s >> 2) & 1)) + b;   // column (lane half 0)
                                        // (a bit_cast applied directly to a vector element reads element 0)
                                        const float vs = v[e2], ds = der[e2];
                                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vs), ry,
                                                                              vbase + cu * (OSTR * 4), 0, 0);
                                        if (MODE == 2)
                                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ds), rd,
                                                                                  vbase + cu * (OSTR * 4), 0, 0);
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (NB > 1) {
#pragma unroll 1
            for (int nb = 0; nb < NB; ++nb) {
                step(0, 0, true, false, nb);
                step(1, 1, false, true, nb);
            }
        } else if (S16 <= 4) {
#pragma unroll
            for (int k16 = 0; k16 < S16; ++k16) step(k16, k16 & 1, k16 == 0, k16 + 1 == S16, 0);
        } else {
            // more than two step pairs: the first pair apart (it starts the accumulators), the rest
            // a loop, which also keeps the code inside the instruction cache
            step(0, 0, true, false, 0);
            step(1, 1, false, false, 0);
#pragma unroll 1
            for (int kk = 2; kk < S16; kk += 2) {
                step(kk, 0, false, false, 0);
                step(kk + 1, 1, false, kk + 2 == S16, 0);
            }
        }
        t += gridDim.x;
        cur = nxt;
    }
}

}  // namespace
// 3x3 'same' convolutions that the Winograd kernel covers: 32 output channels in the forward layout
// (weights [tap][k][n]) or the data-gradient layout ([tap][n][k], flipped taps); with an input
// multiplier (the data-gradients of the discriminator blocks: 32 input channels, no activation) also
// 128 output channels, computed 32 at a time from one staged input tile.
bool mgx_wconv_applicable(const mgx_conv_desc& d, bool mul) {
    if (d.kh != 3 || d.kw != 3 || d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1) return false;
    if (d.pt != 1 || d.pl != 1 || d.hout != d.hin || d.wout != d.win || d.accumulate) return false;
    if (d.hin % 4 || d.hin < 8 || d.cin % 32 || d.cin < 32 || d.cin > 128 || d.cout != d.ngemm) return false;
    // 8 rows fill half of a 16-row tile: with 32 input channels the direct 8-row tile kernel is faster there
    if (d.hin < 16 && d.cin == 32) return false;
    if (mul) {
        if (d.cin != 32 || d.act_alpha > 0.f || (d.ngemm != 32 && d.ngemm != 128)) return false;
    } else if (d.ngemm != 32) return false;
    const bool fwd_layout = d.wsn == 1 && d.wsk == d.ngemm, bwd_layout = d.wsk == 1 && d.wsn == d.cin;
    if (!(fwd_layout || bwd_layout) || d.wtap != (int64_t)d.cin * d.ngemm) return false;
    if (d.nbias != 0 && d.nbias != d.ngemm) return false;
    if (((long)WIH * d.win + WTW + 2) * d.cin * 4 >= (1L << 31) || (long)d.win * d.ngemm * 4 >= (1L << 31)) return false;
    const long ntiles = (long)d.nimg * ((d.hin + WTH - 1) / WTH) * ((d.win + WTW - 1) / WTW);
    return ntiles < (1L << 31);
}

int mgx_wconv_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                     float* y, float* dact, hipStream_t st) {
    const int tiles_h = (d.hin + WTH - 1) / WTH, tiles_w = (d.win + WTW - 1) / WTW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
#define MGX_W3(MM, SS, UU, OO, Y, WP, BP)                                                                           \
    do {                                                                                                            \
        static std::atomic<uint64_t> allowed{0};                                                                    \
        mgx_allow_dynamic_lds(wino3_kernel<MM, SS, UU, OO>, W_LDS_BYTES, allowed);                                  \
        hipLaunchKernelGGL((wino3_kernel<MM, SS, UU, OO>), dim3(grid), dim3(256), W_LDS_BYTES, st, d, x, xmul, WP,  \
                           BP, Y, dact, ntiles, tiles_h, tiles_w);                                                  \
    } while (0)
#define MGX_W3S(MM)                                                                                                 \
    do {                                                                                                            \
        if (S16 == 2) MGX_W3(MM, 2, false, 32, y, w, bias);                                                         \
        else if (S16 == 4) MGX_W3(MM, 4, false, 32, y, w, bias);                                                    \
        else if (S16 == 6) MGX_W3(MM, 6, false, 32, y, w, bias);                                                    \
        else MGX_W3(MM, 8, false, 32, y, w, bias);                                                                  \
    } while (0)
    const int S16 = d.cin / 16;
    const bool act = d.act_alpha > 0.f;
    if (xmul && d.ngemm == 128) MGX_W3(0, 2, true, 128, y, w, bias);
    else if (xmul) MGX_W3(0, 2, true, 32, y, w, bias);
    else if (act && dact) MGX_W3S(2);
    else if (act) MGX_W3S(1);
    else MGX_W3S(0);
#undef MGX_W3S
#undef MGX_W3
    return mgx_launch_status();
}
