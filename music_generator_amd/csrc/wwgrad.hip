// Weight gradient of the 3x3 stride-1 'same' convolutions with 32 output channels by the Winograd
// minimal-filtering algorithm F(2x2, 3x3), the transpose of the forward kernel in wconv.hip:
//
//   dU[xi,nu] = sum over 2x2 output patches of (B^T d B)[xi,nu]^T (A dY A^T)[xi,nu],   dg = G^T dU G
//
// d: 4x4 input patch (32 input channels), dY: 2x2 patch of the output gradient (32 output channels,
// already multiplied by the saved activation derivative when MUL).  Per pair of patches 16 MFMAs
// v_mfma_f32_32x32x2_f32 (one per transform position, contraction index = patch) instead of the
// 36 of the direct form (9 taps x 4 positions).
//
// A workgroup of 4 waves (one per SIMD) walks tiles of 8 x 32 output positions; wave w owns patch
// row w (16 patches = 8 MFMA steps).  LDS holds two buffers of the input window with halo (10 x 34
// positions) and the output-gradient tile, both as [position][32 channels]: lane (r, h) reads
// channel r of the positions of patch 2s+h with conflict-free 4-byte reads, transforms them in
// registers (two patches at a time in the two halves of packed instructions) and feeds the MFMAs;
// the 16 accumulator tiles (input channel x output channel per transform position) stay in the
// accumulation registers for the whole launch.  The next tile is loaded into registers before the
// MFMA phase and written to the other LDS buffer after it.  A launch handles one 32-channel slice
// of the input per blockIdx.y.  Each wave leaves its 16 tiles as a slab; the slabs are summed in
// fixed order and dg = G^T dU G is formed by the kernels at the end of this file (bitwise
// reproducible, no float atomics).  The factors -1 of row / column 3 of A are applied there too.
#include "common.h"
#include "wino.h"
#include "../../include/mgx.h"

namespace {

constexpr int GTH = 8, GTW = 32, GIH = 10, GIW = 34;
constexpr int G_X = GIH * GIW * 32;                  // floats of the input window
constexpr int G_G = GTH * GTW * 32;                  // floats of the output-gradient tile
constexpr int G_BUF = G_X + G_G;
constexpr int G_LDS_BYTES = 2 * G_BUF * 4;           // 152 576

struct GTile {
    int img, th, tw;
};

template <bool MUL>
__global__ __launch_bounds__(256, 1) void wino3_wgrad_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ dymul,
                                                             float* __restrict__ partial, float* __restrict__ bpartial,
                                                             long ntiles, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int H = d.hin, W = d.win, cin = d.cin;
    // The cin/32 input chunks of a tile share its output-gradient tile.  Workgroups go to the 8 XCDs
    // round-robin by linear id: re-map the ids so that the chunks of one tile index run on one XCD
    // and find that tile in its L2.  Slabs are addressed by (chunk, split): the sums do not change.
    unsigned bx = blockIdx.x;
    int kc = blockIdx.y;
    const int kchunks = cin / 32;
    if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, q = lin >> 3;
        kc = (int)(q % gridDim.y);
        bx = (q / gridDim.y) * 8 + (lin & 7);
    }
    const bool do_bias = bpartial != nullptr && kc == 0;

    // ---- tile walk in scalar registers ----
    GTile dstep, cur;
    {
        const unsigned per_img = (unsigned)(tiles_h * tiles_w);
        const unsigned g = gridDim.x, b = bx;
        dstep.img = (int)(g / per_img);
        unsigned rem = g - (unsigned)dstep.img * per_img;
        dstep.th = (int)(rem / (unsigned)tiles_w);
        dstep.tw = (int)(rem - (unsigned)dstep.th * tiles_w);
        cur.img = (int)(b / per_img);
        rem = b - (unsigned)cur.img * per_img;
        cur.th = (int)(rem / (unsigned)tiles_w);
        cur.tw = (int)(rem - (unsigned)cur.th * tiles_w);
    }
    auto advance = [&](GTile t) {
        t.tw += dstep.tw;
        if (t.tw >= tiles_w) t.tw -= tiles_w, ++t.th;
        t.th += dstep.th;
        if (t.th >= tiles_h) t.th -= tiles_h, ++t.img;
        t.img += dstep.img;
        return t;
    };

    // ---- staging: thread (column tid/8, channel quad tid%8) moves its quad of every row of the
    //      window and of the gradient tile; the two window columns beyond 32 are one more load of the
    //      first 160 threads.  Out-of-image positions read as zeros (buffer range check / empty buffer).
    const int c4 = tid & 7, col = tid >> 3;
    const int rrow = tid >> 4, rcol = 32 + ((tid >> 3) & 1), rq = tid & 7;        // residual position (tid < 160)
    f32x4 sx[GIH + 1], sg[GTH], sm[MUL ? GTH : 1];
    auto stage_load = [&](GTile t) {
        const int h0 = t.th * GTH, w0 = t.tw * GTW, hb = h0 - 1, wb = w0 - 1;
        const float* xo = x + (((long)t.img * H + hb) * (long)W + wb) * cin + kc * 32;
        const unsigned xrange = (unsigned)((GIH * W + GIW) * cin * 4);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xo, 0, xrange, 0x00020000);
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)xo, 0, 0, 0x00020000);
        const unsigned vmain = (unsigned)(col + wb) < (unsigned)W ? (unsigned)((col * cin + c4 * 4) * 4) : 0x80000000u;
        const int xrow_bytes = W * cin * 4;
#pragma unroll
        for (int i = 0; i < GIH; ++i) {
            const bool row_ok = (unsigned)(hb + i) < (unsigned)H;                 // uniform
            sx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(row_ok ? rx : rz, vmain, i * xrow_bytes, 0));
        }
        const bool rok = tid < 160 && (unsigned)(hb + rrow) < (unsigned)H && (unsigned)(rcol + wb) < (unsigned)W;
        const unsigned vres = rok ? (unsigned)(((rrow * W + rcol) * cin + rq * 4) * 4) : 0x80000000u;
        sx[GIH] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vres, 0, 0));
        // output gradient: the rows of a tile are always inside the image (H is a multiple of 8)
        const long go = (((long)t.img * H + h0) * (long)W + w0) * 32;
        const unsigned grange = (unsigned)((GTH * W + GTW) * 32 * 4);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + go), 0, grange, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm =
            __builtin_amdgcn_make_buffer_rsrc((void*)((MUL ? dymul : dy) + go), 0, grange, 0x00020000);
        const unsigned vg = col + w0 < W ? (unsigned)((col * 32 + c4 * 4) * 4) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < GTH; ++i) {
            sg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, vg, i * W * 128, 0));
            if (MUL) sm[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, vg, i * W * 128, 0));
        }
    };
    auto stage_store = [&](int buf) {
        float* xb = lds + buf * G_BUF;
        float* gb = xb + G_X;
#pragma unroll
        for (int i = 0; i < GIH; ++i) *(f32x4*)(xb + (i * GIW + col) * 32 + c4 * 4) = sx[i];
        if (tid < 160) *(f32x4*)(xb + (rrow * GIW + rcol) * 32 + rq * 4) = sx[GIH];
#pragma unroll
        for (int i = 0; i < GTH; ++i) *(f32x4*)(gb + (i * GTW + col) * 32 + c4 * 4) = MUL ? sg[i] * sm[i] : sg[i];
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
    f32x2 bs2 = {0.f, 0.f};

    long t = bx;
    if (t < ntiles) {
        stage_load(cur);
        stage_store(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // per-lane read bases: window rows 2w.., columns 2h..; gradient rows 2w.., columns 2h..
    const float* xlane = lds + ((2 * wv) * GIW + 2 * h) * 32 + r;
    const float* glane = lds + G_X + ((2 * wv) * GTW + 2 * h) * 32 + r;

    int buf = 0;
    while (t < ntiles) {
        const GTile nxt = advance(cur);
        const bool more_tiles = t + gridDim.x < ntiles;
        if (more_tiles) stage_load(nxt);
        const float* xp = xlane + buf * G_BUF;
        const float* gp = glane + buf * G_BUF;
        // element 0 / 1 of a pair: patch columns 4 sp + h and 4 sp + 2 + h (MFMA steps 2 sp, 2 sp + 1)
        f32x2 dd[4][4], gy[2][2];
        auto rd = [&](int sp) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dd[i][j] = f32x2{xp[(i * GIW + 8 * sp + j) * 32], xp[(i * GIW + 8 * sp + 4 + j) * 32]};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    gy[a][b] = f32x2{gp[(a * GTW + 8 * sp + b) * 32], gp[(a * GTW + 8 * sp + 4 + b) * 32]};
        };
        rd(0);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            f32x2 V[4][4], M[4][4];
            {
                f32x2 T[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T[0][j] = pk_sub(dd[0][j], dd[2][j]);
                    T[1][j] = pk_add(dd[1][j], dd[2][j]);
                    T[2][j] = pk_sub(dd[2][j], dd[1][j]);
                    T[3][j] = pk_sub(dd[1][j], dd[3][j]);
                }
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    V[xi][0] = pk_sub(T[xi][0], T[xi][2]);
                    V[xi][1] = pk_add(T[xi][1], T[xi][2]);
                    V[xi][2] = pk_sub(T[xi][2], T[xi][1]);
                    V[xi][3] = pk_sub(T[xi][1], T[xi][3]);
                }
                // A dY A^T with the factors -1 of row 3 and column 3 left out
                f32x2 R[4][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    R[0][b] = gy[0][b];
                    R[1][b] = pk_add(gy[0][b], gy[1][b]);
                    R[2][b] = pk_sub(gy[0][b], gy[1][b]);
                    R[3][b] = gy[1][b];
                }
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    M[xi][0] = R[xi][0];
                    M[xi][1] = pk_add(R[xi][0], R[xi][1]);
                    M[xi][2] = pk_sub(R[xi][0], R[xi][1]);
                    M[xi][3] = R[xi][1];
                }
                if (do_bias) bs2 = pk_add(bs2, pk_add(pk_add(gy[0][0], gy[0][1]), pk_add(gy[1][0], gy[1][1])));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sp + 1 < 4) rd(sp + 1);
            // the compiler does not see that the MFMAs below read registers the packed instructions
            // above have just written
            asm volatile("s_nop 1" ::: "memory");
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) mfma_acc(acc[xi * 4 + nu], V[xi][nu][e], M[xi][nu][e]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more_tiles) stage_store(buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf ^= 1;
        t += gridDim.x;
        cur = nxt;
    }

    // slabs: partial[((split * tiles_total + tile) * 16 + reg) * 64 + lane], split = (workgroup, wave),
    // tile = position * kchunks + kc
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    const long split = (long)bx * 4 + wv;
    const long tiles_total = 16L * kchunks;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const f32x16 m = acc_read16(acc[p]);
        float* dst = partial + ((split * tiles_total + (long)p * kchunks + kc) * 16) * 64 + lane;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) dst[reg * 64] = m[reg];
    }
    if (do_bias) bpartial[split * 64 + lane] = bs2[0] + bs2[1];
}

// dw[tap a,b][k][n] += sum over (xi, nu) of G[xi][a] G[nu][b] s(xi) s(nu) dU[xi,nu][k][n] with
// G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] and s(3) = -1 (the sign the gradient transform left out).
__global__ __launch_bounds__(256) void wino3_wgrad_finish_kernel(mgx_conv_desc d, const float* __restrict__ du,
                                                                 float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= d.cin * 32) return;
    const int k = idx >> 5, n = idx & 31;
    float u[4][4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float v = du[((long)(xi * 4 + nu) * d.cin + k) * 32 + n];
            u[xi][nu] = ((xi == 3) != (nu == 3)) ? -v : v;
        }
    float t[3][4];                      // G^T u
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
        t[0][nu] = u[0][nu] + 0.5f * (u[1][nu] + u[2][nu]);
        t[1][nu] = 0.5f * (u[1][nu] - u[2][nu]);
        t[2][nu] = u[3][nu] + 0.5f * (u[1][nu] + u[2][nu]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float g0 = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        const float g1 = 0.5f * (t[a][1] - t[a][2]);
        const float g2 = t[a][3] + 0.5f * (t[a][1] + t[a][2]);
        float* dst = dw + (long)(a * 3) * d.wtap + (long)k * d.wsk + (long)n * d.wsn;
        dst[0] += g0;
        dst[d.wtap] += g1;
        dst[2 * d.wtap] += g2;
    }
}

int wgrid(long ntiles) { return (int)(ntiles < 256 ? ntiles : 256); }

}  // namespace

// 3x3 stride-1 'same' convolutions with 32 output channels, input channels a multiple of 32, image
// height a multiple of 8.
bool mgx_wwgrad_applicable(const mgx_conv_desc& d) {
    if (d.kh != 3 || d.kw != 3 || d.sh != 1 || d.sw != 1 || d.dkh != 1 || d.dkw != 1 || d.flip) return false;
    if (d.pt != 1 || d.pl != 1 || d.hout != d.hin || d.wout != d.win) return false;
    if (d.hin % GTH || d.cin % 32 || d.cin < 32 || d.ngemm != 32 || d.cout != 32) return false;
    if (d.nbias != 0 && d.nbias != 32) return false;
    if (((long)GIH * d.win + GIW) * d.cin * 4 >= (1L << 31)) return false;
    const long ntiles = (long)d.nimg * (d.hin / GTH) * ((d.win + GTW - 1) / GTW);
    return ntiles < (1L << 31);
}

long mgx_wwgrad_splits(const mgx_conv_desc& d) {
    return (long)wgrid((long)d.nimg * (d.hin / GTH) * ((d.win + GTW - 1) / GTW)) * 4;
}

// floats of the transform-domain gradient dU[16][cin][32] between the slab reduction and the final transform
long mgx_wwgrad_du_floats(const mgx_conv_desc& d) { return 16L * d.cin * 32; }

int mgx_wwgrad_launch(const mgx_conv_desc& d, const float* x, const float* dy, const float* dymul, float* partial,
                      float* bpartial, hipStream_t st) {
    const int tiles_h = d.hin / GTH, tiles_w = (d.win + GTW - 1) / GTW;
    const long ntiles = (long)d.nimg * tiles_h * tiles_w;
    dim3 grid(wgrid(ntiles), d.cin / 32), block(256);
    static std::atomic<uint64_t> allowed_mul{0}, allowed_plain{0};
    mgx_allow_dynamic_lds(wino3_wgrad_kernel<true>, G_LDS_BYTES, allowed_mul);
    mgx_allow_dynamic_lds(wino3_wgrad_kernel<false>, G_LDS_BYTES, allowed_plain);
    if (dymul)
        hipLaunchKernelGGL((wino3_wgrad_kernel<true>), grid, block, G_LDS_BYTES, st, d, x, dy, dymul, partial, bpartial,
                           ntiles, tiles_h, tiles_w);
    else
        hipLaunchKernelGGL((wino3_wgrad_kernel<false>), grid, block, G_LDS_BYTES, st, d, x, dy, dymul, partial,
                           bpartial, ntiles, tiles_h, tiles_w);
    return mgx_launch_status();
}

int mgx_wwgrad_finish(const mgx_conv_desc& d, const float* du, float* dw, hipStream_t st) {
    hipLaunchKernelGGL(wino3_wgrad_finish_kernel, dim3((unsigned)((d.cin * 32 + 255) / 256)), dim3(256), 0, st, d, du, dw);
    return mgx_launch_status();
}
