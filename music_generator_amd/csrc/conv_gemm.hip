// Implicit-GEMM forward / data-gradient contraction on v_mfma_f32_32x32x2_f32 (gfx950).
//
// One wave owns PT tiles of 32 consecutive GEMM rows (output positions) and NT tiles of 32 GEMM
// columns.  The A operand (activations) is read straight from global memory as one float4 per
// lane along the contraction axis (lane half h takes k0+4h..k0+4h+3 of row l&31), the B operand
// (weights, L2 resident) as four dwords (row-major [k][n]) or one float4 ([n][k]) per lane.  The
// order in which k is consumed is free, so step j of a chunk pairs k0+j (lanes 0-31) with
// k0+4+j (lanes 32-63) on both operands.  No LDS: the f32 MFMA takes 64 cycles per SIMD, which
// leaves the load path far from saturated.
#include <cstdlib>

#include "common.h"
#include "../../include/mgx.h"

bool mgx_tconv_applicable(const mgx_conv_desc& d);
bool mgx_wconv_applicable(const mgx_conv_desc& d, bool mul);
int mgx_wconv_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                     float* y, float* dact, hipStream_t st);
int mgx_tconv_launch(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                     float* y, float* dact, hipStream_t st);
bool mgx_pgemm_applicable(const mgx_conv_desc& d, bool mul);   // pgemm.hip
bool mgx_tapsum_applicable(const mgx_conv_desc& d, bool mul);
int mgx_tapsum_launch(const mgx_conv_desc& d, const float* x, const float* w, float* y, hipStream_t st);
int mgx_pgemm_launch(const mgx_conv_desc& d, const float* x, const float* w, const float* bias, float* y,
                     hipStream_t st);

namespace {

struct RowPos {
    long img;
    int ho, wo;
};

__device__ __forceinline__ RowPos decode_row(long row, int hout, int wout) {
    const int hw = hout * wout;
    RowPos p;
    p.img = row / hw;
    const int rem = (int)(row - p.img * hw);
    p.ho = rem / wout;
    p.wo = rem - p.ho * wout;
    return p;
}

template <int PT, int NT, int WMODE, bool KVEC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(mgx_conv_desc d, const float* __restrict__ x,
                                                        const float* __restrict__ xmul,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ dact) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long R = (long)d.nimg * d.hout * d.wout;
    const long row0 = ((long)blockIdx.x * 4 + wv) * (PT * 32);
    if (row0 >= R) return;  // wave-uniform
    const int n0 = blockIdx.y * (NT * 32);

    bool valid[PT];
    long img[PT];
    int ho[PT], wo[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const long row = row0 + pt * 32 + r;
        valid[pt] = row < R;
        const RowPos p = decode_row(valid[pt] ? row : 0, d.hout, d.wout);
        img[pt] = p.img; ho[pt] = p.ho; wo[pt] = p.wo;
    }

    f32x16 acc[PT][NT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[pt][nt][i] = 0.f;

    int ncol[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ncol[nt] = n0 + nt * 32 + r;
        nok[nt] = ncol[nt] < d.ngemm;
    }

    // One flat loop over (tap, 8-channel step) with the operands of the next step loaded before the
    // MFMAs of the current one: the loads are per-lane gathers straight from global memory, their
    // latency must not sit between every two groups of MFMAs.
    const int ksteps = (d.cin + 7) / 8;
    const int total = d.kh * d.kw * ksteps;
    int ta = 0, tb = 0, tk = 0;                 // tap and channel step of the next load
    const float* wtap = w;
    bool ok[PT];
    long xoff[PT];
    auto tap_state = [&]() {
        const int wt = d.flip ? ((d.kh - 1 - ta) * d.kw + (d.kw - 1 - tb)) : (ta * d.kw + tb);
        wtap = w + (long)wt * d.wtap;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int hi = ho[pt] * d.sh + ta - d.pt;
            const int wi = wo[pt] * d.sw + tb - d.pl;
            ok[pt] = valid[pt] && hi >= 0 && hi < d.hin && wi >= 0 && wi < d.win;
            xoff[pt] = ((img[pt] * d.hin + hi) * d.win + wi) * (long)d.cin;
        }
    };
    auto load = [&](f32x4 (&xa)[PT], f32x4 (&wb)[NT]) {
        const int kk = tk * 8 + 4 * h;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            xa[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (ok[pt]) {
                if (KVEC) {
                    xa[pt] = *(const f32x4*)(x + xoff[pt] + kk);
                    if (xmul) xa[pt] *= *(const f32x4*)(xmul + xoff[pt] + kk);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kk + j < d.cin) {
                            float v = x[xoff[pt] + kk + j];
                            if (xmul) v *= xmul[xoff[pt] + kk + j];
                            xa[pt][j] = v;
                        }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wb[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (nok[nt]) {
                if (WMODE == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (KVEC || kk + j < d.cin) wb[nt][j] = wtap[(long)(kk + j) * d.wsk + ncol[nt]];
                } else {
                    if (KVEC) {
                        wb[nt] = *(const f32x4*)(wtap + (long)ncol[nt] * d.wsn + kk);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (kk + j < d.cin) wb[nt][j] = wtap[(long)ncol[nt] * d.wsn + kk + j];
                    }
                }
            }
        }
        // advance to the step after this one
        if (++tk == ksteps) {
            tk = 0;
            if (++tb == d.kw) tb = 0, ++ta;
            if (ta < d.kh) tap_state();
        }
    };
    auto mma = [&](const f32x4 (&xa)[PT], const f32x4 (&wb)[NT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = mfma32(xa[pt][j], wb[nt][j], acc[pt][nt]);
    };
    {
        f32x4 xa0[PT], wb0[NT], xa1[PT], wb1[NT];
        tap_state();
        load(xa0, wb0);
        for (int s = 0; s < total; s += 2) {
            if (s + 1 < total) load(xa1, wb1);
            mma(xa0, wb0);
            if (s + 1 < total) {
                if (s + 2 < total) load(xa0, wb0);
                mma(xa1, wb1);
            }
        }
    }

    // ---- epilogue: bias, activation, depth-to-space store ----
    // with dkh == 1 the dkw x cout columns of a row are contiguous in the output: no scatter needed
    const bool scatter = d.dkh != 1;
    const float inv_alpha = d.act_alpha > 0.f ? 1.0f / d.act_alpha : 0.f;
    const long out_w = (long)d.wout * d.dkw;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (!nok[nt]) continue;
        const int n = ncol[nt];
        const float bv = d.nbias > 0 ? bias[n % d.nbias] : 0.f;
        long npart = n;
        if (scatter) {
            const int ab = n / d.cout, co = n - ab * d.cout;
            const int a2 = ab / d.dkw, b2 = ab - a2 * d.dkw;
            npart = (a2 * out_w + b2) * d.cout + co;
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const long rbase = row0 + pt * 32;
            RowPos p0;
            if (scatter) p0 = decode_row(rbase, d.hout, d.wout);
            // offsets first, then loads (accumulate), arithmetic and stores in separate sweeps so
            // that the memory operations of the 16 elements are in flight together
            long off[16];
            bool okr[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = acc_row(reg, lane);
                const long row = rbase + rl;
                okr[reg] = row < R;
                if (scatter) {
                    long im = p0.img;
                    int hh = p0.ho, ww = p0.wo + rl;
                    while (ww >= d.wout) { ww -= d.wout; ++hh; }
                    while (hh >= d.hout) { hh -= d.hout; ++im; }
                    off[reg] = ((im * d.hout + hh) * d.dkh * out_w + (long)ww * d.dkw) * d.cout + npart;
                } else {
                    off[reg] = row * d.ngemm + npart;
                }
                if (!okr[reg]) off[reg] = 0;
            }
            float v[16], der[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) v[reg] = acc[pt][nt][reg] + bv;
            const bool act = d.act_alpha > 0.f;
            if (act) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float t, s2;
                    tanh_sech2(v[reg], t, s2);
                    der[reg] = s2 + inv_alpha;
                    v[reg] = t + v[reg] / d.act_alpha;
                }
            }
            if (d.accumulate) {
                float old[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) old[reg] = okr[reg] ? y[off[reg]] : 0.f;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) v[reg] += old[reg];
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                if (okr[reg]) y[off[reg]] = v[reg];
            if (act && dact) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    if (okr[reg]) dact[off[reg]] = der[reg];
            }
        }
    }
}

template <int PT, int NT>
int launch_pn(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
              float* y, float* dact, hipStream_t st) {
    const long R = (long)d.nimg * d.hout * d.wout;
    const long rows_per_block = 4L * PT * 32;
    const long gx = (R + rows_per_block - 1) / rows_per_block;
    const int gy = (d.ngemm + NT * 32 - 1) / (NT * 32);
    if (gx <= 0 || gx > 0x7fffffffL) return MGX_ERR_ARG;
    dim3 grid((unsigned)gx, (unsigned)gy), block(256);
    const bool kvec = (d.cin % 8) == 0;
    const int wmode = (d.wsn == 1) ? 0 : 1;
    if (wmode == 0) {
        if (kvec) hipLaunchKernelGGL((conv_gemm_kernel<PT, NT, 0, true>), grid, block, 0, st, d, x, xmul, w, bias, y, dact);
        else hipLaunchKernelGGL((conv_gemm_kernel<PT, NT, 0, false>), grid, block, 0, st, d, x, xmul, w, bias, y, dact);
    } else {
        if (kvec) hipLaunchKernelGGL((conv_gemm_kernel<PT, NT, 1, true>), grid, block, 0, st, d, x, xmul, w, bias, y, dact);
        else hipLaunchKernelGGL((conv_gemm_kernel<PT, NT, 1, false>), grid, block, 0, st, d, x, xmul, w, bias, y, dact);
    }
    return mgx_launch_status();
}

// ------------------------------------------------------------------------------------------
// 1x1 convolution of a 2-channel input (the discriminator's from-audio layers, models.py:236 and
// 262): two multiply-adds per output, so it runs on the vector ALUs at the HBM write rate.  A
// thread owns 4 consecutive output channels (weights and bias in registers) and walks positions;
// a wave writes two whole 128-channel rows (1 KiB) per store instruction.
// ------------------------------------------------------------------------------------------
template <bool ACT, bool DACT>
__global__ __launch_bounds__(256) void pointwise2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ dact, long npos, int n, float alpha,
                                                         int pos_per_block) {
    const int q = n >> 2;                         // threads per position
    const int c = (threadIdx.x % q) * 4, pl = threadIdx.x / q, pstep = 256 / q;
    const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + n + c), b = *(const f32x4*)(bias + c);
    const float inv_alpha = ACT ? 1.0f / alpha : 0.f;
    const long p0 = (long)blockIdx.x * pos_per_block;
    const long p1 = p0 + pos_per_block < npos ? p0 + pos_per_block : npos;
#pragma unroll 2
    for (long p = p0 + pl; p < p1; p += pstep) {
        const float2 xv = *(const float2*)(x + p * 2);
        f32x4 v = w0 * xv.x + (w1 * xv.y + b);
        if (ACT) {
            f32x4 der;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t, s2;
                tanh_sech2(v[j], t, s2);
                der[j] = s2 + inv_alpha;
                v[j] = t + v[j] / alpha;
            }
            if (DACT) __builtin_nontemporal_store(der, (f32x4*)(dact + p * n + c));
        }
        __builtin_nontemporal_store(v, (f32x4*)(y + p * n + c));
    }
}

bool pointwise2_applicable(const mgx_conv_desc& d, const float* xmul) {
    return d.cin == 2 && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.dkh == 1 && d.dkw == 1 &&
           d.pt == 0 && d.pl == 0 && d.hout == d.hin && d.wout == d.win && d.ngemm == d.cout && d.ngemm % 4 == 0 &&
           256 % (d.ngemm / 4) == 0 && d.ngemm <= 1024 && d.wsn == 1 && d.wsk == d.ngemm && d.nbias == d.ngemm &&
           !d.accumulate && !xmul;
}

int launch_pointwise2(const mgx_conv_desc& d, const float* x, const float* w, const float* bias, float* y,
                      float* dact, hipStream_t st) {
    const long npos = (long)d.nimg * d.hin * d.win;
    const int pstep = 256 / (d.ngemm / 4);
    // about 8 workgroups per CU, whole multiples of the positions one pass covers
    long ppb = (npos + 2047) / 2048;
    ppb = ((ppb + pstep - 1) / pstep) * pstep;
    if (ppb < pstep) ppb = pstep;
    const long blocks = (npos + ppb - 1) / ppb;
    dim3 grid((unsigned)blocks), block(256);
    const bool act = d.act_alpha > 0.f;
    if (act && dact)
        hipLaunchKernelGGL((pointwise2_kernel<true, true>), grid, block, 0, st, x, w, bias, y, dact, npos, d.ngemm,
                           d.act_alpha, (int)ppb);
    else if (act)
        hipLaunchKernelGGL((pointwise2_kernel<true, false>), grid, block, 0, st, x, w, bias, y, dact, npos, d.ngemm,
                           d.act_alpha, (int)ppb);
    else
        hipLaunchKernelGGL((pointwise2_kernel<false, false>), grid, block, 0, st, x, w, bias, y, dact, npos, d.ngemm,
                           1.0f, (int)ppb);
    return mgx_launch_status();
}

}  // namespace

bool mgx_pointwise2_applicable(const mgx_conv_desc& d) { return pointwise2_applicable(d, nullptr); }

namespace {

// The value of another lane of the same 16-lane row, chosen by a DPP control word (a vector ALU move, no LDS
// crossbar round trip).
template <int CTRL>
__device__ __forceinline__ float row_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// 1x1 convolution to 2 output channels (the data-gradient of the from-audio layers): two dot
// products per position over C = 4 q input channels, read once at the HBM rate.  q lanes share a
// position (16 bytes each, optionally times the multiplier), reduce across lanes, lane 0 stores.
template <bool MUL>
__global__ __launch_bounds__(256) void pointwise_to2_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ xmul,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, long npos, int cin, int wsk, int wsn,
                                                            int nbias, int pos_per_block) {
    const int q = cin >> 2;                       // lanes per position: a power of two <= 64
    const int c = (threadIdx.x % q) * 4, pl = threadIdx.x / q, pstep = 256 / q;
    f32x4 w0, w1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w0[j] = w[(long)(c + j) * wsk];
        w1[j] = w[(long)(c + j) * wsk + wsn];
    }
    const float b0 = nbias > 0 ? bias[0] : 0.f, b1 = nbias > 0 ? bias[1 % nbias] : 0.f;
    const long p0 = (long)blockIdx.x * pos_per_block;
    const long p1 = p0 + pos_per_block < npos ? p0 + pos_per_block : npos;
    const long span = p1 - p0;
    // every lane runs the same number of iterations (the shuffles need all lanes of a group)
    for (long i = pl; i < ((span + pstep - 1) / pstep) * pstep; i += pstep) {
        const long p = p0 + i;
        const bool ok = i < span;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) {
            v = __builtin_nontemporal_load((const f32x4*)(x + p * cin + c));
            if (MUL) v *= __builtin_nontemporal_load((const f32x4*)(xmul + p * cin + c));
        }
        float s0 = v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
        float s1 = v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
        // Sum over the q lanes of a position.  Lanes of one 16-lane row exchange through the vector ALU's own
        // lane crossbar (quad permutes, then mirrors: after the quad steps every lane of a quad holds the quad's
        // sum, so the mirrored lane supplies the other quad's / half's sum); only the two widest steps go through
        // the LDS crossbar, with offsets known at compile time.
        if (q > 1) { s0 += row_lane<0xB1>(s0); s1 += row_lane<0xB1>(s1); }     // quad_perm [1,0,3,2]
        if (q > 2) { s0 += row_lane<0x4E>(s0); s1 += row_lane<0x4E>(s1); }     // quad_perm [2,3,0,1]
        if (q > 4) { s0 += row_lane<0x141>(s0); s1 += row_lane<0x141>(s1); }   // row_half_mirror
        if (q > 8) { s0 += row_lane<0x140>(s0); s1 += row_lane<0x140>(s1); }   // row_mirror
        if (q > 16) { s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64); }
        if (q > 32) { s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64); }
        if (ok && c == 0) *(float2*)(y + p * 2) = make_float2(s0 + b0, s1 + b1);
    }
}

bool pointwise_to2_applicable(const mgx_conv_desc& d) {
    const int q = d.cin / 4;
    return d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.dkh == 1 && d.dkw == 1 && d.pt == 0 && d.pl == 0 &&
           d.hout == d.hin && d.wout == d.win && d.ngemm == 2 && d.cout == 2 && d.cin % 4 == 0 && q >= 1 && q <= 64 &&
           (q & (q - 1)) == 0 && d.act_alpha <= 0.f && !d.accumulate && (d.nbias == 0 || d.nbias == 2);
}

int launch_pointwise_to2(const mgx_conv_desc& d, const float* x, const float* xmul, const float* w, const float* bias,
                         float* y, hipStream_t st) {
    const long npos = (long)d.nimg * d.hin * d.win;
    const int ppb = 512;
    dim3 grid((unsigned)((npos + ppb - 1) / ppb)), block(256);
    if (xmul)
        hipLaunchKernelGGL((pointwise_to2_kernel<true>), grid, block, 0, st, x, xmul, w, bias, y, npos, d.cin, d.wsk,
                           d.wsn, d.nbias, ppb);
    else
        hipLaunchKernelGGL((pointwise_to2_kernel<false>), grid, block, 0, st, x, xmul, w, bias, y, npos, d.cin, d.wsk,
                           d.wsn, d.nbias, ppb);
    return mgx_launch_status();
}

}  // namespace

// route 2 also covers the 1x1 convolutions to 2 output channels (any input multiplier)
bool mgx_pointwise_to2_applicable(const mgx_conv_desc& d) { return pointwise_to2_applicable(d); }

// The Winograd kernels are used unless the environment says MGX_WINOGRAD=0.  The variable is read at
// every call (forward, data-gradient, weight-gradient workspace query and launch alike), so a process
// can compare the two kernel families on the same operands by changing it between calls; a caller that
// changes it between a workspace query and the launch it sized must query again.
bool mgx_wconv_enabled() {
    const char* e = getenv("MGX_WINOGRAD");
    return !(e && e[0] == '0');
}

extern "C" int mgx_conv_gemm_f32(const mgx_conv_desc* dp, const float* x, const float* xmul, const float* w,
                                 const float* bias, float* y, float* dact, void* stream) {
    if (!dp || !x || !w || !y) return MGX_ERR_ARG;
    const mgx_conv_desc d = *dp;
    if (d.nimg <= 0 || d.hin <= 0 || d.win <= 0 || d.cin <= 0 || d.hout <= 0 || d.wout <= 0) return MGX_ERR_ARG;
    if (d.kh <= 0 || d.kw <= 0 || d.sh <= 0 || d.sw <= 0 || d.dkh <= 0 || d.dkw <= 0 || d.cout <= 0) return MGX_ERR_ARG;
    if (d.ngemm != d.dkh * d.dkw * d.cout) return MGX_ERR_ARG;
    if (d.wsk != 1 && d.wsn != 1) return MGX_ERR_ARG;
    if (d.nbias > 0 && !bias) return MGX_ERR_ARG;
    // the last gather position of every row grid must be reachable by construction of the layer:
    // out-of-range positions are treated as zero padding, never dereferenced.
    hipStream_t st = (hipStream_t)stream;
    if (mgx_wconv_enabled() && mgx_wconv_applicable(d, xmul != nullptr))
        return mgx_wconv_launch(d, x, xmul, w, bias, y, dact, st);
    if (mgx_tconv_applicable(d)) return mgx_tconv_launch(d, x, xmul, w, bias, y, dact, st);
    if (pointwise2_applicable(d, xmul)) return launch_pointwise2(d, x, w, bias, y, dact, st);
    if (pointwise_to2_applicable(d) && !dact) return launch_pointwise_to2(d, x, xmul, w, bias, y, st);
    if (mgx_pgemm_applicable(d, xmul != nullptr)) return mgx_pgemm_launch(d, x, w, bias, y, st);
    if (mgx_tapsum_applicable(d, xmul != nullptr) && !dact) return mgx_tapsum_launch(d, x, w, y, st);
    const int ntiles = (d.ngemm + 31) / 32;
    if (ntiles == 1) return launch_pn<4, 1>(d, x, xmul, w, bias, y, dact, st);
    if (ntiles == 2) return launch_pn<2, 2>(d, x, xmul, w, bias, y, dact, st);
    return launch_pn<2, 4>(d, x, xmul, w, bias, y, dact, st);
}
