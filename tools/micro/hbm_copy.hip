// Micro-benchmark: what rate a streaming kernel reaches on this part for the read : write mixes of the HBM-bound
// kernels of the step, and what the launch shape and the cache policy of its loads and stores change.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_copy.hip -o tools/micro/hbm_copy && ./hbm_copy
// R input streams are read and W output streams written, 16 bytes a lane, grid-stride, `waves` waves per CU;
// policies: plain, nontemporal loads (nt), nontemporal stores (nt), both.  Prints TB/s of bytes moved.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n, long stride_in,
                                                     long stride_out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f32x4 v = NTL ? __builtin_nontemporal_load(in + r * stride_in + i) : in[r * stride_in + i];
            acc += v;
        }
        if (W == 0 && acc[0] == 12345.678f) out[i] = acc;       // never true: keeps the loads alive
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const f32x4 v = acc * (float)(w + 1);
            if (NTS) __builtin_nontemporal_store(v, out + w * stride_out + i);
            else out[w * stride_out + i] = v;
        }
    }
}

// two-level unroll: each thread keeps U independent 16-byte loads in flight before it stores
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_unrolled(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n) {
    const long step = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) v[u] = NTL ? __builtin_nontemporal_load(in + i + u * step) : in[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * step < n) {
                if (NTS) __builtin_nontemporal_store(v[u], out + i + u * step);
                else out[i + u * step] = v[u];
            }
    }
}

template <typename F>
double timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

template <int R, int W>
void mix(const char* name, const f32x4* in, f32x4* out, long n, long stride) {
    printf("%-34s", name);
    for (int waves : {8, 16, 32}) {
        const int grid = 256 * waves / 4;
        const double bytes = (double)(R + W) * n * 16;
        double t0 = timed([&] { hipLaunchKernelGGL((stream_kernel<R, W, false, false>), dim3(grid), dim3(256), 0, 0, in, out, n, stride, stride); });
        double t1 = timed([&] { hipLaunchKernelGGL((stream_kernel<R, W, true, false>), dim3(grid), dim3(256), 0, 0, in, out, n, stride, stride); });
        double t2 = timed([&] { hipLaunchKernelGGL((stream_kernel<R, W, false, true>), dim3(grid), dim3(256), 0, 0, in, out, n, stride, stride); });
        double t3 = timed([&] { hipLaunchKernelGGL((stream_kernel<R, W, true, true>), dim3(grid), dim3(256), 0, 0, in, out, n, stride, stride); });
        printf("  %2d waves/CU: plain %.2f  nt-load %.2f  nt-store %.2f  both %.2f", waves, bytes / t0 / 1e12, bytes / t1 / 1e12,
               bytes / t2 / 1e12, bytes / t3 / 1e12);
    }
    printf("  TB/s\n");
    fflush(stdout);
}

int main() {
    const long n = 3L << 26;                       // 16-byte elements per stream: 3 GiB a stream
    f32x4 *in, *out;
    if (hipMalloc(&in, 4 * n * 16) != hipSuccess || hipMalloc(&out, 2 * n * 16) != hipSuccess) { printf("no memory\n"); return 1; }
    hipMemset(in, 0, 4 * n * 16);
    hipMemset(out, 0, 2 * n * 16);
    mix<1, 0>("read only (checksum)", in, out, n, n);
    mix<1, 1>("1 read : 1 write (copy)", in, out, n, n);
    mix<2, 1>("2 reads : 1 write (blend, add)", in, out, n, n);
    mix<1, 2>("1 read : 2 writes (taped forward)", in, out, n, n);
    mix<3, 2>("3 reads : 2 writes (blend backward)", in, out, n, n);
    mix<4, 1>("4 reads : 1 write", in, out, n, n);
    printf("copy with U loads in flight per thread, 16 waves/CU:\n");
    for (int rep = 0; rep < 1; ++rep) {
        const int grid = 256 * 4;
        const double bytes = 2.0 * n * 16;
        double a = timed([&] { hipLaunchKernelGGL((copy_unrolled<2, false, false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
        double b = timed([&] { hipLaunchKernelGGL((copy_unrolled<4, false, false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
        double c = timed([&] { hipLaunchKernelGGL((copy_unrolled<8, false, false>), dim3(grid), dim3(256), 0, 0, in, out, n); });
        double d = timed([&] { hipLaunchKernelGGL((copy_unrolled<4, true, true>), dim3(grid), dim3(256), 0, 0, in, out, n); });
        double e = timed([&] { hipLaunchKernelGGL((copy_unrolled<8, true, true>), dim3(grid), dim3(256), 0, 0, in, out, n); });
        printf("  U=2 %.2f  U=4 %.2f  U=8 %.2f  U=4 nt %.2f  U=8 nt %.2f TB/s\n", bytes / a / 1e12, bytes / b / 1e12, bytes / c / 1e12,
               bytes / d / 1e12, bytes / e / 1e12);
    }
    hipFree(in);
    hipFree(out);
    return 0;
}
