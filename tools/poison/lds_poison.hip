// Fills the LDS of every CU with a NaN pattern (0xFFFFFFFF words): launched in front of other kernels it makes a
// read of LDS that the reading kernel did not write itself visible as NaN in the results.  One workgroup takes the
// whole 160 KB of a CU, 1024 workgroups with a short spin each so that every CU is visited.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/poison/lds_poison.hip -o tools/poison/liblds_poison.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void lds_poison_kernel(unsigned* sink, int spin) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) lds[i] = 0xFFFFFFFFu;
    __syncthreads();
    unsigned acc = 0;
    for (int k = 0; k < spin; ++k) acc += lds[(threadIdx.x * 7 + k) % (160 * 1024 / 4)];
    if (acc == 12345u) sink[0] = acc;
}

extern "C" int lds_poison(void* stream, unsigned* sink) {
    static bool allowed = false;
    if (!allowed) {
        if (hipFuncSetAttribute((const void*)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return 1;
        allowed = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(1024), 160 * 1024, (hipStream_t)stream, sink, 64);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
