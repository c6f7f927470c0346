#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, float* out) {
    unsigned u = in[threadIdx.x];
    bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
    out[threadIdx.x] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, u), ones, 0.5f, false);
}
int main() {
    unsigned h[4] = {0x40004040u /* hi 2.0, lo 3.0 */, 0x3f800000u /* hi 1, lo 0 */, 0x00003f80u /* lo 1 */, 0xc0003f80u /* hi -2 lo 1 */};
    unsigned* d; float* o; hipMalloc(&d, 16); hipMalloc(&o, 16);
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o);
    float r[4]; hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    printf("%g %g %g %g (expect 5.5 1.5 1.5 -0.5)\n", r[0], r[1], r[2], r[3]);
    return 0;
}
