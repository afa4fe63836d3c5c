#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int int4v __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, unsigned bytes) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x == 3) off = 0x80000000u + 16;
    int4v v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    y[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v.x);
    y[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v.y);
    y[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v.z);
    y[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v.w);
    float s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4 + 256, 0, 0));
    y[64 * 4 + threadIdx.x] = s;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(y, 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 7.0f + threadIdx.x), ro, 2048 + threadIdx.x * 4 + (threadIdx.x == 5 ? 0x80000000u : 0u), 0, 0);
}
int main() {
    float *x, *y; hipMalloc(&x, 4096); hipMalloc(&y, 4096);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMemcpy(x, h, 4096, hipMemcpyHostToDevice); hipMemset(y, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 2048u);
    hipMemcpy(h, y, 4096, hipMemcpyDeviceToHost);
    for (int t = 0; t < 6; ++t) printf("t%d: %g %g %g %g\n", t, h[t*4], h[t*4+1], h[t*4+2], h[t*4+3]);
    printf("t33 (off 528 >= 2048? no): %g ; t63: %g\n", h[33*4], h[63*4]);
    printf("b32: %g %g %g\n", h[256], h[257], h[256+63]);
    printf("store: %g %g %g %g %g %g %g\n", h[512], h[513], h[514], h[515], h[516], h[517], h[518]);
    return 0;
}
