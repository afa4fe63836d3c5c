// Bare bf16 MFMA loop, operands in registers (random bits), one or two waves per SIMD: v_mfma_f32_32x32x16_bf16 against
// v_mfma_f32_16x16x32_bf16 at the same FLOPs per wave — TFLOP/s and the in-kernel shader clock (MI355X_MICROARCH.md, DVFS item 7).
//   usage: mfma_shape_bench <waves per SIMD 1|2> [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int int4r __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const int4r* __restrict__ ops, float* __restrict__ out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    int4r a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(t * 8 + i) & 0xffff]; b[i] = ops[(t * 8 + 4 + i) & 0xffff]; }
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float res = 0.f;
    if (SHAPE == 32) {
        floatx16 acc[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    } else {
        floatx4 acc[16] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {                       // two 16x16x32 per 32x32x16 of the other loop: same FLOPs
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i * 4 + j], 0, 0, 0);
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[i]), acc[i * 4 + j], 0, 0, 0);
                }
        for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) res += acc[i][r];
    }
    out[t] = res;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int SHAPE>
static void run(int wps, int iters) {
    const int blocks = 256 * wps;
    std::vector<int> h(65536 * 4);
    srand(3);
    for (auto& v : h) {                                           // random bf16 pairs with moderate exponents
        const unsigned lo = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
        v = (int)(lo | (hi << 16));
    }
    int4r* dops; float* dout; unsigned long long* dclk;
    hipMalloc(&dops, h.size() * 4); hipMalloc(&dout, blocks * 256 * 4); hipMalloc(&dclk, blocks * 16);
    hipMemcpy(dops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, dops, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, dops, dout, iters, dclk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> c(blocks * 2);
    hipMemcpy(c.data(), dclk, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> g;
    for (int b = 0; b < blocks; ++b) if (c[2 * b + 1]) g.push_back((double)c[2 * b] / c[2 * b + 1] * 0.1);
    std::sort(g.begin(), g.end());
    const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("shape %s  %d waves/SIMD  iters %d: %.3f ms  %.0f TFLOP/s (bf16)  clock %.2f GHz\n", SHAPE == 32 ? "32x32x16" : "16x16x32", wps, iters, ms,
           flop / ms * 1e-9, g.empty() ? 0.0 : g[g.size() / 2]);
    hipFree(dops); hipFree(dout); hipFree(dclk);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    run<32>(wps, iters); run<16>(wps, iters); run<32>(wps, iters); run<16>(wps, iters);
    return 0;
}
