// Microbenchmark: one 128x128x16 k-tile worth of fp32 MFMA work per iteration, operands read from LDS in two layouts:
//   A: k-major [16][132] with ds_read_b32 per k-step (the layout of conv_igemm.hip today)
//   B: m-major [128][20] with one ds_read_b128 per operand per 4 k-steps (proposed)
// No global traffic inside the loop (LDS content is static): isolates LDS-read / MFMA-issue interaction.
// usage: mfma_lds_bench <waves_per_simd 1|2|3> ; prints TFLOP/s for both layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2000;

__global__ __launch_bounds__(256) void kmajor(float* out, int iters) {
    __shared__ float As[16][132];
    __shared__ float Bs[16][132];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l32 = lane & 31, kh = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    for (int i = tid; i < 16 * 132; i += 256) { (&As[0][0])[i] = 0.001f * (i % 97); (&Bs[0][0])[i] = 0.002f * (i % 89); }
    __syncthreads();
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int k = 2 * ks + kh;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[k][wm * 64 + i * 32 + l32];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[k][wn * 64 + j * 32 + l32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

__global__ __launch_bounds__(256) void mmajor(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128][20];
    __shared__ __attribute__((aligned(16))) float Bs[128][20];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l32 = lane & 31, kh = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    for (int i = tid; i < 128 * 20; i += 256) { (&As[0][0])[i] = 0.001f * (i % 97); (&Bs[0][0])[i] = 0.002f * (i % 89); }
    __syncthreads();
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f4*>(&As[wm * 64 + i * 32 + l32][8 * h + 4 * kh]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f4*>(&Bs[wn * 64 + j * 32 + l32][8 * h + 4 * kh]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int blocks = 256 * wps;
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(kmajor, dim3(blocks), dim3(256), 0, 0, out, ITERS);
            else hipLaunchKernelGGL(mmajor, dim3(blocks), dim3(256), 0, 0, out, ITERS);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 128 * 128 * 16 * (double)ITERS * blocks;
            if (rep == 1) printf("%s layout, %d workgroup(s) per CU: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "k-major b32" : "m-major b128", wps, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
