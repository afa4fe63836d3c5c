// Shared host/device helpers for libreidgan_hip.so (gfx950 / MI355X only).
//
// Conventions of the C ABI (see include/reidgan_hip.h):
//   * every entry point returns 0 on success, a negative rg_status on failure and records a
//     message retrievable through rg_last_error();
//   * nothing here allocates, frees or synchronises: the caller owns all device memory and
//     passes the HIP stream the work is enqueued on;
//   * all tensors are contiguous fp32 NCHW unless a signature says otherwise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define RG_OK 0
#define RG_ERR_INVALID (-1)
#define RG_ERR_LAUNCH (-2)
#define RG_ERR_WORKSPACE (-3)

// mirror of include/reidgan_hip.h (the public header is C and is not included by the kernels)
#define RG_SN_MAX_BATCH 16
typedef struct rg_sn_desc {
    const float* w;
    float* u;
    float* v;
    float* w_sn;
    float* sigma;
    float* uv_saved;
    int K;
    int M;
} rg_sn_desc;

namespace rg {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Optional per-launch timing (HIP events on the launch stream), grouped by kernel family.
// Enabled through rg_profile_enable(); used by bench.py for the roofline object.
enum Family : int {
    FAM_CONV_FWD = 0,
    FAM_CONV_DGRAD = 1,
    FAM_CONV_WGRAD = 2,
    FAM_NORM = 3,
    FAM_ELTWISE = 4,
    FAM_POOL = 5,
    FAM_LOSS = 6,
    FAM_CM = 7,
    FAM_OPTIM = 8,
    FAM_MISC = 9,
    FAM_CONV_F8 = 10,
    FAM_COUNT = 11
};

struct ProfScope {
    int fam;
    hipStream_t stream;
    int slot;
    double flops;
    ProfScope(int fam, hipStream_t stream, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace rg

#define RG_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            rg::set_error(__VA_ARGS__);       \
            return RG_ERR_INVALID;            \
        }                                     \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#define RG_WAVE 64

__device__ __forceinline__ float rg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float rg_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024). `red` is >= 16 floats of LDS.
// Every thread gets the total.
__device__ __forceinline__ float rg_block_sum(float v, float* red) {
    v = rg_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float rg_block_max(float v, float* red) {
    v = rg_wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// activation codes shared by conv epilogues and the element-wise kernels
#define RG_ACT_NONE 0
#define RG_ACT_RELU 1
#define RG_ACT_LEAKY 2
#define RG_ACT_TANH 3

__device__ __forceinline__ float rg_apply_act(float v, int act, float slope) {
    switch (act) {
        case RG_ACT_RELU: return v > 0.f ? v : 0.f;
        case RG_ACT_LEAKY: return v > 0.f ? v : v * slope;
        case RG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
