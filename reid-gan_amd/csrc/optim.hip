// Fused optimizer steps over flat parameter arenas (one launch per hyper-parameter group instead of
// torch.optim's per-tensor element-wise chains).  Semantics restate torch.optim.Adam / SGD as the
// reference configures them:
//   Adam(betas=(0.5,0.999)) for G (+E)              FD/fdgan/model.py:101-114
//   SGD(momentum=0.9, weight_decay=1e-4) for D      FD/fdgan/model.py:103-118
//   Adam(lr=3.5e-4, weight_decay=5e-4) for ReID     CC/examples/cluster_contrast_gan_train_usl_infomap.py:281-284
// weight_decay is the L2 form (added to the gradient), as in torch.optim.Adam (not AdamW).
#include "rg_common.h"

namespace {

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                            float bc1, float bc2_sqrt, float grad_scale) {
    const float step_size = lr / bc1;
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
#define RG_ADAM1(c)                                           \
    {                                                         \
        const float gg = G.c * grad_scale + wd * P.c;         \
        M.c = beta1 * M.c + (1.f - beta1) * gg;               \
        V.c = beta2 * V.c + (1.f - beta2) * gg * gg;          \
        const float denom = sqrtf(V.c) / bc2_sqrt + eps;      \
        P.c -= step_size * (M.c / denom);                     \
    }
        RG_ADAM1(x) RG_ADAM1(y) RG_ADAM1(z) RG_ADAM1(w)
#undef RG_ADAM1
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    for (int64_t e = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[e] * grad_scale + wd * p[e];
        const float mm = beta1 * m[e] + (1.f - beta1) * gg;
        const float vv = beta2 * v[e] + (1.f - beta2) * gg * gg;
        m[e] = mm;
        v[e] = vv;
        p[e] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}

// Device-side Adam clock (so that a captured hipGraph replays correct bias corrections): double state[4] =
// {step, beta1^step, beta2^step, -}; advanced once per optimizer step, read by adam_dev_kernel.
__global__ void adam_advance_kernel(double* __restrict__ st, double beta1, double beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st[0] += 1.0;
        st[1] *= beta1;
        st[2] *= beta2;
    }
}

__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                                const double* __restrict__ st, float grad_scale) {
    const float bc1 = (float)(1.0 - st[1]);
    const float bc2_sqrt = (float)sqrt(1.0 - st[2]);
    const float step_size = lr / bc1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float gg = g[e] * grad_scale + wd * p[e];
        const float mm = beta1 * m[e] + (1.f - beta1) * gg;
        const float vv = beta2 * v[e] + (1.f - beta2) * gg * gg;
        m[e] = mm;
        v[e] = vv;
        p[e] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}

__global__ void u64_add_kernel(unsigned long long* __restrict__ p, unsigned long long v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += v;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                           float lr, float momentum, float wd, int first_step, float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gg = g[i] * grad_scale + wd * p[i];
        if (momentum != 0.f) {
            const float b = first_step ? gg : momentum * buf[i] + gg;
            buf[i] = b;
            gg = b;
        }
        p[i] -= lr * gg;
    }
}

}  // namespace

// One Adam step on a contiguous range (step >= 1 is the 1-based step count after this update).
extern "C" int rg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                            hipStream_t stream) {
    RG_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "rg_adam_step: bad arguments");
    const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
    RG_REQUIRE(aligned, "rg_adam_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    rg::ProfScope prof(rg::FAM_OPTIM, stream, 0.0, 28.0 * n);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, p, g, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return rg::check_launch("rg_adam_step");
}

extern "C" int rg_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                           float weight_decay, int first_step, float grad_scale, hipStream_t stream) {
    RG_REQUIRE(p && g && n > 0 && (momentum == 0.f || momentum_buf), "rg_sgd_step: bad arguments");
    rg::ProfScope prof(rg::FAM_OPTIM, stream, 0.0, 20.0 * n);
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, momentum_buf, n, lr, momentum,
                       weight_decay, first_step, grad_scale);
    return rg::check_launch("rg_sgd_step");
}

// ---- device-side clocks: the forms a captured hipGraph can replay (no per-step values in the kernel arguments) ---------
extern "C" int rg_adam_advance(double* state, float beta1, float beta2, hipStream_t stream) {
    RG_REQUIRE(state, "rg_adam_advance: null state");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, stream, state, (double)beta1, (double)beta2);
    return rg::check_launch("rg_adam_advance");
}

// Adam update whose bias corrections come from the device clock {step, beta1^step, beta2^step} (rg_adam_advance)
extern "C" int rg_adam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, const double* state, float grad_scale,
                                hipStream_t stream) {
    RG_REQUIRE(p && g && exp_avg && exp_avg_sq && state && n > 0, "rg_adam_step_dev: bad arguments");
    rg::ProfScope prof(rg::FAM_OPTIM, stream, 0.0, 28.0 * n);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                       eps, weight_decay, state, grad_scale);
    return rg::check_launch("rg_adam_step_dev");
}

extern "C" int rg_u64_add(unsigned long long* p, unsigned long long v, hipStream_t stream) {
    RG_REQUIRE(p, "rg_u64_add: null pointer");
    hipLaunchKernelGGL(u64_add_kernel, dim3(1), dim3(64), 0, stream, p, v);
    return rg::check_launch("rg_u64_add");
}
