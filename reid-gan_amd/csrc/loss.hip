// Loss kernels (fused forward reductions, deterministic two-stage sums, no atomics).
//   GANLoss = sigmoid + binary_cross_entropy against a constant target   FD/fdgan/losses.py:29-32
//   L1 reconstruction / same-pose L1 over the same-identity pairs        FD/fdgan/model.py:190-194
//   cross_entropy on the verification score and on the cluster logits    FD/fdgan/model.py:189, CC/.../cm.py:134-135
//   lsgan MSE against a constant label                                   CC/dual_gan/models/external_function.py:53-57
#include "rg_common.h"

namespace {

constexpr int kMaxPartials = 1024;

static unsigned partial_grid(int64_t n) {
    int64_t g = rg::cdiv64(n, 256 * 8);
    if (g > kMaxPartials) g = kMaxPartials;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// torch.binary_cross_entropy clamps both logs at -100
__device__ __forceinline__ float bce_term(float x, float t) {
    const float s = sigmoidf(x);
    const float l1 = fmaxf(logf(s), -100.f);
    const float l0 = fmaxf(logf(1.f - s), -100.f);
    return -(t * l1 + (1.f - t) * l0);
}

__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                          int64_t n, float target) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += bce_term(x[i], target);
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void finalize_sum_kernel(const float* __restrict__ part, int nparts,
                                                           float* __restrict__ out, float scale) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// d/dx of mean(bce(sigmoid(x), t)) following torch's two backward formulas
// (bce: (s-t)/max(s(1-s),1e-12); sigmoid: s(1-s)), times the upstream scalar gradient.
__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ dx,
                               int64_t n, float target, float scale) {
    const float g = (gout ? gout[0] : 1.f) * scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = sigmoidf(x[i]);
        const float ss = s * (1.f - s);
        dx[i] = g * (s - target) / fmaxf(ss, 1e-12f) * ss;
    }
}

__global__ __launch_bounds__(256) void mse_const_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                int64_t n, float target) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = x[i] - target;
        s += d * d;
    }
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void mse_const_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                     float* __restrict__ dx, int64_t n, float target, float scale) {
    const float g = (gout ? gout[0] : 1.f) * scale * 2.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = g * (x[i] - target);
}

// mean(f(a + b * x)), f = ReLU (clamp != 0) or identity: the hinge / wgangp / generator branches of the dual_gan GANLoss
// (CC/dual_gan/models/external_function.py:58-68): hinge D real ReLU(1 - x), D fake ReLU(1 + x), G / wgangp +-mean(x)
__global__ __launch_bounds__(256) void affine_relu_mean_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                                       int64_t n, float a, float b, int clamp) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = a + b * x[i];
        s += clamp ? fmaxf(v, 0.f) : v;
    }
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void affine_relu_mean_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ dx,
                                            int64_t n, float a, float b, int clamp, float scale) {
    const float g = (gout ? gout[0] : 1.f) * scale * b;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = (!clamp || a + b * x[i] > 0.f) ? g : 0.f;
}

// L1 over rows selected by labels[row] == 1 (labels == NULL selects every row)
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const int64_t* __restrict__ labels,
                                                         float* __restrict__ part, int64_t n, int64_t inner) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (labels && labels[i / inner] != 1) continue;
        s += fabsf(a[i] - b[i]);
    }
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = sum(parts) / (selected_rows * inner);  out[1] = 1 / (selected_rows * inner) for the backward
__global__ __launch_bounds__(256) void l1_finalize_kernel(const float* __restrict__ part, int nparts,
                                                          const int64_t* __restrict__ labels, int rows, int64_t inner,
                                                          float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = rg_block_sum(s, red);
    float cnt = 0.f;
    if (labels) {
        for (int i = threadIdx.x; i < rows; i += 256) cnt += labels[i] == 1 ? 1.f : 0.f;
        cnt = rg_block_sum(cnt, red);
    } else {
        cnt = (float)rows;
    }
    if (threadIdx.x == 0) {
        const float denom = cnt * (float)inner;
        out[0] = s / denom;  // 0/0 = NaN when nothing is selected, like torch's mean of an empty tensor
        out[1] = 1.f / denom;
    }
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                              const int64_t* __restrict__ labels, const float* __restrict__ gout,
                              const float* __restrict__ inv_denom, float* __restrict__ da, float* __restrict__ db,
                              int64_t n, int64_t inner, float scale) {
    const float g = (gout ? gout[0] : 1.f) * scale * inv_denom[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (!labels || labels[i / inner] == 1) {
            const float d = a[i] - b[i];
            v = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        }
        if (da) da[i] = v;
        if (db) db[i] = -v;
    }
}

// out[r] = mean_i |a[r][i] - b[r][i]| (nn.L1Loss(reduction='none')(a, b).flatten(1).mean(-1)); one workgroup per row
__global__ __launch_bounds__(256) void l1_rows_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int64_t inner) {
    __shared__ float red[16];
    const float* ar = a + (int64_t)blockIdx.x * inner;
    const float* br = b + (int64_t)blockIdx.x * inner;
    float s = 0.f;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < inner; i += 256) s += fabsf(ar[i] - br[i]);
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s / (float)inner;
}

// da[r][i] = sign(a - b) * g[r] / inner, db = -da (either may be NULL)
__global__ void l1_rows_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                   float* __restrict__ da, float* __restrict__ db, int64_t n, int64_t inner) {
    const float inv = 1.f / (float)inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gr = g[i / inner] * inv;
        const float d = a[i] - b[i];
        const float v = d > 0.f ? gr : (d < 0.f ? -gr : 0.f);
        if (da) da[i] = v;
        if (db) db[i] = -v;
    }
}

// out[r] = mean_i (x[r][i] - c)^2 (the lsgan map of GANLoss against a constant label, per sample) and its backward
__global__ __launch_bounds__(256) void mse_const_rows_fwd_kernel(const float* __restrict__ x, float c, float* __restrict__ out,
                                                                 int64_t inner) {
    __shared__ float red[16];
    const float* xr = x + (int64_t)blockIdx.x * inner;
    float s = 0.f;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < inner; i += 256) {
        const float d = xr[i] - c;
        s += d * d;
    }
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s / (float)inner;
}

__global__ void mse_const_rows_bwd_kernel(const float* __restrict__ x, float c, const float* __restrict__ g, float* __restrict__ dx,
                                          int64_t n, int64_t inner) {
    const float k = 2.f / (float)inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = (x[i] - c) * (g[i / inner] * k);
}

// per-row cross entropy of scale*logits; one workgroup per row; also stores logsumexp for the backward
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels,
                                                             float* __restrict__ loss, float* __restrict__ lse, int K,
                                                             float scale) {
    __shared__ float red[16];
    const float* zr = z + (int64_t)blockIdx.x * K;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < K; i += 256) mx = fmaxf(mx, zr[i] * scale);
    mx = rg_block_max(mx, red);
    float s = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) s += expf(zr[i] * scale - mx);
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) {
        const float l = mx + logf(s);
        const int64_t y = labels[blockIdx.x];
        lse[blockIdx.x] = l;
        loss[blockIdx.x] = (y >= 0 && y < K) ? l - zr[y] * scale : 0.f;
    }
}

// dz[b][k] = g_b * scale * (softmax - onehot); g_b = grow[b] (per-row upstream) * gscale
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels,
                                                             const float* __restrict__ lse,
                                                             const float* __restrict__ grow, float* __restrict__ dz,
                                                             int K, float scale, float gscale) {
    const int b = blockIdx.x;
    const float* zr = z + (int64_t)b * K;
    float* dr = dz + (int64_t)b * K;
    const float g = (grow ? grow[b] : 1.f) * gscale * scale;
    const float l = lse[b];
    const int64_t y = labels[b];
    for (int i = threadIdx.x; i < K; i += 256) {
        float p = expf(zr[i] * scale - l);
        if (i == y) p -= 1.f;
        dr[i] = g * p;
    }
}

// out[0] = sum_i x[i]*w[i] * scale   (w may be NULL)
__global__ __launch_bounds__(256) void wsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ out, int64_t n, float scale) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += w ? x[i] * w[i] : x[i];
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// dx[i] = gout[0] * scale * w[i]
__global__ void wsum_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ w, float* __restrict__ dx,
                                int64_t n, float scale) {
    const float g = (gout ? gout[0] : 1.f) * scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = w ? g * w[i] : g;
}

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}


// WGAN-GP penalty rows, CC/dual_gan/models/external_function.py:100-101: per sample r with g = grads[r] + 1e-16,
// n = |g|_2: pen[r] = scale * (n - c)^2 and v[r] = d pen[r] / d grads[r] = scale * 2 (n - c) g / n.  One workgroup per row.
__global__ __launch_bounds__(256) void grad_penalty_rows_kernel(const float* __restrict__ g, float* __restrict__ pen,
                                                                float* __restrict__ v, int D, float c, float scale) {
    __shared__ float red[16];
    const float* gr = g + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float t = gr[i] + 1e-16f;
        s += t * t;
    }
    s = rg_block_sum(s, red);
    const float n = sqrtf(s);
    if (threadIdx.x == 0) pen[blockIdx.x] = scale * (n - c) * (n - c);
    const float k = n > 0.f ? scale * 2.f * (n - c) / n : 0.f;
    float* vr = v + (int64_t)blockIdx.x * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) vr[i] = k * (gr[i] + 1e-16f);
}

}  // namespace

extern "C" size_t rg_loss_workspace(void) { return kMaxPartials * sizeof(float); }

#define RG_NEED_WS(name)                                                               \
    if (!workspace || workspace_bytes < kMaxPartials * sizeof(float)) {                \
        rg::set_error(name ": workspace too small (need rg_loss_workspace() bytes)"); \
        return RG_ERR_WORKSPACE;                                                       \
    }

extern "C" int rg_sigmoid_bce_fwd(const float* x, float* loss, int64_t n, float target, void* workspace,
                                  size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(x && loss && n > 0, "rg_sigmoid_bce_fwd: bad arguments");
    RG_NEED_WS("rg_sigmoid_bce_fwd");
    float* part = static_cast<float*>(workspace);
    const unsigned g = partial_grid(n);
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 4.0 * n);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(g), dim3(256), 0, stream, x, part, n, target);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, stream, part, (int)g, loss, 1.f / (float)n);
    return rg::check_launch("rg_sigmoid_bce_fwd");
}

extern "C" int rg_sigmoid_bce_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float target,
                                  float grad_scale, hipStream_t stream) {
    RG_REQUIRE(x && dx && n > 0, "rg_sigmoid_bce_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, grad_out, dx, n, target,
                       grad_scale / (float)n);
    return rg::check_launch("rg_sigmoid_bce_bwd");
}

extern "C" int rg_mse_const_fwd(const float* x, float* loss, int64_t n, float target, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(x && loss && n > 0, "rg_mse_const_fwd: bad arguments");
    RG_NEED_WS("rg_mse_const_fwd");
    float* part = static_cast<float*>(workspace);
    const unsigned g = partial_grid(n);
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 4.0 * n);
    hipLaunchKernelGGL(mse_const_partial_kernel, dim3(g), dim3(256), 0, stream, x, part, n, target);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, stream, part, (int)g, loss, 1.f / (float)n);
    return rg::check_launch("rg_mse_const_fwd");
}

extern "C" int rg_mse_const_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float target,
                                float grad_scale, hipStream_t stream) {
    RG_REQUIRE(x && dx && n > 0, "rg_mse_const_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(mse_const_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, grad_out, dx, n, target,
                       grad_scale / (float)n);
    return rg::check_launch("rg_mse_const_bwd");
}

extern "C" int rg_affine_relu_mean_fwd(const float* x, float* loss, int64_t n, float a, float b, int clamp, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(x && loss && n > 0, "rg_affine_relu_mean_fwd: bad arguments");
    RG_NEED_WS("rg_affine_relu_mean_fwd");
    float* part = static_cast<float*>(workspace);
    const unsigned g = partial_grid(n);
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 4.0 * n);
    hipLaunchKernelGGL(affine_relu_mean_partial_kernel, dim3(g), dim3(256), 0, stream, x, part, n, a, b, clamp);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, stream, part, (int)g, loss, 1.f / (float)n);
    return rg::check_launch("rg_affine_relu_mean_fwd");
}

extern "C" int rg_affine_relu_mean_bwd(const float* x, const float* grad_out, float* dx, int64_t n, float a, float b,
                                       int clamp, float grad_scale, hipStream_t stream) {
    RG_REQUIRE(x && dx && n > 0, "rg_affine_relu_mean_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(affine_relu_mean_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, grad_out, dx, n, a, b, clamp,
                       grad_scale / (float)n);
    return rg::check_launch("rg_affine_relu_mean_bwd");
}

// out[0] = mean |a-b| over the selected rows, out[1] = 1/(selected elements) (kept for the backward)
extern "C" int rg_l1_fwd(const float* a, const float* b, const int64_t* row_labels, float* out2, int rows,
                         int64_t inner, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(a && b && out2 && rows > 0 && inner > 0, "rg_l1_fwd: bad arguments");
    RG_NEED_WS("rg_l1_fwd");
    float* part = static_cast<float*>(workspace);
    const int64_t n = (int64_t)rows * inner;
    const unsigned g = partial_grid(n);
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(l1_partial_kernel, dim3(g), dim3(256), 0, stream, a, b, row_labels, part, n, inner);
    hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(256), 0, stream, part, (int)g, row_labels, rows, inner, out2);
    return rg::check_launch("rg_l1_fwd");
}

extern "C" int rg_l1_bwd(const float* a, const float* b, const int64_t* row_labels, const float* grad_out,
                         const float* out2, float* da, float* db, int rows, int64_t inner, float grad_scale,
                         hipStream_t stream) {
    RG_REQUIRE(a && b && out2 && (da || db) && rows > 0 && inner > 0, "rg_l1_bwd: bad arguments");
    const int64_t n = (int64_t)rows * inner;
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 16.0 * n);
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, row_labels, grad_out, out2 + 1, da,
                       db, n, inner, grad_scale);
    return rg::check_launch("rg_l1_bwd");
}

extern "C" int rg_l1_rows_fwd(const float* a, const float* b, float* out, int rows, int64_t inner, hipStream_t stream) {
    RG_REQUIRE(a && b && out && rows > 0 && inner > 0, "rg_l1_rows_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * rows * (double)inner);
    hipLaunchKernelGGL(l1_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, a, b, out, inner);
    return rg::check_launch("rg_l1_rows_fwd");
}

extern "C" int rg_l1_rows_bwd(const float* a, const float* b, const float* grad_rows, float* da, float* db, int rows, int64_t inner,
                              hipStream_t stream) {
    RG_REQUIRE(a && b && grad_rows && (da || db) && rows > 0 && inner > 0, "rg_l1_rows_bwd: bad arguments");
    const int64_t n = (int64_t)rows * inner;
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 16.0 * n);
    hipLaunchKernelGGL(l1_rows_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, grad_rows, da, db, n, inner);
    return rg::check_launch("rg_l1_rows_bwd");
}

extern "C" int rg_mse_const_rows_fwd(const float* x, float c, float* out, int rows, int64_t inner, hipStream_t stream) {
    RG_REQUIRE(x && out && rows > 0 && inner > 0, "rg_mse_const_rows_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 4.0 * rows * (double)inner);
    hipLaunchKernelGGL(mse_const_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, x, c, out, inner);
    return rg::check_launch("rg_mse_const_rows_fwd");
}

extern "C" int rg_mse_const_rows_bwd(const float* x, float c, const float* grad_rows, float* dx, int rows, int64_t inner,
                                     hipStream_t stream) {
    RG_REQUIRE(x && grad_rows && dx && rows > 0 && inner > 0, "rg_mse_const_rows_bwd: bad arguments");
    const int64_t n = (int64_t)rows * inner;
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(mse_const_rows_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, c, grad_rows, dx, n, inner);
    return rg::check_launch("rg_mse_const_rows_bwd");
}

extern "C" int rg_softmax_ce_fwd(const float* logits, const int64_t* labels, float* loss_rows, float* lse, int B, int K,
                                 float scale, hipStream_t stream) {
    RG_REQUIRE(logits && labels && loss_rows && lse && B > 0 && K > 0, "rg_softmax_ce_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 4.0 * B * (double)K);
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(B), dim3(256), 0, stream, logits, labels, loss_rows, lse, K, scale);
    return rg::check_launch("rg_softmax_ce_fwd");
}

extern "C" int rg_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_rows,
                                 float* dlogits, int B, int K, float scale, float grad_scale, hipStream_t stream) {
    RG_REQUIRE(logits && labels && lse && dlogits && B > 0 && K > 0, "rg_softmax_ce_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 8.0 * B * (double)K);
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3(B), dim3(256), 0, stream, logits, labels, lse, grad_rows, dlogits, K,
                       scale, grad_scale);
    return rg::check_launch("rg_softmax_ce_bwd");
}

extern "C" int rg_weighted_sum_fwd(const float* x, const float* w, float* out, int64_t n, float scale,
                                   hipStream_t stream) {
    RG_REQUIRE(x && out && n > 0, "rg_weighted_sum_fwd: bad arguments");
    hipLaunchKernelGGL(wsum_kernel, dim3(1), dim3(256), 0, stream, x, w, out, n, scale);
    return rg::check_launch("rg_weighted_sum_fwd");
}

extern "C" int rg_weighted_sum_bwd(const float* grad_out, const float* w, float* dx, int64_t n, float scale,
                                   hipStream_t stream) {
    RG_REQUIRE(dx && n > 0, "rg_weighted_sum_bwd: bad arguments");
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, grad_out, w, dx, n, scale);
    return rg::check_launch("rg_weighted_sum_bwd");
}

extern "C" int rg_grad_penalty_rows(const float* grads, float* pen, float* v, int rows, int D, float constant, float scale,
                                    hipStream_t stream) {
    RG_REQUIRE(grads && pen && v && rows > 0 && D > 0, "rg_grad_penalty_rows: bad arguments");
    rg::ProfScope prof(rg::FAM_LOSS, stream, 0.0, 12.0 * rows * D);
    hipLaunchKernelGGL(grad_penalty_rows_kernel, dim3(rows), dim3(256), 0, stream, grads, pen, v, D, constant, scale);
    return rg::check_launch("rg_grad_penalty_rows");
}
