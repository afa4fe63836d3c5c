// ClusterMemory momentum update (the Python loop of CM.backward / CM_Hard.backward,
// CC/clustercontrast/models/cm.py:29-31 and :57-70) as ONE launch.
//
// The reference walks the batch in order: features[y] = m*features[y] + (1-m)*x; features[y] /= ||features[y]||.
// Samples with the same label therefore chain on the same centroid row, in batch order, while
// different labels are independent.  Workgroup b owns label targets[b] iff b is the FIRST batch
// position with that label; it then replays every later sample of that label in order on its
// row, keeping the row in registers (D/256 values per thread) and using a wave-shuffle + LDS
// reduction for each norm.  No atomics, bit-reproducible, independent of dispatch order.
//
// The (B x D)·(D x K) logits GEMM and its dgrad run on the MFMA implicit-GEMM kernels
// (conv_igemm.hip with 1x1 geometry); see the host code.
#include "rg_common.h"

namespace {

constexpr int kMaxPerThread = 16;  // D <= 256*16 = 4096

__global__ __launch_bounds__(256) void cm_update_kernel(const float* __restrict__ inputs,
                                                        const int64_t* __restrict__ targets,
                                                        float* __restrict__ features, int B, int D, int K,
                                                        float momentum, int normalize_eps) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int64_t y = targets[b];
    if (y < 0 || y >= K) return;
    for (int j = 0; j < b; ++j)
        if (targets[j] == y) return;  // an earlier workgroup owns this centroid (uniform branch)

    float* row = features + y * (int64_t)D;
    float f[kMaxPerThread];
#pragma unroll
    for (int i = 0; i < kMaxPerThread; ++i) {
        const int d = threadIdx.x + 256 * i;
        f[i] = d < D ? row[d] : 0.f;
    }
    for (int j = b; j < B; ++j) {
        if (targets[j] != y) continue;
        const float* x = inputs + (int64_t)j * D;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxPerThread; ++i) {
            const int d = threadIdx.x + 256 * i;
            if (d < D) {
                f[i] = momentum * f[i] + (1.f - momentum) * x[d];
                ss += f[i] * f[i];
            }
        }
        ss = rg_block_sum(ss, red);
        float nr = sqrtf(ss);
        if (normalize_eps) nr = fmaxf(nr, 1e-12f);  // F.normalize flavour used for gan_features (cm.py:103)
        const float inv = 1.f / nr;
#pragma unroll
        for (int i = 0; i < kMaxPerThread; ++i) f[i] *= inv;
    }
#pragma unroll
    for (int i = 0; i < kMaxPerThread; ++i) {
        const int d = threadIdx.x + 256 * i;
        if (d < D) row[d] = f[i];
    }
}

// CM_Hard: per distinct label, pick the batch sample of that label with the smallest dot product
// with the (pre-update) centroid — first minimum in batch order, as np.argmin — and apply one update.
__global__ __launch_bounds__(256) void cm_update_hard_kernel(const float* __restrict__ inputs,
                                                             const int64_t* __restrict__ targets,
                                                             float* __restrict__ features, int B, int D, int K,
                                                             float momentum) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const int64_t y = targets[b];
    if (y < 0 || y >= K) return;
    for (int j = 0; j < b; ++j)
        if (targets[j] == y) return;

    float* row = features + y * (int64_t)D;
    float f[kMaxPerThread];
#pragma unroll
    for (int i = 0; i < kMaxPerThread; ++i) {
        const int d = threadIdx.x + 256 * i;
        f[i] = d < D ? row[d] : 0.f;
    }
    float best = INFINITY;
    int best_j = b;
    for (int j = b; j < B; ++j) {
        if (targets[j] != y) continue;
        const float* x = inputs + (int64_t)j * D;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxPerThread; ++i) {
            const int d = threadIdx.x + 256 * i;
            if (d < D) dot += f[i] * x[d];
        }
        dot = rg_block_sum(dot, red);
        if (dot < best) {
            best = dot;
            best_j = j;
        }
    }
    const float* x = inputs + (int64_t)best_j * D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxPerThread; ++i) {
        const int d = threadIdx.x + 256 * i;
        if (d < D) {
            f[i] = f[i] * momentum + (1.f - momentum) * x[d];
            ss += f[i] * f[i];
        }
    }
    ss = rg_block_sum(ss, red);
    const float inv = 1.f / sqrtf(ss);
#pragma unroll
    for (int i = 0; i < kMaxPerThread; ++i) {
        const int d = threadIdx.x + 256 * i;
        if (d < D) row[d] = f[i] * inv;
    }
}

}  // namespace

extern "C" int rg_cm_update(const float* inputs, const int64_t* targets, float* features, int B, int D, int K,
                            float momentum, int normalize_eps, hipStream_t stream) {
    RG_REQUIRE(inputs && targets && features && B > 0 && D > 0 && K > 0, "rg_cm_update: bad arguments");
    RG_REQUIRE(D <= 256 * kMaxPerThread, "rg_cm_update: feature dim %d > %d", D, 256 * kMaxPerThread);
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 12.0 * B * (double)D);
    hipLaunchKernelGGL(cm_update_kernel, dim3(B), dim3(256), 0, stream, inputs, targets, features, B, D, K, momentum,
                       normalize_eps);
    return rg::check_launch("rg_cm_update");
}

extern "C" int rg_cm_update_hard(const float* inputs, const int64_t* targets, float* features, int B, int D, int K,
                                 float momentum, hipStream_t stream) {
    RG_REQUIRE(inputs && targets && features && B > 0 && D > 0 && K > 0, "rg_cm_update_hard: bad arguments");
    RG_REQUIRE(D <= 256 * kMaxPerThread, "rg_cm_update_hard: feature dim %d > %d", D, 256 * kMaxPerThread);
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 12.0 * B * (double)D);
    hipLaunchKernelGGL(cm_update_hard_kernel, dim3(B), dim3(256), 0, stream, inputs, targets, features, B, D, K,
                       momentum);
    return rg::check_launch("rg_cm_update_hard");
}

// ClusterMemory_Gradient.update_clusters (CC/clustercontrast/models/cm.py:184-190): g[id] /= |g[id]| + eps for every listed
// row, one wave per listed row.  A row listed twice is divided twice on the reference's sequential loop; the second
// division is by 1 + eps up to rounding, so rows are de-duplicated here by letting only the first occurrence act.
namespace {
__global__ __launch_bounds__(256) void normalize_listed_rows_kernel(float* __restrict__ g, const long long* __restrict__ ids,
                                                                    int n_ids, int rows, int D, float eps) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n_ids) return;
    const long long id = ids[j];
    if (id < 0 || id >= rows) return;
    for (int t = 0; t < j; ++t)
        if (ids[t] == id) return;                    // wave-uniform
    float* r = g + id * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += r[d] * r[d];
    s = rg_wave_sum(s);
    const float inv = 1.f / (sqrtf(s) + eps);
    for (int d = lane; d < D; d += 64) r[d] *= inv;
}
}  // namespace

extern "C" int rg_normalize_listed_rows(float* g, const void* ids, int n_ids, int rows, int D, float eps, hipStream_t stream) {
    RG_REQUIRE(g && ids && n_ids > 0 && rows > 0 && D > 0, "rg_normalize_listed_rows: bad arguments");
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 8.0 * n_ids * (double)D);
    hipLaunchKernelGGL(normalize_listed_rows_kernel, dim3(rg::cdiv(n_ids, 4)), dim3(256), 0, stream, g,
                       static_cast<const long long*>(ids), n_ids, rows, D, eps);
    return rg::check_launch("rg_normalize_listed_rows");
}
