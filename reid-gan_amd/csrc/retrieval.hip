// Per-epoch pseudo-labelling front half and evaluation distances, HBM/L2-bound helpers around the
// MFMA GEMM (which is the 1x1 convolution kernel):
//   brute-force inner-product kNN      CC/clustercontrast/utils/infomap_cluster.py:51-78 (faiss IndexFlatIP.search)
//   pairwise squared distances         CC/clustercontrast/evaluators.py:71-88, FD/reid/evaluators.py:76-98
//   cluster centroid means             CC/examples/cluster_contrast_gan_train_usl_infomap.py:332-348
#include "rg_common.h"

namespace {

// Top-k of every row of s[rows][cols] in the order (value descending, index ascending): k selection passes over the
// row (<= 64 KB, L2-resident), one workgroup per row.  Deterministic, including ties.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ s, int cols, int k, int* __restrict__ idx,
                                                        float* __restrict__ val) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const float* row = s + (int64_t)blockIdx.x * cols;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float pv = INFINITY;
    int pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < cols; c += 256) {
            const float v = row[c];
            const bool after_prev = (v < pv) || (v == pv && c > pi);
            const bool better = (v > bv) || (v == bv && c < bi);
            if (after_prev && better) {
                bv = v;
                bi = c;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            rv[wid] = bv;
            ri[wid] = bi;
        }
        __syncthreads();
        bv = rv[0];
        bi = ri[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) {
                bv = rv[w];
                bi = ri[w];
            }
        __syncthreads();
        pv = bv;
        pi = bi;
        if (tid == 0) {
            idx[(int64_t)blockIdx.x * k + j] = bi == 0x7fffffff ? -1 : bi;
            val[(int64_t)blockIdx.x * k + j] = bv;
        }
    }
}

// out[r] = sum_d x[r][d]^2, one wave per row
__global__ __launch_bounds__(256) void row_sqsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int D) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + (int64_t)r * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d] * xr[d];
    s = rg_wave_sum(s);
    if (lane == 0) out[r] = s;
}

// m[r][c] = alpha * m[r][c] + a * rowv[r] + b * colv[c]   (rowv / colv may be NULL)
__global__ void add_outer_terms_kernel(float* __restrict__ m, const float* __restrict__ rowv, const float* __restrict__ colv,
                                       float alpha, float a, float b, int cols, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float v = alpha * m[i];
        if (rowv) v += a * rowv[r];
        if (colv) v += b * colv[c];
        m[i] = v;
    }
}

// out[s][:] = mean of x[order[j]][:] for j in [offsets[s], offsets[s+1]), members added in list order
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, const long long* __restrict__ order,
                                                           const long long* __restrict__ offsets, float* __restrict__ out,
                                                           int D) {
    const int s = blockIdx.x;
    const long long beg = offsets[s], end = offsets[s + 1];
    const float inv = 1.f / (float)(end - beg);
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (long long j = beg; j < end; ++j) acc += x[order[j] * D + d];
        out[(int64_t)s * D + d] = acc * inv;
    }
}

}  // namespace

extern "C" int rg_topk_rows(const float* s, int rows, int cols, int k, int* idx, float* val, hipStream_t stream) {
    RG_REQUIRE(s && idx && val && rows > 0 && cols > 0 && k > 0 && k <= cols, "rg_topk_rows: bad arguments (k must be <= cols)");
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 4.0 * rows * (double)cols * k);
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, stream, s, cols, k, idx, val);
    return rg::check_launch("rg_topk_rows");
}

extern "C" int rg_row_sqsum(const float* x, float* out, int rows, int D, hipStream_t stream) {
    RG_REQUIRE(x && out && rows > 0 && D > 0, "rg_row_sqsum: bad arguments");
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 4.0 * rows * (double)D);
    hipLaunchKernelGGL(row_sqsum_kernel, dim3(rg::cdiv(rows, 4)), dim3(256), 0, stream, x, out, rows, D);
    return rg::check_launch("rg_row_sqsum");
}

extern "C" int rg_add_outer_terms(float* m, const float* rowv, const float* colv, float alpha, float a, float b, int rows,
                                  int cols, hipStream_t stream) {
    RG_REQUIRE(m && rows > 0 && cols > 0, "rg_add_outer_terms: bad arguments");
    const int64_t total = (int64_t)rows * cols;
    int64_t g = rg::cdiv64(total, 256);
    if (g > 8192) g = 8192;
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(add_outer_terms_kernel, dim3((unsigned)g), dim3(256), 0, stream, m, rowv, colv, alpha, a, b, cols, total);
    return rg::check_launch("rg_add_outer_terms");
}

extern "C" int rg_segment_mean(const float* x, const void* order, const void* offsets, float* out, int segments, int D,
                               hipStream_t stream) {
    RG_REQUIRE(x && order && offsets && out && segments > 0 && D > 0, "rg_segment_mean: bad arguments");
    rg::ProfScope prof(rg::FAM_CM, stream, 0.0, 0.0);
    hipLaunchKernelGGL(segment_mean_kernel, dim3(segments), dim3(256), 0, stream, x, static_cast<const long long*>(order),
                       static_cast<const long long*>(offsets), out, D);
    return rg::check_launch("rg_segment_mean");
}
