// Bicubic resize fused with the per-channel (x - mean) / std normalisation (HBM-bound).
//
//   my_resize / my_normalize / my_transform     CC/clustercontrast/utils/data/diff_augs.py:6-16
//   (torchvision resize of a float tensor = F.interpolate(mode='bicubic', align_corners=False), cubic A = -0.75,
//    border taps clamped; no antialias because the path only up-samples 128x64 -> 256x128)
//
// Forward: one thread per output pixel, 16 taps from L2/LDS-free gathers (the 2x up-sample re-reads each input
// pixel 64 times, all from cache: the input is 1/4 of the output).  Backward: one thread per INPUT pixel gathers
// every output pixel whose (clamped) taps touch it — deterministic, no atomics.
#include "rg_common.h"

namespace {

#define RG_CUBIC_A (-0.75f)

__device__ __forceinline__ float cubic1(float x) { return ((RG_CUBIC_A + 2.f) * x - (RG_CUBIC_A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) {
    return ((RG_CUBIC_A * x - 5.f * RG_CUBIC_A) * x + 8.f * RG_CUBIC_A) * x - 4.f * RG_CUBIC_A;
}
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    c[0] = cubic2(t + 1.f);
    c[1] = cubic1(t);
    c[2] = cubic1(1.f - t);
    c[3] = cubic2(2.f - t);
}
// source coordinate of output index o (align_corners = False, cubic: not clamped at 0)
__device__ __forceinline__ float src_index(float scale, int o) { return scale * (o + 0.5f) - 0.5f; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void bicubic_norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int C, int H, int W, int OH, int OW, float sh, float sw,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ stdv, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        const int64_t r = i / OW;
        const int oy = (int)(r % OH);
        const int64_t nc = r / OH;
        const int c = (int)(nc % C);
        const float* src = x + nc * (int64_t)H * W;
        float v;
        if (OH == H && OW == W) {
            v = src[(int64_t)oy * W + ox];
        } else {
            const float ry = src_index(sh, oy), rx = src_index(sw, ox);
            const float fy = floorf(ry), fx = floorf(rx);
            const int iy = (int)fy, ix = (int)fx;
            float cy[4], cx[4];
            cubic_coeffs(ry - fy, cy);
            cubic_coeffs(rx - fx, cx);
            v = 0.f;
            float rows[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float* row = src + (int64_t)clampi(iy - 1 + a, 0, H - 1) * W;
                rows[a] = row[clampi(ix - 1, 0, W - 1)] * cx[0] + row[clampi(ix, 0, W - 1)] * cx[1] +
                          row[clampi(ix + 1, 0, W - 1)] * cx[2] + row[clampi(ix + 2, 0, W - 1)] * cx[3];
            }
            v = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
        }
        if (mean) v = (v - mean[c]) / stdv[c];
        y[i] = v;
    }
}

// weight with which output index o reads input index `in` along one axis (sum over clamped taps)
__device__ __forceinline__ float axis_weight(float scale, int o, int in, int len) {
    const float r = src_index(scale, o);
    const float f = floorf(r);
    const int i0 = (int)f;
    float c[4];
    cubic_coeffs(r - f, c);
    float w = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) w += (clampi(i0 - 1 + a, 0, len - 1) == in) ? c[a] : 0.f;
    return w;
}

__global__ __launch_bounds__(256) void bicubic_norm_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                               int C, int H, int W, int OH, int OW, float sh, float sw,
                                                               float ish, float isw, const float* __restrict__ stdv,
                                                               int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W);
        const int64_t r = i / W;
        const int iy = (int)(r % H);
        const int64_t nc = r / H;
        const int c = (int)(nc % C);
        const float* g = dy + nc * (int64_t)OH * OW;
        float acc;
        if (OH == H && OW == W) {
            acc = g[(int64_t)iy * OW + ix];
        } else {
            // outputs whose unclamped tap window [i0-1, i0+2] can reach iy: i0 in [iy-2, iy+1]; border pixels also
            // collect the clamped out-of-range taps, which the same window covers.  +-1 margin for rounding.
            int y0 = (int)floorf((iy - 2 + 0.5f) * ish - 0.5f) - 1, y1 = (int)ceilf((iy + 2 + 0.5f) * ish - 0.5f) + 1;
            int x0 = (int)floorf((ix - 2 + 0.5f) * isw - 0.5f) - 1, x1 = (int)ceilf((ix + 2 + 0.5f) * isw - 0.5f) + 1;
            if (iy == 0) y0 = 0;
            if (iy == H - 1) y1 = OH - 1;
            if (ix == 0) x0 = 0;
            if (ix == W - 1) x1 = OW - 1;
            y0 = clampi(y0, 0, OH - 1);
            y1 = clampi(y1, 0, OH - 1);
            x0 = clampi(x0, 0, OW - 1);
            x1 = clampi(x1, 0, OW - 1);
            acc = 0.f;
            for (int oy = y0; oy <= y1; ++oy) {
                const float wy = axis_weight(sh, oy, iy, H);
                if (wy == 0.f) continue;
                float rowacc = 0.f;
                for (int ox = x0; ox <= x1; ++ox) {
                    const float wx = axis_weight(sw, ox, ix, W);
                    rowacc += wx * g[(int64_t)oy * OW + ox];
                }
                acc += wy * rowacc;
            }
        }
        if (stdv) acc /= stdv[c];
        dx[i] = acc;
    }
}

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int rg_bicubic_normalize_fwd(const float* x, float* y, int N, int C, int H, int W, int OH, int OW,
                                        const float* mean, const float* stdv, hipStream_t stream) {
    RG_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "rg_bicubic_normalize_fwd: bad arguments");
    RG_REQUIRE((mean == nullptr) == (stdv == nullptr), "rg_bicubic_normalize_fwd: mean and std go together");
    const int64_t total = (int64_t)N * C * OH * OW;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 4.0 * (total + (double)N * C * H * W));
    hipLaunchKernelGGL(bicubic_norm_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, C, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW, mean, stdv, total);
    return rg::check_launch("rg_bicubic_normalize_fwd");
}

extern "C" int rg_bicubic_normalize_bwd(const float* dy, float* dx, int N, int C, int H, int W, int OH, int OW,
                                        const float* stdv, hipStream_t stream) {
    RG_REQUIRE(dy && dx && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "rg_bicubic_normalize_bwd: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 4.0 * (total + (double)N * C * OH * OW));
    hipLaunchKernelGGL(bicubic_norm_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, C, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW, (float)OH / (float)H, (float)OW / (float)W, stdv,
                       total);
    return rg::check_launch("rg_bicubic_normalize_bwd");
}
