// Pooling kernels for the ResNet trunks (HBM-bound).
//   max-pool 3x3/2 pad 1 after the stem           CC/clustercontrast/models/resnet_ibn_a.py:121 (torchvision layout)
//   global average pool (F.avg_pool2d full map)   FD/reid/models/resnet.py:71-72
//   GeM pooling, learnable p                      CC/clustercontrast/models/pooling.py:57-103
#include "rg_common.h"

namespace {

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// y[n,c,p,q] = max over the window; arg = window-local index (r*KW+s) of the FIRST maximum in scan
// order (torch semantics), NaN propagates like torch (a NaN wins).
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ arg,
                                   int64_t total, int H, int W, int P, int Q, int KH, int KW, int SH, int SW, int PH,
                                   int PW) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        const int64_t t = i / Q;
        const int p = (int)(t % P);
        const int64_t plane = t / P;
        const float* xp = x + plane * H * W;
        const int h0 = p * SH - PH, w0 = q * SW - PW;
        float best = -INFINITY;
        int bi = -1;
        for (int r = 0; r < KH; ++r) {
            const int h = h0 + r;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int s = 0; s < KW; ++s) {
                const int w = w0 + s;
                if ((unsigned)w >= (unsigned)W) continue;
                const float v = xp[h * W + w];
                if (bi < 0) bi = r * KW + s;  // torch starts at the first in-bounds element
                if (v > best || (v != v)) {
                    best = v;
                    bi = r * KW + s;
                }
            }
        }
        y[i] = best;
        arg[i] = (unsigned char)bi;
    }
}

// gather form (no atomics, deterministic): each input pixel sums dy of the windows that selected it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                          float* __restrict__ dx, int H, int W, int P, int Q, int KH, int KW,
                                                          int SH, int SW, int PH, int PW) {
    // grid: (pixel blocks of one plane, planes) — 32-bit index math only
    const int plane = blockIdx.y;
    const int HW = H * W;
    const float* gp = dy + (int64_t)plane * P * Q;
    const unsigned char* ap = arg + (int64_t)plane * P * Q;
    float* dxp = dx + (int64_t)plane * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int h = i / W, w = i - h * W;
        float acc = 0.f;
        // windows p with p*SH - PH <= h <= p*SH - PH + KH - 1
        int p_lo = (h + PH - KH + 1 + SH - 1);
        p_lo = p_lo > 0 ? p_lo / SH : 0;
        int p_hi = (h + PH) / SH;
        if (p_hi > P - 1) p_hi = P - 1;
        int q_lo = (w + PW - KW + 1 + SW - 1);
        q_lo = q_lo > 0 ? q_lo / SW : 0;
        int q_hi = (w + PW) / SW;
        if (q_hi > Q - 1) q_hi = Q - 1;
        for (int p = p_lo; p <= p_hi; ++p) {
            const int r = h - (p * SH - PH);
            for (int q = q_lo; q <= q_hi; ++q) {
                const int s = w - (q * SW - PW);
                if (ap[p * Q + q] == (unsigned char)(r * KW + s)) acc += gp[p * Q + q];
            }
        }
        dxp[i] = acc;
    }
}

// The ResNet stem's pool (3x3 / 2, pad 1; W % 4 == 0): one thread per 4 consecutive input pixels of a row.  An input
// pixel with even coordinate lies in one window along that axis, an odd one in two, so the 4 pixels share the 3 windows
// q0 .. q0+2 (q0 = w0 / 2) of at most two window rows: 6 (argmax, dy) pairs are loaded once and distributed, and the row
// is written as one float4.  Same gather form and summation order (window rows ascending, then columns) as the generic
// kernel: bit-identical results.
__global__ __launch_bounds__(256) void maxpool_bwd_3x3s2_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg,
                                                                float* __restrict__ dx, int H, int W, int P, int Q) {
    const int plane = blockIdx.y;
    const float* gp = dy + (int64_t)plane * P * Q;
    const unsigned char* ap = arg + (int64_t)plane * P * Q;
    float* dxp = dx + (int64_t)plane * H * W;
    const int W4 = W >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W4; i += gridDim.x * 256) {
        const int h = i / W4, w0 = (i - h * W4) * 4;
        const int q0 = w0 >> 1;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // window rows p with 2p - 1 <= h <= 2p + 1
        const int p_lo = h >> 1;                 // h even: h/2 only;  h odd: (h-1)/2 and (h+1)/2
        const int p_hi = (h + 1) >> 1;
        for (int p = p_lo; p <= p_hi; ++p) {
            if (p >= P) break;
            const int r = h - (2 * p - 1);
            float g[3];
            int a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int q = q0 + t;
                const bool ok = q < Q;
                g[t] = ok ? gp[p * Q + q] : 0.f;
                a[t] = ok ? (int)ap[p * Q + q] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // pixel w0 + j: windows q with 2q - 1 <= w <= 2q + 1, ascending q as in the generic kernel
                const int t_lo = j >> 1, t_hi = (j + 1) >> 1;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (t < t_lo || t > t_hi) continue;
                    const int sidx = (w0 + j) - (2 * (q0 + t) - 1);
                    if (a[t] == r * 3 + sidx) acc[j] += g[t];
                }
            }
        }
        *reinterpret_cast<float4*>(dxp + h * W + w0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// one wave per (n,c) plane
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes,
                                                      int HW) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* xp = x + (int64_t)plane * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += xp[i];
    s = rg_wave_sum(s);
    if (lane == 0) y[plane] = s / (float)HW;
}

__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int HW) {
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = dy[i / HW] * inv;
}

// GeM: y = (mean(clamp(x, eps)^p))^(1/p); one wave per plane.
__global__ __launch_bounds__(256) void gem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pp,
                                                      float* __restrict__ y, int planes, int HW, float eps) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float p = pp[0];
    const float* xp = x + (int64_t)plane * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += exp2f(p * log2f(fmaxf(xp[i], eps)));      // xc^p, xc >= eps > 0
    s = rg_wave_sum(s);
    if (lane == 0) y[plane] = powf(s / (float)HW, 1.f / p);
}

// dx = dy * m^(1/p - 1) * xc^(p-1) / HW * [x >= eps]
// dp_part[plane] = dy * y * ( -log(m)/p^2 + (1/p) * mean(xc^p log xc) / m )
__global__ __launch_bounds__(256) void gem_bwd_kernel(const float* __restrict__ x, const float* __restrict__ pp,
                                                      const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, float* __restrict__ dp_part, int planes,
                                                      int HW, float eps) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float p = pp[0];
    const float* xp = x + (int64_t)plane * HW;
    float* dp_ = dx + (int64_t)plane * HW;
    const float yv = y[plane], g = dy[plane];
    const float m = powf(yv, p);  // mean(xc^p)
    const float coef = g * yv / (m * (float)HW);  // g * m^(1/p-1) / HW
    float sl = 0.f;
    for (int i = lane; i < HW; i += 64) {
        const float xv = xp[i];
        const float xc = fmaxf(xv, eps);
        const float l2 = log2f(xc);
        const float xpw = exp2f((p - 1.f) * l2);                 // xc^(p-1)
        dp_[i] = xv >= eps ? coef * xpw : 0.f;
        sl += xpw * xc * (l2 * 0.6931471805599453f);
    }
    sl = rg_wave_sum(sl);
    if (lane == 0 && dp_part) {
        const float mean_l = sl / (float)HW;
        dp_part[plane] = g * yv * (-logf(m) / (p * p) + mean_l / (p * m));
    }
}

__global__ __launch_bounds__(1024) void sum_all_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
    __shared__ float red[16];
    float s = 0.f;
    const int64_t nv = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (n >> 2) : 0;
    for (int64_t i = threadIdx.x; i < nv; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += 1024) s += x[i];
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace

extern "C" int rg_maxpool2d_fwd(const float* x, float* y, unsigned char* argmax, int N, int C, int H, int W, int KH,
                                int KW, int SH, int SW, int PH, int PW, int P, int Q, hipStream_t stream) {
    RG_REQUIRE(x && y && argmax && N > 0 && C > 0 && H > 0 && W > 0 && P > 0 && Q > 0, "rg_maxpool2d_fwd: bad arguments");
    RG_REQUIRE(KH * KW <= 255 && PH < KH && PW < KW, "rg_maxpool2d_fwd: unsupported window");
    const int64_t total = (int64_t)N * C * P * Q;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * N * C * ((double)H * W + 1.25 * P * Q));
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, argmax, total, H, W, P, Q,
                       KH, KW, SH, SW, PH, PW);
    return rg::check_launch("rg_maxpool2d_fwd");
}

extern "C" int rg_maxpool2d_bwd(const float* dy, const unsigned char* argmax, float* dx, int N, int C, int H, int W,
                                int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q, hipStream_t stream) {
    RG_REQUIRE(dy && dx && argmax && N > 0 && C > 0, "rg_maxpool2d_bwd: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * N * C * ((double)H * W + 1.25 * P * Q));
    RG_REQUIRE((int64_t)N * C <= 65535 * 1024ll, "rg_maxpool2d_bwd: too many planes");
    int gx = rg::cdiv(H * W, 256);
    if (gx > 64) gx = 64;
    int64_t planes = (int64_t)N * C;
    // grid.y is limited to 65535: fold the excess into z
    const int gy = planes > 65535 ? 65535 : (int)planes;
    RG_REQUIRE(planes <= 65535, "rg_maxpool2d_bwd: N*C > 65535 planes");
    if (KH == 3 && KW == 3 && SH == 2 && SW == 2 && PH == 1 && PW == 1 && (W & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(dx) & 15) == 0) {
        int g4 = rg::cdiv(H * (W >> 2), 256);
        if (g4 > 64) g4 = 64;
        hipLaunchKernelGGL(maxpool_bwd_3x3s2_kernel, dim3(g4, gy), dim3(256), 0, stream, dy, argmax, dx, H, W, P, Q);
        return rg::check_launch("rg_maxpool2d_bwd");
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(gx, gy), dim3(256), 0, stream, dy, argmax, dx, H, W, P, Q, KH, KW, SH, SW,
                       PH, PW);
    return rg::check_launch("rg_maxpool2d_bwd");
}

extern "C" int rg_global_avgpool_fwd(const float* x, float* y, int N, int C, int HW, hipStream_t stream) {
    RG_REQUIRE(x && y && N > 0 && C > 0 && HW > 0, "rg_global_avgpool_fwd: bad arguments");
    const int planes = N * C;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * planes * (double)HW);
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(rg::cdiv(planes, 4)), dim3(256), 0, stream, x, y, planes, HW);
    return rg::check_launch("rg_global_avgpool_fwd");
}

extern "C" int rg_global_avgpool_bwd(const float* dy, float* dx, int N, int C, int HW, hipStream_t stream) {
    RG_REQUIRE(dy && dx && N > 0 && C > 0 && HW > 0, "rg_global_avgpool_bwd: bad arguments");
    const int64_t total = (int64_t)N * C * HW;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * total);
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, total, HW);
    return rg::check_launch("rg_global_avgpool_bwd");
}

extern "C" int rg_gem_pool_fwd(const float* x, const float* p, float* y, int N, int C, int HW, float eps,
                               hipStream_t stream) {
    RG_REQUIRE(x && p && y && N > 0 && C > 0 && HW > 0, "rg_gem_pool_fwd: bad arguments");
    const int planes = N * C;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * planes * (double)HW);
    hipLaunchKernelGGL(gem_fwd_kernel, dim3(rg::cdiv(planes, 4)), dim3(256), 0, stream, x, p, y, planes, HW, eps);
    return rg::check_launch("rg_gem_pool_fwd");
}

// workspace: N*C floats (per-plane dp partials).  dp (1 element) may be NULL.
extern "C" int rg_gem_pool_bwd(const float* x, const float* p, const float* y, const float* dy, float* dx, float* dp,
                               int N, int C, int HW, float eps, void* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
    RG_REQUIRE(x && p && y && dy && dx && N > 0 && C > 0 && HW > 0, "rg_gem_pool_bwd: bad arguments");
    const int planes = N * C;
    float* part = nullptr;
    if (dp) {
        if (!workspace || workspace_bytes < (size_t)planes * sizeof(float)) {
            rg::set_error("rg_gem_pool_bwd: workspace too small");
            return RG_ERR_WORKSPACE;
        }
        part = static_cast<float*>(workspace);
    }
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 8.0 * planes * (double)HW);
    hipLaunchKernelGGL(gem_bwd_kernel, dim3(rg::cdiv(planes, 4)), dim3(256), 0, stream, x, p, y, dy, dx, part, planes, HW,
                       eps);
    if (dp) hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(1024), 0, stream, part, dp, (int64_t)planes);
    return rg::check_launch("rg_gem_pool_bwd");
}
