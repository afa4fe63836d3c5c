// Element-wise kernels (HBM-bound, float4 streams, grid-stride capped at 4096 workgroups).
//
// Cover the stand-alone activations / dropout / tensor arithmetic of the hot path:
//   LeakyReLU(0.2) / ReLU / Tanh / Dropout of CustomPoseGenerator   FD/fdgan/networks.py:96-156
//   LeakyReLU(0.2) of NLayerDiscriminator                            FD/fdgan/networks.py:208-229
//   (x1 - x2)^2 of EltwiseSubEmbed                                   FD/reid/models/embedding.py:26-31
//   row L2-normalisation (F.normalize)                               CC/clustercontrast/models/cm.py:125, resnet.py:90-107
//   channel concat for D_pd input / G fuse                           FD/fdgan/model.py:160-161, networks.py:175
#include "rg_common.h"

namespace {

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define RG_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int act, float slope) {
    const int64_t nv = n >> 2;
    RG_GRID_STRIDE(i, nv) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = rg_apply_act(v.x, act, slope);
        v.y = rg_apply_act(v.y, act, slope);
        v.z = rg_apply_act(v.z, act, slope);
        v.w = rg_apply_act(v.w, act, slope);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    RG_GRID_STRIDE(j, n - (nv << 2)) { y[(nv << 2) + j] = rg_apply_act(x[(nv << 2) + j], act, slope); }
}

__device__ __forceinline__ float act_grad(float yv, int act, float slope) {
    switch (act) {
        case RG_ACT_RELU: return yv > 0.f ? 1.f : 0.f;
        case RG_ACT_LEAKY: return yv > 0.f ? 1.f : slope;
        case RG_ACT_TANH: return 1.f - yv * yv;
        default: return 1.f;
    }
}

// dx = dy * act'(y) expressed through the forward OUTPUT y (valid for relu / leaky(slope>0) / tanh)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               int64_t n, int act, float slope) {
    const int64_t nv = n >> 2;
    RG_GRID_STRIDE(i, nv) {
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        g.x *= act_grad(v.x, act, slope);
        g.y *= act_grad(v.y, act, slope);
        g.z *= act_grad(v.z, act, slope);
        g.w *= act_grad(v.w, act, slope);
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    RG_GRID_STRIDE(j, n - (nv << 2)) {
        const int64_t e = (nv << 2) + j;
        dx[e] = dy[e] * act_grad(y[e], act, slope);
    }
}

// y = alpha*a + beta*b   (b may be NULL)
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n,
                             float alpha, float beta) {
    const int64_t nv = n >> 2;
    RG_GRID_STRIDE(i, nv) {
        float4 v = reinterpret_cast<const float4*>(a)[i];
        v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
        if (b) {
            const float4 w = reinterpret_cast<const float4*>(b)[i];
            v.x += beta * w.x; v.y += beta * w.y; v.z += beta * w.z; v.w += beta * w.w;
        }
        reinterpret_cast<float4*>(y)[i] = v;
    }
    RG_GRID_STRIDE(j, n - (nv << 2)) {
        const int64_t e = (nv << 2) + j;
        y[e] = alpha * a[e] + (b ? beta * b[e] : 0.f);
    }
}

__global__ void sub_square_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                      int64_t n) {
    RG_GRID_STRIDE(i, n) {
        const float d = a[i] - b[i];
        y[i] = d * d;
    }
}

// da = 2*(a-b)*dy, db = -da
__global__ void sub_square_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ dy, float* __restrict__ da, float* __restrict__ db,
                                      int64_t n) {
    RG_GRID_STRIDE(i, n) {
        const float g = 2.f * (a[i] - b[i]) * dy[i];
        if (da) da[i] = g;
        if (db) db[i] = -g;
    }
}

// Counter-based dropout mask: keep iff hash(seed, element) >= p * 2^32; y = x * keep / (1-p).
// The same (seed, index) -> bit function is restated on the CPU by oracle/ so masks can be reproduced.
__device__ __forceinline__ unsigned mix32(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 32);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p,
                               unsigned long long seed, const unsigned long long* __restrict__ clock) {
    // `clock` (optional): a device-side step counter mixed into the seed, so a captured launch draws a new mask per replay
    if (clock) seed += 0xD1B54A32D192ED03ull * *clock;
    const unsigned thr = (unsigned)fminf(p * 4294967296.f, 4294967295.f);
    const float sc = 1.f / (1.f - p);
    RG_GRID_STRIDE(i, n) {
        const unsigned r = mix32(seed * 0x100000001B3ull + (unsigned long long)i);
        y[i] = r >= thr ? x[i] * sc : 0.f;
    }
}

// one workgroup per row: y = x / max(||x||_2, eps); also stores the norm
__global__ __launch_bounds__(256) void l2norm_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              float* __restrict__ norm, int D, float eps) {
    __shared__ float red[16];
    const float* xr = x + (int64_t)blockIdx.x * D;
    float* yr = y + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += xr[i] * xr[i];
    s = rg_block_sum(s, red);
    const float nr = sqrtf(s);
    const float inv = 1.f / fmaxf(nr, eps);
    for (int i = threadIdx.x; i < D; i += blockDim.x) yr[i] = xr[i] * inv;
    if (norm && threadIdx.x == 0) norm[blockIdx.x] = nr;
}

// dx = (dy - y * <dy, y>) / max(norm, eps)   (for norm > eps; for norm <= eps: dy / eps)
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              const float* __restrict__ norm, float* __restrict__ dx,
                                                              int D, float eps) {
    __shared__ float red[16];
    const float* yr = y + (int64_t)blockIdx.x * D;
    const float* gr = dy + (int64_t)blockIdx.x * D;
    float* dr = dx + (int64_t)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += yr[i] * gr[i];
    s = rg_block_sum(s, red);
    const float nr = norm[blockIdx.x];
    const float inv = 1.f / fmaxf(nr, eps);
    const float k = nr > eps ? s : 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) dr[i] = (gr[i] - yr[i] * k) * inv;
}

// The same normalisation along the CHANNEL axis of an [N][C][HW] map (F.normalize(x, dim=1) of a feature map: unit norm over the
// channels at every pixel) without the two permuted copies a row view would need.  Workgroup = 64 pixels x 4 channel groups of one
// image: lanes run along the pixels (coalesced rows of every channel), group g owns channels g, g + 4, ...; eight loads in flight.
// PX = 16 pixels x 16 channel groups when the 64-pixel form would leave most of the chip idle (a [32, 2048, 16 x 8] map is 64
// workgroups of 64 pixels: 102 us for 100 MB; 256 workgroups of 16 pixels run at the memory rate).
template <int PX>
__global__ __launch_bounds__(256) void l2norm_channels_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  float* __restrict__ norm, int C, int HW, float eps) {
    constexpr int G = 256 / PX;
    __shared__ float red[G][PX];
    const int lp = threadIdx.x % PX, g = threadIdx.x / PX;
    const int px = blockIdx.x * PX + lp, n = blockIdx.y;
    const bool ok = px < HW;
    const float* xb = x + (int64_t)n * C * HW + (ok ? px : 0);
    float s = 0.f;
#pragma unroll 8
    for (int c = g; c < C; c += G) {
        const float v = ok ? xb[(int64_t)c * HW] : 0.f;
        s += v * v;
    }
    red[g][lp] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) tot += red[i][lp];
    const float nr = sqrtf(tot);
    const float inv = 1.f / fmaxf(nr, eps);
    if (!ok) return;
    float* yb = y + (int64_t)n * C * HW + px;
#pragma unroll 8
    for (int c = g; c < C; c += G) yb[(int64_t)c * HW] = xb[(int64_t)c * HW] * inv;
    if (norm && g == 0) norm[(int64_t)n * HW + px] = nr;
}

template <int PX>
__global__ __launch_bounds__(256) void l2norm_channels_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                  const float* __restrict__ norm, float* __restrict__ dx, int C,
                                                                  int HW, float eps) {
    constexpr int G = 256 / PX;
    __shared__ float red[G][PX];
    const int lp = threadIdx.x % PX, g = threadIdx.x / PX;
    const int px = blockIdx.x * PX + lp, n = blockIdx.y;
    const bool ok = px < HW;
    const int64_t base = (int64_t)n * C * HW + (ok ? px : 0);
    float s = 0.f;
#pragma unroll 8
    for (int c = g; c < C; c += G) s += ok ? y[base + (int64_t)c * HW] * dy[base + (int64_t)c * HW] : 0.f;
    red[g][lp] = s;
    __syncthreads();
    if (!ok) return;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) dot += red[i][lp];
    const float nr = norm[(int64_t)n * HW + px];
    const float inv = 1.f / fmaxf(nr, eps);
    const float k = nr > eps ? dot : 0.f;
#pragma unroll 8
    for (int c = g; c < C; c += G) dx[base + (int64_t)c * HW] = (dy[base + (int64_t)c * HW] - y[base + (int64_t)c * HW] * k) * inv;
}

// channel-block copy between NCHW tensors: dst[n][dc0 + c][hw] = src[n][sc0 + c][hw], c < Cc
__global__ void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t total, int Cc,
                                     int HW, int Cs, int sc0, int Cd, int dc0, int accumulate) {
    RG_GRID_STRIDE(i, total) {
        const int hw = (int)(i % HW);
        const int64_t t = i / HW;
        const int c = (int)(t % Cc);
        const int64_t n = t / Cc;
        const float v = src[(n * Cs + sc0 + c) * HW + hw];
        float* d = dst + (n * Cd + dc0 + c) * HW + hw;
        *d = accumulate ? *d + v : v;
    }
}

__global__ void fill_kernel(float* __restrict__ y, int64_t n, float v) {
    RG_GRID_STRIDE(i, n) y[i] = v;
}

}  // namespace

extern "C" int rg_act_fwd(const float* x, float* y, int64_t n, int act, float slope, hipStream_t stream) {
    RG_REQUIRE(x && y && n >= 0, "rg_act_fwd: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, x, y, n, act, slope);
    return rg::check_launch("rg_act_fwd");
}

extern "C" int rg_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope,
                          hipStream_t stream) {
    RG_REQUIRE(dy && y && dx && n >= 0, "rg_act_bwd: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 12.0 * n);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, dy, y, dx, n, act, slope);
    return rg::check_launch("rg_act_bwd");
}

extern "C" int rg_axpby(const float* a, const float* b, float* y, int64_t n, float alpha, float beta,
                        hipStream_t stream) {
    RG_REQUIRE(a && y && n >= 0, "rg_axpby: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, (b ? 12.0 : 8.0) * n);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, a, b, y, n, alpha, beta);
    return rg::check_launch("rg_axpby");
}

// out[i] = a[i], out[B + i] = take_a[i] ? a[i] : b[i]  (i < B samples of `per` floats; take_a == NULL: always b).
// grid (chunks, 2B): one sample row per blockIdx.y, float4 when `per` allows.
__global__ __launch_bounds__(256) void pair_cat_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const int64_t* __restrict__ take_a, float* __restrict__ out, int B,
                                                       int64_t per) {
    const int row = blockIdx.y;
    const int i = row < B ? row : row - B;
    const float* src = (row < B || (take_a && take_a[i] != 0)) ? a + (int64_t)i * per : b + (int64_t)i * per;
    float* dst = out + (int64_t)row * per;
    if ((per & 3) == 0) {
        const int64_t nv = per >> 2;
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nv; j += (int64_t)gridDim.x * 256)
            reinterpret_cast<float4*>(dst)[j] = reinterpret_cast<const float4*>(src)[j];
    } else {
        for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < per; j += (int64_t)gridDim.x * 256) dst[j] = src[j];
    }
}

extern "C" int rg_pair_cat(const float* a, const float* b, const int64_t* take_a, float* out, int B, int64_t per,
                           hipStream_t stream) {
    RG_REQUIRE(a && b && out && B > 0 && per > 0 && 2 * (int64_t)B <= 65535, "rg_pair_cat: bad arguments");
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 16.0 * B * (double)per);
    int64_t chunks = rg::cdiv64((per & 3) ? per : per / 4, 256 * 4);
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(pair_cat_kernel, dim3((unsigned)chunks, 2 * B), dim3(256), 0, stream, a, b, take_a, out, B, per);
    return rg::check_launch("rg_pair_cat");
}

extern "C" int rg_fill(float* y, int64_t n, float v, hipStream_t stream) {
    RG_REQUIRE(y && n >= 0, "rg_fill: bad arguments");
    if (n == 0) return RG_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, n, v);
    return rg::check_launch("rg_fill");
}

// One wave that does nothing for `us` microseconds of the constant-rate wall clock (bounded loop: every lane leaves after at most
// 2^24 polls whatever the clock says).  rg_hip.ops.concurrent_stream launches it on two streams to find out whether they share a
// hardware queue: kernels of streams mapped onto the same queue run one after the other.
__global__ void spin_kernel(long long ticks, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned polls = 0;
    while (wall_clock64() - t0 < ticks && polls < (1u << 24)) ++polls;
    if (sink && polls == 0xffffffffu) *sink = polls;
}

extern "C" int rg_spin_us(int us, hipStream_t stream) {
    RG_REQUIRE(us > 0 && us <= 100000, "rg_spin_us: us must be in 1..100000");
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        khz = 100000;                               // the constant 100 MHz counter of gfx9
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, (long long)us * khz / 1000, (unsigned*)nullptr);
    return rg::check_launch("rg_spin_us");
}

extern "C" int rg_sub_square_fwd(const float* a, const float* b, float* y, int64_t n, hipStream_t stream) {
    RG_REQUIRE(a && b && y && n >= 0, "rg_sub_square_fwd: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 12.0 * n);
    hipLaunchKernelGGL(sub_square_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, y, n);
    return rg::check_launch("rg_sub_square_fwd");
}

extern "C" int rg_sub_square_bwd(const float* a, const float* b, const float* dy, float* da, float* db, int64_t n,
                                 hipStream_t stream) {
    RG_REQUIRE(a && b && dy && (da || db) && n >= 0, "rg_sub_square_bwd: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 20.0 * n);
    hipLaunchKernelGGL(sub_square_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, dy, da, db, n);
    return rg::check_launch("rg_sub_square_bwd");
}

extern "C" int rg_dropout(const float* x, float* y, int64_t n, float p, unsigned long long seed, hipStream_t stream) {
    RG_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, "rg_dropout: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n, p, seed, (const unsigned long long*)nullptr);
    return rg::check_launch("rg_dropout");
}

// the same mask function with seed + 0xD1B54A32D192ED03 * *clock: the forward and the backward launch of one step read the same
// clock value, a replayed hipGraph draws a fresh mask every time the clock is advanced (rg_u64_add)
extern "C" int rg_dropout_clocked(const float* x, float* y, int64_t n, float p, unsigned long long seed,
                                  const unsigned long long* clock, hipStream_t stream) {
    RG_REQUIRE(x && y && clock && n >= 0 && p >= 0.f && p < 1.f, "rg_dropout_clocked: bad arguments");
    if (n == 0) return RG_OK;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * n);
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n, p, seed, clock);
    return rg::check_launch("rg_dropout_clocked");
}

extern "C" int rg_l2norm_rows_fwd(const float* x, float* y, float* norm, int rows, int D, float eps,
                                  hipStream_t stream) {
    RG_REQUIRE(x && y && rows > 0 && D > 0, "rg_l2norm_rows_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * rows * (double)D);
    hipLaunchKernelGGL(l2norm_rows_fwd_kernel, dim3(rows), dim3(256), 0, stream, x, y, norm, D, eps);
    return rg::check_launch("rg_l2norm_rows_fwd");
}

extern "C" int rg_l2norm_rows_bwd(const float* y, const float* dy, const float* norm, float* dx, int rows, int D,
                                  float eps, hipStream_t stream) {
    RG_REQUIRE(y && dy && norm && dx && rows > 0 && D > 0, "rg_l2norm_rows_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 12.0 * rows * (double)D);
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(rows), dim3(256), 0, stream, y, dy, norm, dx, D, eps);
    return rg::check_launch("rg_l2norm_rows_bwd");
}

extern "C" int rg_l2norm_channels_fwd(const float* x, float* y, float* norm, int N, int C, int HW, float eps, hipStream_t stream) {
    RG_REQUIRE(x && y && N > 0 && C > 0 && HW > 0 && N <= 65535, "rg_l2norm_channels_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * N * (double)C * HW);
    if ((int64_t)rg::cdiv(HW, 64) * N >= 1024)
        hipLaunchKernelGGL(l2norm_channels_fwd_kernel<64>, dim3(rg::cdiv(HW, 64), N), dim3(256), 0, stream, x, y, norm, C, HW, eps);
    else
        hipLaunchKernelGGL(l2norm_channels_fwd_kernel<16>, dim3(rg::cdiv(HW, 16), N), dim3(256), 0, stream, x, y, norm, C, HW, eps);
    return rg::check_launch("rg_l2norm_channels_fwd");
}

extern "C" int rg_l2norm_channels_bwd(const float* y, const float* dy, const float* norm, float* dx, int N, int C, int HW, float eps,
                                      hipStream_t stream) {
    RG_REQUIRE(y && dy && norm && dx && N > 0 && C > 0 && HW > 0 && N <= 65535, "rg_l2norm_channels_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 12.0 * N * (double)C * HW);
    if ((int64_t)rg::cdiv(HW, 64) * N >= 1024)
        hipLaunchKernelGGL(l2norm_channels_bwd_kernel<64>, dim3(rg::cdiv(HW, 64), N), dim3(256), 0, stream, y, dy, norm, dx, C, HW, eps);
    else
        hipLaunchKernelGGL(l2norm_channels_bwd_kernel<16>, dim3(rg::cdiv(HW, 16), N), dim3(256), 0, stream, y, dy, norm, dx, C, HW, eps);
    return rg::check_launch("rg_l2norm_channels_bwd");
}

extern "C" int rg_copy_channels(const float* src, float* dst, int N, int Cc, int HW, int Cs, int sc0, int Cd, int dc0,
                                int accumulate, hipStream_t stream) {
    RG_REQUIRE(src && dst && N > 0 && Cc > 0 && HW > 0, "rg_copy_channels: bad arguments");
    RG_REQUIRE(sc0 >= 0 && sc0 + Cc <= Cs && dc0 >= 0 && dc0 + Cc <= Cd, "rg_copy_channels: channel range out of bounds");
    const int64_t total = (int64_t)N * Cc * HW;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, total, Cc, HW, Cs,
                       sc0, Cd, dc0, accumulate);
    return rg::check_launch("rg_copy_channels");
}

// AEModel.hard_mix, CC/dual_gan/models/AE_model.py:274-292: out[j] = lam * src[ia[j]] + (1 - lam) * src[ib[j]] over rows of
// `len` floats (feature maps of the image encoder).  The backward gathers per SOURCE row (deterministic, no atomics).
namespace {
__global__ void mix_rows_fwd_kernel(const float* __restrict__ src, const long long* __restrict__ ia,
                                    const long long* __restrict__ ib, float lam, float* __restrict__ out, int64_t len,
                                    int64_t total) {
    RG_GRID_STRIDE(i, total) {
        const int64_t j = i / len, e = i - j * len;
        out[i] = lam * src[ia[j] * len + e] + (1.f - lam) * src[ib[j] * len + e];
    }
}
__global__ void mix_rows_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ ia,
                                    const long long* __restrict__ ib, float lam, float* __restrict__ dsrc, int rows_out,
                                    int64_t len, int64_t total) {
    RG_GRID_STRIDE(i, total) {
        const int64_t r = i / len, e = i - r * len;
        float acc = 0.f;
        for (int j = 0; j < rows_out; ++j) {
            const float gv = g[(int64_t)j * len + e];
            if (ia[j] == r) acc += lam * gv;
            if (ib[j] == r) acc += (1.f - lam) * gv;
        }
        dsrc[i] = acc;
    }
}
}  // namespace

extern "C" int rg_mix_rows_fwd(const float* src, const void* idx_a, const void* idx_b, float lam, float* out, int rows_src,
                               int rows_out, int64_t len, hipStream_t stream) {
    RG_REQUIRE(src && idx_a && idx_b && out && rows_src > 0 && rows_out > 0 && len > 0, "rg_mix_rows_fwd: bad arguments");
    const int64_t total = (int64_t)rows_out * len;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 12.0 * total);
    hipLaunchKernelGGL(mix_rows_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src,
                       static_cast<const long long*>(idx_a), static_cast<const long long*>(idx_b), lam, out, len, total);
    return rg::check_launch("rg_mix_rows_fwd");
}

extern "C" int rg_mix_rows_bwd(const float* g, const void* idx_a, const void* idx_b, float lam, float* dsrc, int rows_src,
                               int rows_out, int64_t len, hipStream_t stream) {
    RG_REQUIRE(g && idx_a && idx_b && dsrc && rows_src > 0 && rows_out > 0 && len > 0, "rg_mix_rows_bwd: bad arguments");
    const int64_t total = (int64_t)rows_src * len;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 4.0 * total * (1.0 + rows_out));
    hipLaunchKernelGGL(mix_rows_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, g,
                       static_cast<const long long*>(idx_a), static_cast<const long long*>(idx_b), lam, dsrc, rows_out, len,
                       total);
    return rg::check_launch("rg_mix_rows_bwd");
}
