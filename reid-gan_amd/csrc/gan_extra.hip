// Small HBM/latency-bound kernels of the dual_gan networks (CC/dual_gan/models/base_function.py):
//   nn.AvgPool2d(2, 2) of ResBlockEncoder(Optimized) shortcuts            :372-420
//   nn.ReflectionPad2d(1) of the Output block                             :423-443
//   torch.nn.utils.spectral_norm on every discriminator conv              :121-126, networks.py:917-955
#include "rg_common.h"

namespace {

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define RG_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---- average pooling, kernel == stride == k, no padding ------------------------------------------------------
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int k, int P, int Q,
                                   int64_t total) {
    const float inv = 1.f / (float)(k * k);
    RG_GRID_STRIDE(i, total) {
        const int q = (int)(i % Q);
        const int64_t r = i / Q;
        const int pp = (int)(r % P);
        const int64_t nc = r / P;
        const float* src = x + (nc * H + (int64_t)pp * k) * W + (int64_t)q * k;
        float s = 0.f;
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) s += src[(int64_t)a * W + b];
        y[i] = s * inv;
    }
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int k, int P, int Q,
                                   int64_t total) {
    const float inv = 1.f / (float)(k * k);
    RG_GRID_STRIDE(i, total) {
        const int w = (int)(i % W);
        const int64_t r = i / W;
        const int h = (int)(r % H);
        const int64_t nc = r / H;
        const int pp = h / k, q = w / k;
        dx[i] = (pp < P && q < Q) ? dy[(nc * P + pp) * Q + q] * inv : 0.f;
    }
}

// ---- reflection padding --------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect(int t, int len) {
    if (t < 0) t = -t;
    if (t >= len) t = 2 * (len - 1) - t;
    return t;
}

// one thread per element, 32-bit index arithmetic with multiply-high divisions by the two row lengths (the 64-bit `/` and `%` of a
// generic grid-stride loop cost more than the copy itself)
struct PadDiv {
    unsigned mul, shr, d;
};
static PadDiv make_paddiv(unsigned d) {
    PadDiv f;
    f.d = d ? d : 1;
    f.mul = 0;
    f.shr = 0;
    if (f.d == 1) return f;
    unsigned l = 0;
    while ((1ull << l) < f.d) ++l;
    const unsigned p = 31 + l;
    f.mul = (unsigned)(((1ull << p) + f.d - 1) / f.d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ int pdiv(int n, const PadDiv& f) { return f.d == 1 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shr); }

__global__ __launch_bounds__(256) void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                              int pad, int OH, int OW, int total, PadDiv d_ow, PadDiv d_oh) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = pdiv(i, d_ow), ow = i - r * OW;
        const int nc = pdiv(r, d_oh), oh = r - nc * OH;
        y[i] = x[((int64_t)nc * H + reflect(oh - pad, H)) * W + reflect(ow - pad, W)];
    }
}

// gather form of the adjoint: input (h, w) collects its own copy plus the mirrored border copies
__global__ __launch_bounds__(256) void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                              int pad, int OH, int OW, int total, PadDiv d_w, PadDiv d_h) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = pdiv(i, d_w), w = i - r * W;
        const int nc = pdiv(r, d_h), h = r - nc * H;
        int hs[3], ws[3], nh = 0, nw = 0;
        hs[nh++] = h + pad;
        if (h >= 1 && h <= pad) hs[nh++] = pad - h;
        if (h <= H - 2 && h >= H - 1 - pad) hs[nh++] = pad + 2 * (H - 1) - h;
        ws[nw++] = w + pad;
        if (w >= 1 && w <= pad) ws[nw++] = pad - w;
        if (w <= W - 2 && w >= W - 1 - pad) ws[nw++] = pad + 2 * (W - 1) - w;
        const float* base = dy + (int64_t)nc * OH * OW;
        float s = 0.f;
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) s += base[hs[a] * OW + ws[b]];
        dx[i] = s;
    }
}

// The same pair for W % 4 == 0 and pad <= 3 (the Output blocks: pad 1 on 64-channel maps at full resolution, 134 MB at 64 crops),
// one float4 of an INPUT row per thread, with the element-wise activation in front of the padding folded in:
//   forward   y = pad(act(x)): the four values go to row h + pad and to the (at most one) mirrored row, the first / last float4 of a
//             row also writes the mirrored border columns — unaligned 16-byte buffer stores (dword alignment is what they need);
//   backward  dx = act'(x) * gather(dy): the same rows / columns read back in the scalar kernel's summation order.
// act'(.) is taken from the sign of x (ReLU / LeakyReLU with a positive slope: sign(act(x)) == sign(x)).
typedef __amdgpu_buffer_rsrc_t prsrc_t;
constexpr unsigned POOB = 0x80000000u;
__device__ __forceinline__ prsrc_t p_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
typedef float pf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 p_load4(prsrc_t r, unsigned off) {
    const pf4 v = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float p_load(prsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void p_store4(prsrc_t r, unsigned off, float4 v) {
    const pf4 w = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, w), r, off, 0, 0);
}
__device__ __forceinline__ void p_store(prsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}
__device__ __forceinline__ float pad_act(float v, int act, float slope) {          // uniform act: selects
    const float neg = act == RG_ACT_LEAKY ? slope : 0.f;
    return act == RG_ACT_NONE ? v : (v > 0.f ? v : v * neg);
}
__device__ __forceinline__ float pad_act_grad(float xv, int act, float slope) {
    const float neg = act == RG_ACT_LEAKY ? slope : 0.f;
    return act == RG_ACT_NONE ? 1.f : (xv > 0.f ? 1.f : neg);
}

// the (at most) three padded rows / columns input index i of an axis of length n feeds: i + pad, and its mirror images
__device__ __forceinline__ int pad_targets(int i, int n, int pad, int (&t)[3]) {
    int c = 0;
    t[c++] = i + pad;
    if (i >= 1 && i <= pad) t[c++] = pad - i;
    if (i <= n - 2 && i >= n - 1 - pad) t[c++] = pad + 2 * (n - 1) - i;
    return c;
}

__global__ __launch_bounds__(256) void reflect_pad_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                                  int pad, int OH, int OW, int total4, PadDiv d_w4, PadDiv d_h,
                                                                  int act, float slope, unsigned ybytes) {
    const prsrc_t ry = p_rsrc(y, ybytes);
    const int W4 = W >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int r = pdiv(i, d_w4), w0 = (i - r * W4) << 2;
        const int nc = pdiv(r, d_h), h = r - nc * H;
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = pad_act(v.x, act, slope); v.y = pad_act(v.y, act, slope); v.z = pad_act(v.z, act, slope); v.w = pad_act(v.w, act, slope);
        const float e[4] = {v.x, v.y, v.z, v.w};
        int rows[3];
        const int nr = pad_targets(h, H, pad, rows);
        for (int a = 0; a < nr; ++a) {
            const unsigned row = (unsigned)((nc * OH + rows[a]) * OW) * 4u;
            p_store4(ry, row + (unsigned)(w0 + pad) * 4u, v);
            if (w0 == 0)                                  // columns pad - w, w = 1 .. pad
                for (int w = 1; w <= pad; ++w) p_store(ry, row + (unsigned)(pad - w) * 4u, e[w]);
            if (w0 == W - 4)                              // columns pad + 2 (W - 1) - w, w = W - 1 - pad .. W - 2
                for (int w = W - 1 - pad; w <= W - 2; ++w) p_store(ry, row + (unsigned)(pad + 2 * (W - 1) - w) * 4u, e[w - w0]);
        }
    }
}

__global__ __launch_bounds__(256) void reflect_pad_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ xact,
                                                                  float* __restrict__ dx, int H, int W, int pad, int OH, int OW,
                                                                  int total4, PadDiv d_w4, PadDiv d_h, int act, float slope,
                                                                  unsigned dybytes) {
    const prsrc_t rd = p_rsrc(dy, dybytes);
    const int W4 = W >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int r = pdiv(i, d_w4), w0 = (i - r * W4) << 2;
        const int nc = pdiv(r, d_h), h = r - nc * H;
        int rows[3];
        const int nr = pad_targets(h, H, pad, rows);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nr; ++a) {
            const unsigned row = (unsigned)((nc * OH + rows[a]) * OW) * 4u;
            const float4 m = p_load4(rd, row + (unsigned)(w0 + pad) * 4u);
            const float mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + j;
                s[j] += mv[j];
                if (w >= 1 && w <= pad) s[j] += p_load(rd, row + (unsigned)(pad - w) * 4u);
                if (w <= W - 2 && w >= W - 1 - pad) s[j] += p_load(rd, row + (unsigned)(pad + 2 * (W - 1) - w) * 4u);
            }
        }
        float4 o = make_float4(s[0], s[1], s[2], s[3]);
        if (act != RG_ACT_NONE) {
            const float4 xv = reinterpret_cast<const float4*>(xact)[i];
            o.x *= pad_act_grad(xv.x, act, slope); o.y *= pad_act_grad(xv.y, act, slope);
            o.z *= pad_act_grad(xv.z, act, slope); o.w *= pad_act_grad(xv.w, act, slope);
        }
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

// ---- spectral norm ---------------------------------------------------------------------------------------------
// One workgroup per weight matrix W[K][M] (discriminator filters: K <= 128, M <= 2048, <= 1 MB).
// training: v <- normalize(W^T u), u <- normalize(W v) (one power iteration, eps-clamped norms, as
// torch.nn.utils.spectral_norm with n_power_iterations = 1), then sigma = u . (W v).
// sigma is written to sigma_out[0] and 1/sigma to sigma_out[1]; the scaled weight comes from the second kernel.
constexpr int SN_THREADS = 1024;
constexpr int SN_MAX_M = 12288;         // v (and W^T u) live in LDS

__device__ __forceinline__ void sn_power(const float* __restrict__ w, float* __restrict__ u, float* __restrict__ v,
                                         float* __restrict__ sigma_out, float* __restrict__ uv_saved, int K, int M, int training,
                                         float eps, float* vs, float* us, float* ss, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int k = tid; k < K; k += SN_THREADS) us[k] = u[k];
    __syncthreads();
    if (training) {
        // t = W^T u (threads along m: coalesced rows), v = t / max(|t|, eps)
        float part = 0.f;
        for (int m = tid; m < M; m += SN_THREADS) {
            float t = 0.f;
            for (int k = 0; k < K; ++k) t += w[(int64_t)k * M + m] * us[k];
            vs[m] = t;
            part += t * t;
        }
        const float nrm = sqrtf(rg_block_sum(part, red));
        const float inv = 1.f / fmaxf(nrm, eps);
        for (int m = tid; m < M; m += SN_THREADS) {
            const float t = vs[m] * inv;
            vs[m] = t;
            v[m] = t;
            if (uv_saved) uv_saved[K + m] = t;
        }
    } else {
        for (int m = tid; m < M; m += SN_THREADS) {
            vs[m] = v[m];
            if (uv_saved) uv_saved[K + m] = vs[m];
        }
    }
    __syncthreads();
    // s = W v: one wave per row
    for (int k = wid; k < K; k += SN_THREADS / 64) {
        float s = 0.f;
        for (int m = lane; m < M; m += 64) s += w[(int64_t)k * M + m] * vs[m];
        s = rg_wave_sum(s);
        if (lane == 0) ss[k] = s;
    }
    __syncthreads();
    float dotp = 0.f;
    if (training) {
        float part = 0.f;
        for (int k = tid; k < K; k += SN_THREADS) part += ss[k] * ss[k];
        const float nrm = sqrtf(rg_block_sum(part, red));
        const float inv = 1.f / fmaxf(nrm, eps);
        for (int k = tid; k < K; k += SN_THREADS) {
            const float un = ss[k] * inv;
            u[k] = un;
            if (uv_saved) uv_saved[k] = un;
            dotp += un * ss[k];
        }
    } else {
        for (int k = tid; k < K; k += SN_THREADS) {
            dotp += us[k] * ss[k];
            if (uv_saved) uv_saved[k] = us[k];
        }
    }
    const float sigma = rg_block_sum(dotp, red);
    if (tid == 0) {
        sigma_out[0] = sigma;
        sigma_out[1] = 1.f / sigma;
    }
}

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_power_kernel(const float* __restrict__ w, float* __restrict__ u,
                                                                        float* __restrict__ v, float* __restrict__ sigma_out,
                                                                        float* __restrict__ uv_saved, int K, int M, int training,
                                                                        float eps) {
    __shared__ float vs[SN_MAX_M];
    __shared__ float us[1024];
    __shared__ float ss[1024];
    __shared__ float red[32];
    sn_power(w, u, v, sigma_out, uv_saved, K, M, training, eps, vs, us, ss, red);
}

// all spectral-normed filters of one network forward in two launches: workgroup b iterates matrix b, then grid (x, b) scales it
struct SNBatch {
    rg_sn_desc d[RG_SN_MAX_BATCH];
};

__global__ __launch_bounds__(SN_THREADS) void spectral_norm_power_multi_kernel(const SNBatch batch, int training, float eps) {
    __shared__ float vs[SN_MAX_M];
    __shared__ float us[1024];
    __shared__ float ss[1024];
    __shared__ float red[32];
    const rg_sn_desc& d = batch.d[blockIdx.x];
    sn_power(d.w, d.u, d.v, d.sigma, d.uv_saved, d.K, d.M, training, eps, vs, us, ss, red);
}

__global__ __launch_bounds__(256) void spectral_norm_scale_multi_kernel(const SNBatch batch) {
    const rg_sn_desc& d = batch.d[blockIdx.y];
    const float f = d.sigma[1];
    const int64_t n = (int64_t)d.K * d.M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d.w_sn[i] = d.w[i] * f;
}

__global__ void scale_by_device_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ s,
                                              int64_t n) {
    const float f = s[0];
    RG_GRID_STRIDE(i, n) y[i] = x[i] * f;
}

// dW = (dWsn - (sum dWsn * Wsn) u v^T) / sigma      (u, v constants of the forward)
__global__ __launch_bounds__(SN_THREADS) void spectral_norm_bwd_kernel(const float* __restrict__ dwsn,
                                                                      const float* __restrict__ wsn,
                                                                      const float* __restrict__ u, const float* __restrict__ v,
                                                                      const float* __restrict__ sigma, float* __restrict__ dw,
                                                                      int K, int M, int accumulate) {
    __shared__ float red[32];
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)K * M;
    float part = 0.f;
    for (int64_t i = tid; i < n; i += SN_THREADS) part += dwsn[i] * wsn[i];
    const float c = rg_block_sum(part, red);
    const float inv = sigma[1];
    for (int64_t i = tid; i < n; i += SN_THREADS) {
        const int k = (int)(i / M), m = (int)(i - (int64_t)k * M);
        const float g = (dwsn[i] - c * u[k] * v[m]) * inv;
        dw[i] = accumulate ? dw[i] + g : g;
    }
}

}  // namespace

extern "C" int rg_avgpool2d_fwd(const float* x, float* y, int N, int C, int H, int W, int k, hipStream_t stream) {
    RG_REQUIRE(x && y && N > 0 && C > 0 && k > 0 && H >= k && W >= k, "rg_avgpool2d_fwd: bad arguments");
    const int P = H / k, Q = W / k;
    const int64_t total = (int64_t)N * C * P * Q;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * (total + (double)N * C * H * W));
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, H, W, k, P, Q, total);
    return rg::check_launch("rg_avgpool2d_fwd");
}

extern "C" int rg_avgpool2d_bwd(const float* dy, float* dx, int N, int C, int H, int W, int k, hipStream_t stream) {
    RG_REQUIRE(dy && dx && N > 0 && C > 0 && k > 0 && H >= k && W >= k, "rg_avgpool2d_bwd: bad arguments");
    const int P = H / k, Q = W / k;
    const int64_t total = (int64_t)N * C * H * W;
    rg::ProfScope prof(rg::FAM_POOL, stream, 0.0, 4.0 * (total + (double)N * C * P * Q));
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, H, W, k, P, Q, total);
    return rg::check_launch("rg_avgpool2d_bwd");
}

// y = pad(act(x)); act: RG_ACT_NONE / RELU / LEAKY (the activation the Output block applies in front of its padding)
extern "C" int rg_reflection_pad2d_fwd(const float* x, float* y, int N, int C, int H, int W, int pad, int act, float slope,
                                       hipStream_t stream) {
    RG_REQUIRE(x && y && N > 0 && C > 0 && pad >= 0 && pad < H && pad < W, "rg_reflection_pad2d_fwd: pad must be < H and W");
    RG_REQUIRE(act == RG_ACT_NONE || act == RG_ACT_RELU || (act == RG_ACT_LEAKY && slope > 0.f),
               "rg_reflection_pad2d_fwd: fused activation must be none, relu or leaky relu with a positive slope");
    const int OH = H + 2 * pad, OW = W + 2 * pad;
    const int64_t total = (int64_t)N * C * OH * OW;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 4.0 * (total + (double)N * C * H * W));
    RG_REQUIRE(total < (1ll << 29) - (1 << 22), "rg_reflection_pad2d_fwd: more than 2^29 elements");
    const bool vec = (W & 3) == 0 && pad <= 3 && W >= 8 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) {
        const int64_t total4 = (int64_t)N * C * H * (W >> 2);
        hipLaunchKernelGGL(reflect_pad_fwd_vec_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, x, y, H, W, pad, OH, OW, (int)total4,
                           make_paddiv(W >> 2), make_paddiv(H), act, slope, (unsigned)(total * 4));
        return rg::check_launch("rg_reflection_pad2d_fwd");
    }
    RG_REQUIRE(act == RG_ACT_NONE, "rg_reflection_pad2d_fwd: the fused activation needs W %% 4 == 0, W >= 8 and pad <= 3");
    hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, H, W, pad, OH, OW, (int)total,
                       make_paddiv(OW), make_paddiv(OH));
    return rg::check_launch("rg_reflection_pad2d_fwd");
}

// dx = act'(x_act) * pad^T(dy); x_act: the tensor the forward padded (needed when act != RG_ACT_NONE)
extern "C" int rg_reflection_pad2d_bwd(const float* dy, const float* x_act, float* dx, int N, int C, int H, int W, int pad, int act,
                                       float slope, hipStream_t stream) {
    RG_REQUIRE(dy && dx && N > 0 && C > 0 && pad >= 0 && pad < H && pad < W, "rg_reflection_pad2d_bwd: pad must be < H and W");
    RG_REQUIRE(act == RG_ACT_NONE || (x_act && (act == RG_ACT_RELU || (act == RG_ACT_LEAKY && slope > 0.f))),
               "rg_reflection_pad2d_bwd: fused activation must be none, relu or leaky relu with a positive slope, and needs x");
    const int OH = H + 2 * pad, OW = W + 2 * pad;
    const int64_t total = (int64_t)N * C * H * W;
    rg::ProfScope prof(rg::FAM_ELTWISE, stream, 0.0, 4.0 * ((act != RG_ACT_NONE ? 2.0 : 1.0) * total + (double)N * C * OH * OW));
    RG_REQUIRE((int64_t)N * C * OH * OW < (1ll << 29) - (1 << 22), "rg_reflection_pad2d_bwd: more than 2^29 elements");
    const bool vec = (W & 3) == 0 && pad <= 3 && W >= 8 && (((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(x_act)) & 15) == 0);
    if (vec) {
        const int64_t total4 = total >> 2;
        hipLaunchKernelGGL(reflect_pad_bwd_vec_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, dy, x_act, dx, H, W, pad, OH, OW,
                           (int)total4, make_paddiv(W >> 2), make_paddiv(H), act, slope, (unsigned)((int64_t)N * C * OH * OW * 4));
        return rg::check_launch("rg_reflection_pad2d_bwd");
    }
    RG_REQUIRE(act == RG_ACT_NONE, "rg_reflection_pad2d_bwd: the fused activation needs W %% 4 == 0, W >= 8 and pad <= 3");
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, dx, H, W, pad, OH, OW, (int)total,
                       make_paddiv(W), make_paddiv(H));
    return rg::check_launch("rg_reflection_pad2d_bwd");
}

extern "C" int rg_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma, float* uv_saved, int K,
                                    int M, int training, float eps, hipStream_t stream) {
    RG_REQUIRE(w && u && v && w_sn && sigma && K > 0 && M > 0, "rg_spectral_norm_fwd: bad arguments");
    RG_REQUIRE(K <= 1024 && M <= SN_MAX_M, "rg_spectral_norm_fwd: matrix %d x %d exceeds the single-workgroup limits (1024 x %d)",
               K, M, SN_MAX_M);
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 12.0 * K * M);
    hipLaunchKernelGGL(spectral_norm_power_kernel, dim3(1), dim3(SN_THREADS), 0, stream, w, u, v, sigma, uv_saved, K, M, training,
                       eps);
    const int64_t n = (int64_t)K * M;
    hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, w, w_sn, sigma + 1, n);
    return rg::check_launch("rg_spectral_norm_fwd");
}

extern "C" int rg_spectral_norm_fwd_multi(const rg_sn_desc* descs, int count, int training, float eps, hipStream_t stream) {
    RG_REQUIRE(descs && count > 0, "rg_spectral_norm_fwd_multi: bad arguments");
    double bytes = 0.0;
    for (int i = 0; i < count; ++i) {
        const rg_sn_desc& d = descs[i];
        RG_REQUIRE(d.w && d.u && d.v && d.w_sn && d.sigma && d.K > 0 && d.M > 0, "rg_spectral_norm_fwd_multi: bad descriptor %d", i);
        RG_REQUIRE(d.K <= 1024 && d.M <= SN_MAX_M, "rg_spectral_norm_fwd_multi: matrix %d x %d exceeds the single-workgroup limits (1024 x %d)",
                   d.K, d.M, SN_MAX_M);
        bytes += 12.0 * d.K * d.M;
    }
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, bytes);
    for (int i0 = 0; i0 < count; i0 += RG_SN_MAX_BATCH) {
        const int nb = count - i0 < RG_SN_MAX_BATCH ? count - i0 : RG_SN_MAX_BATCH;
        SNBatch batch;
        int64_t nmax = 0;
        for (int i = 0; i < nb; ++i) {
            batch.d[i] = descs[i0 + i];
            const int64_t n = (int64_t)batch.d[i].K * batch.d[i].M;
            if (n > nmax) nmax = n;
        }
        hipLaunchKernelGGL(spectral_norm_power_multi_kernel, dim3(nb), dim3(SN_THREADS), 0, stream, batch, training, eps);
        unsigned gx = (unsigned)rg::cdiv64(nmax, 256 * 4);
        if (gx > 256) gx = 256;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(spectral_norm_scale_multi_kernel, dim3(gx, nb), dim3(256), 0, stream, batch);
    }
    return rg::check_launch("rg_spectral_norm_fwd_multi");
}

namespace {
// larger filters: the dot product as slice partials (one workgroup per slice), then an element-wise pass in which every
// workgroup adds the SN_SLICES partials in the same order — two parallel launches instead of one serial workgroup
constexpr int SN_SLICES = 64;
constexpr int SN_SINGLE_MAX = 16384;      // up to here the one-workgroup kernel (a single launch) is faster

__global__ __launch_bounds__(256) void spectral_norm_dot_partial_kernel(const float* __restrict__ dwsn,
                                                                       const float* __restrict__ wsn, float* __restrict__ part,
                                                                       int64_t n) {
    __shared__ float red[16];
    const int64_t per = (n + SN_SLICES - 1) / SN_SLICES;
    const int64_t beg = (int64_t)blockIdx.x * per, end = beg + per < n ? beg + per : n;
    float s = 0.f;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) s += dwsn[i] * wsn[i];
    s = rg_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void spectral_norm_bwd_apply_kernel(const float* __restrict__ dwsn, const float* __restrict__ part,
                                                                     const float* __restrict__ u, const float* __restrict__ v,
                                                                     const float* __restrict__ sigma, float* __restrict__ dw,
                                                                     int K, int M, int accumulate) {
    float c = 0.f;
    for (int i = 0; i < SN_SLICES; ++i) c += part[i];              // uniform: scalar loads, same order in every workgroup
    const float inv = sigma[1];
    const int64_t n = (int64_t)K * M;
    RG_GRID_STRIDE(i, n) {
        const int k = (int)(i / M), m = (int)(i - (int64_t)k * M);
        const float g = (dwsn[i] - c * u[k] * v[m]) * inv;
        dw[i] = accumulate ? dw[i] + g : g;
    }
}

}  // namespace

extern "C" size_t rg_spectral_norm_bwd_workspace(int K, int M) {
    return (int64_t)K * M > SN_SINGLE_MAX ? SN_SLICES * sizeof(float) : 0;
}

extern "C" int rg_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma,
                                    float* dw, int K, int M, int accumulate, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    RG_REQUIRE(dw_sn && w_sn && u && v && sigma && dw && K > 0 && M > 0, "rg_spectral_norm_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 16.0 * K * M);
    const int64_t n = (int64_t)K * M;
    if (n > SN_SINGLE_MAX) {
        if (!workspace || workspace_bytes < SN_SLICES * sizeof(float)) {
            rg::set_error("rg_spectral_norm_bwd: workspace too small");
            return RG_ERR_WORKSPACE;
        }
        float* part = static_cast<float*>(workspace);
        hipLaunchKernelGGL(spectral_norm_dot_partial_kernel, dim3(SN_SLICES), dim3(256), 0, stream, dw_sn, w_sn, part, n);
        hipLaunchKernelGGL(spectral_norm_bwd_apply_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dw_sn, part, u, v, sigma, dw,
                           K, M, accumulate);
        return rg::check_launch("rg_spectral_norm_bwd");
    }
    hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(1), dim3(SN_THREADS), 0, stream, dw_sn, w_sn, u, v, sigma, dw, K, M,
                       accumulate);
    return rg::check_launch("rg_spectral_norm_bwd");
}
