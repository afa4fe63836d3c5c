// BatchNorm (1d/2d, train and eval) for gfx950 — HBM-bound kernels: coalesced float4 streams,
// wave-shuffle (Chan) merges for the statistics, one read + one write per activation element.
//
// Replaces torch.nn.BatchNorm2d / BatchNorm1d as used by
//   ResNet-50 bottlenecks       CC/clustercontrast/models/resnet_ibn_a.py:70-109 (train mode in cluster-contrast,
//                               eval mode with trainable affine in FD-GAN: FD/fdgan/model.py:72-85, networks.py:57-60)
//   CustomPoseGenerator / NLayerDiscriminator norm layers   FD/fdgan/networks.py:26-35,141-156,218-229
//   EltwiseSubEmbed.bn, feat_bn                              FD/reid/models/embedding.py:16-19, CC/.../resnet.py:58-66
// Semantics follow torch: biased variance for normalisation, unbiased for running_var,
// running = (1-momentum)*running + momentum*batch.
//
// Tensor layout: [N][C][HW] contiguous (HW = 1 for BatchNorm1d on [N][C]).
#include "rg_common.h"

namespace {

struct WStat {
    float n, mean, m2;
};

__device__ __forceinline__ WStat chan_merge(WStat a, WStat b) {
    WStat r;
    r.n = a.n + b.n;
    if (r.n == 0.f) {
        r.mean = 0.f;
        r.m2 = 0.f;
        return r;
    }
    const float d = b.mean - a.mean;
    const float f = b.n / r.n;
    r.mean = a.mean + d * f;
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    return r;
}

__device__ __forceinline__ WStat wave_merge(WStat s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        WStat o;
        o.n = __shfl_xor(s.n, off, 64);
        o.mean = __shfl_xor(s.mean, off, 64);
        o.m2 = __shfl_xor(s.m2, off, 64);
        s = chan_merge(s, o);
    }
    return s;
}

// grid (S, C): block (s, c) reduces elements [s*L, (s+1)*L) of channel c's N*HW values.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                               int N, int C, int HW, int L) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const int64_t total = (int64_t)N * HW;
    const int64_t beg = (int64_t)s * L;
    int64_t end = beg + L;
    if (end > total) end = total;
    float shift = 0.f, sum = 0.f, sq = 0.f, cnt = 0.f;
    bool first = true;
    if ((HW & 3) == 0) {
        for (int64_t e = beg + (int64_t)threadIdx.x * 4; e < end; e += 256 * 4) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)n * C + c) * HW + hw);
            if (first) {
                shift = v.x;
                first = false;
            }
            const float a = v.x - shift, b = v.y - shift, cc = v.z - shift, d = v.w - shift;
            sum += (a + b) + (cc + d);
            sq += (a * a + b * b) + (cc * cc + d * d);
            cnt += 4.f;
        }
    } else {
        for (int64_t e = beg + threadIdx.x; e < end; e += 256) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const float v = x[((int64_t)n * C + c) * HW + hw];
            if (first) {
                shift = v;
                first = false;
            }
            const float a = v - shift;
            sum += a;
            sq += a * a;
            cnt += 1.f;
        }
    }
    WStat st;
    st.n = cnt;
    st.mean = cnt > 0.f ? shift + sum / cnt : 0.f;
    st.m2 = cnt > 0.f ? fmaxf(sq - sum * sum / cnt, 0.f) : 0.f;
    st = wave_merge(st);
    __shared__ WStat red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        WStat t = red[0];
        for (int i = 1; i < 4; ++i) t = chan_merge(t, red[i]);
        float* o = part + ((int64_t)c * S + s) * 3;
        o[0] = t.n;
        o[1] = t.mean;
        o[2] = t.m2;
    }
}

__global__ void bn_stats_finalize_kernel(const float* __restrict__ part, int C, int S, float eps, float momentum,
                                         float* __restrict__ mean, float* __restrict__ invstd,
                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    WStat t;
    t.n = 0.f;
    t.mean = 0.f;
    t.m2 = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* o = part + ((int64_t)c * S + s) * 3;
        WStat b;
        b.n = o[0];
        b.mean = o[1];
        b.m2 = o[2];
        t = chan_merge(t, b);
    }
    const float var = t.n > 0.f ? t.m2 / t.n : 0.f;
    mean[c] = t.mean;
    invstd[c] = rsqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * t.mean;
    if (running_var) {
        const float unb = t.n > 1.f ? t.m2 / (t.n - 1.f) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}

// y = act((x - mean[c]) * invstd[c] * gamma[c] + beta[c] + res)
// stat_is_var: `invstd` holds a variance (eval mode: running_var) and eps is applied here.
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ y, int64_t total, int C, int HW,
                                                           int stat_is_var, float eps, int act, float slope) {
    if ((HW & 3) == 0) {
        const int64_t nv = total >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t e = i << 2;
            const int c = (int)((e / HW) % C);
            float is = invstd[c];
            if (stat_is_var) is = rsqrtf(is + eps);
            const float sc = is * (gamma ? gamma[c] : 1.f);
            const float sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
            float4 v = *reinterpret_cast<const float4*>(x + e);
            v.x = v.x * sc + sh;
            v.y = v.y * sc + sh;
            v.z = v.z * sc + sh;
            v.w = v.w * sc + sh;
            if (res) {
                const float4 r = *reinterpret_cast<const float4*>(res + e);
                v.x += r.x;
                v.y += r.y;
                v.z += r.z;
                v.w += r.w;
            }
            v.x = rg_apply_act(v.x, act, slope);
            v.y = rg_apply_act(v.y, act, slope);
            v.z = rg_apply_act(v.z, act, slope);
            v.w = rg_apply_act(v.w, act, slope);
            *reinterpret_cast<float4*>(y + e) = v;
        }
    } else {
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)((e / HW) % C);
            float is = invstd[c];
            if (stat_is_var) is = rsqrtf(is + eps);
            const float sc = is * (gamma ? gamma[c] : 1.f);
            const float sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
            float v = x[e] * sc + sh;
            if (res) v += res[e];
            y[e] = rg_apply_act(v, act, slope);
        }
    }
}

__device__ __forceinline__ float act_grad_from_out(float yv, int act, float slope) {
    switch (act) {
        case RG_ACT_RELU: return yv > 0.f ? 1.f : 0.f;
        case RG_ACT_LEAKY: return yv > 0.f ? 1.f : slope;
        case RG_ACT_TANH: return 1.f - yv * yv;
        default: return 1.f;
    }
}

// partial sums over a slice: sum(dy_eff), sum(dy_eff * xhat); dy_eff = dy * act'(y)
__global__ __launch_bounds__(256) void bn_bwd_reduce_partial_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ dy,
                                                                    const float* __restrict__ yact,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    float* __restrict__ part, int N, int C, int HW,
                                                                    int L, int stat_is_var, float eps, int act,
                                                                    float slope) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const int64_t total = (int64_t)N * HW;
    const int64_t beg = (int64_t)s * L;
    int64_t end = beg + L;
    if (end > total) end = total;
    const float mu = mean[c];
    float is = invstd[c];
    if (stat_is_var) is = rsqrtf(is + eps);
    float s1 = 0.f, s2 = 0.f;
    if ((HW & 3) == 0) {
        for (int64_t e = beg + (int64_t)threadIdx.x * 4; e < end; e += 256 * 4) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            const float4 xv = *reinterpret_cast<const float4*>(x + o);
            float4 g = *reinterpret_cast<const float4*>(dy + o);
            if (act != RG_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4*>(yact + o);
                g.x *= act_grad_from_out(yv.x, act, slope);
                g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope);
                g.w *= act_grad_from_out(yv.w, act, slope);
            }
            s1 += (g.x + g.y) + (g.z + g.w);
            s2 += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
        }
    } else {
        for (int64_t e = beg + threadIdx.x; e < end; e += 256) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            float g = dy[o];
            if (act != RG_ACT_NONE) g *= act_grad_from_out(yact[o], act, slope);
            s1 += g;
            s2 += g * (x[o] - mu);
        }
    }
    s2 *= is;
    __shared__ float red[16];
    s1 = rg_block_sum(s1, red);
    s2 = rg_block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* o = part + ((int64_t)c * S + s) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

// Eval-mode backward in ONE pass (running statistics: dx does not depend on the channel sums):
//   g = dy * act'(y);  dx = gamma*invstd * g;  dres = g;  partial sums of g and g*xhat for dbeta / dgamma.
// Same (slice, channel) decomposition as the reduction kernel, so it reads x, dy, y once (12 B/element) and
// writes dx (+ dres) instead of the two-kernel reduce + apply sequence (28 B/element).
__global__ __launch_bounds__(256) void bn_eval_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ yact,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ var,
                                                                const float* __restrict__ gamma, float* __restrict__ dx,
                                                                float* __restrict__ dres, float* __restrict__ part,
                                                                int N, int C, int HW, int L, float eps, int act,
                                                                float slope, int want_sums) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const int64_t total = (int64_t)N * HW;
    const int64_t beg = (int64_t)s * L;
    int64_t end = beg + L;
    if (end > total) end = total;
    const float mu = mean[c];
    const float is = rsqrtf(var[c] + eps);
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    float s1 = 0.f, s2 = 0.f;
    if ((HW & 3) == 0) {
        for (int64_t e = beg + (int64_t)threadIdx.x * 4; e < end; e += 256 * 4) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            float4 g = *reinterpret_cast<const float4*>(dy + o);
            if (act != RG_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4*>(yact + o);
                g.x *= act_grad_from_out(yv.x, act, slope);
                g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope);
                g.w *= act_grad_from_out(yv.w, act, slope);
            }
            if (dres) *reinterpret_cast<float4*>(dres + o) = g;
            if (dx) *reinterpret_cast<float4*>(dx + o) = make_float4(gs * g.x, gs * g.y, gs * g.z, gs * g.w);
            if (want_sums) {
                const float4 xv = *reinterpret_cast<const float4*>(x + o);
                s1 += (g.x + g.y) + (g.z + g.w);
                s2 += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
            }
        }
    } else {
        for (int64_t e = beg + threadIdx.x; e < end; e += 256) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            float g = dy[o];
            if (act != RG_ACT_NONE) g *= act_grad_from_out(yact[o], act, slope);
            if (dres) dres[o] = g;
            if (dx) dx[o] = gs * g;
            if (want_sums) {
                s1 += g;
                s2 += g * (x[o] - mu);
            }
        }
    }
    if (!want_sums) return;
    s2 *= is;
    __shared__ float red[16];
    s1 = rg_block_sum(s1, red);
    s2 = rg_block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* o = part + ((int64_t)c * S + s) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

__global__ void bn_bwd_reduce_finalize_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ sum_dy,
                                              float* __restrict__ sum_dy_xhat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < S; ++s) {
        a += part[((int64_t)c * S + s) * 2 + 0];
        b += part[((int64_t)c * S + s) * 2 + 1];
    }
    sum_dy[c] = a;
    sum_dy_xhat[c] = b;
}

// dx = gamma*invstd * (g - train*(sum_dy/cnt + xhat*sum_dy_xhat/cnt)),  g = dy*act'(y);  dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ yact,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_dy,
                                                           const float* __restrict__ sum_dy_xhat,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           int64_t total, int C, int HW, float inv_count, int train,
                                                           int stat_is_var, float eps, int act, float slope) {
    const bool vec = (HW & 3) == 0;
    const int64_t nitems = vec ? (total >> 2) : total;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nitems; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = vec ? (i << 2) : i;
        const int c = (int)((e / HW) % C);
        const float mu = mean[c];
        float is = invstd[c];
        if (stat_is_var) is = rsqrtf(is + eps);
        const float gs = (gamma ? gamma[c] : 1.f) * is;
        const float a = train ? sum_dy[c] * inv_count : 0.f;
        const float b = train ? sum_dy_xhat[c] * inv_count : 0.f;
        if (vec) {
            float4 g = *reinterpret_cast<const float4*>(dy + e);
            if (act != RG_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4*>(yact + e);
                g.x *= act_grad_from_out(yv.x, act, slope);
                g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope);
                g.w *= act_grad_from_out(yv.w, act, slope);
            }
            if (dres) *reinterpret_cast<float4*>(dres + e) = g;
            if (dx) {
                float4 o;
                if (train) {
                    const float4 xv = *reinterpret_cast<const float4*>(x + e);
                    o.x = gs * (g.x - a - (xv.x - mu) * is * b);
                    o.y = gs * (g.y - a - (xv.y - mu) * is * b);
                    o.z = gs * (g.z - a - (xv.z - mu) * is * b);
                    o.w = gs * (g.w - a - (xv.w - mu) * is * b);
                } else {
                    o.x = gs * g.x;
                    o.y = gs * g.y;
                    o.z = gs * g.z;
                    o.w = gs * g.w;
                }
                *reinterpret_cast<float4*>(dx + e) = o;
            }
        } else {
            float g = dy[e];
            if (act != RG_ACT_NONE) g *= act_grad_from_out(yact[e], act, slope);
            if (dres) dres[e] = g;
            if (dx) dx[e] = train ? gs * (g - a - (x[e] - mu) * is * b) : gs * g;
        }
    }
}

// ---- conv + frozen-statistics BatchNorm folded into the convolution ---------------------------------------------
// y = act(conv(x, W) * scale + shift [+ residual]) runs in the conv epilogue (scale = gamma * invstd,
// shift = beta - mean * scale), so the pre-normalisation tensor z is never written.  Backward without z:
//   g      = dy * act'(y)                         (this kernel; + per-channel sum of g = dbeta)
//   G      = wgrad(x, g)                          (conv kernel)        because sum_{n,p,q} g z = sum_{c,r,s} W G
//   dgamma = invstd * (sum W.G - mean * sum g);  dW = scale * G        (bn_fold_wgrad_kernel, one workgroup per filter)
//   dx     = dgrad(g, scale * W)                  (conv kernel on row-scaled filters, scale_rows_kernel)
__global__ __launch_bounds__(256) void act_bwd_sum_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                          float* __restrict__ g_out, float* __restrict__ part, int N, int C,
                                                          int HW, int L, int act, float slope, int want_sum) {
    const int c = blockIdx.y, s = blockIdx.x, S = gridDim.x;
    const int64_t total = (int64_t)N * HW;
    const int64_t beg = (int64_t)s * L;
    int64_t end = beg + L;
    if (end > total) end = total;
    float s1 = 0.f;
    if ((HW & 3) == 0) {
        for (int64_t e = beg + (int64_t)threadIdx.x * 4; e < end; e += 256 * 4) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            float4 g = *reinterpret_cast<const float4*>(dy + o);
            if (act != RG_ACT_NONE) {
                const float4 yv = *reinterpret_cast<const float4*>(yact + o);
                g.x *= act_grad_from_out(yv.x, act, slope);
                g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope);
                g.w *= act_grad_from_out(yv.w, act, slope);
            }
            if (g_out) *reinterpret_cast<float4*>(g_out + o) = g;
            s1 += (g.x + g.y) + (g.z + g.w);
        }
    } else {
        for (int64_t e = beg + threadIdx.x; e < end; e += 256) {
            const int n = (int)(e / HW);
            const int hw = (int)(e - (int64_t)n * HW);
            const int64_t o = ((int64_t)n * C + c) * HW + hw;
            float g = dy[o];
            if (act != RG_ACT_NONE) g *= act_grad_from_out(yact[o], act, slope);
            if (g_out) g_out[o] = g;
            s1 += g;
        }
    }
    if (!want_sum) return;
    __shared__ float red[16];
    s1 = rg_block_sum(s1, red);
    if (threadIdx.x == 0) part[(int64_t)c * S + s] = s1;
}

__global__ void sum_slices_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(int64_t)c * S + s];
    out[c] = a;
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift,
                               float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = rsqrtf(var[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - mean[c] * sc;
    invstd[c] = is;
}

__global__ __launch_bounds__(256) void bn_fold_wgrad_kernel(const float* __restrict__ w, float* __restrict__ g,
                                                            const float* __restrict__ scale, const float* __restrict__ invstd,
                                                            const float* __restrict__ mean, const float* __restrict__ sum_g,
                                                            const float* __restrict__ part, int S, float* __restrict__ dbeta,
                                                            float* __restrict__ dgamma, int M) {
    __shared__ float red[16];
    const int k = blockIdx.x;
    const float* wr = w + (int64_t)k * M;
    float* gr = g + (int64_t)k * M;
    float sg = 0.f;
    if (part) {                       // channel sum of g from slice / tile partials (fixed tree: deterministic)
        float t = 0.f;
        for (int s = threadIdx.x; s < S; s += 256) t += part[(int64_t)k * S + s];
        sg = rg_block_sum(t, red);
        if (dbeta && threadIdx.x == 0) dbeta[k] = sg;
    } else if (sum_g) {
        sg = sum_g[k];
    }
    if (dgamma) {
        float t = 0.f;
        for (int m = threadIdx.x; m < M; m += 256) t += wr[m] * gr[m];
        t = rg_block_sum(t, red);
        if (threadIdx.x == 0) dgamma[k] = invstd[k] * (t - mean[k] * sg);
    }
    const float sc = scale[k];
    for (int m = threadIdx.x; m < M; m += 256) gr[m] *= sc;
}

__global__ void scale_rows_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out, int M,
                                  int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = w[i] * scale[i / M];
}

// One launch folds EVERY (conv, frozen BatchNorm) pair of a network: per pair scale / shift / invstd and the filters
// multiplied by scale[k] in the checkpoint layout [K][C][RS] and, when requested, in [K][RS][C] for the (r,s)-major
// loaders.  The table lives in device memory, 16 int64 words per pair:
//   0 w  1 gamma  2 beta  3 running_mean  4 running_var  5 w_scaled  6 w_scaled_krsc (0 = none)  7 scale  8 shift  9 invstd
//   10 K  11 C  12 RS  13 eps (float bits)  14 first block of the pair  15 unused
constexpr int FOLD_WORDS = 16;
constexpr int FOLD_CHUNK = 2048;          // filter elements per workgroup

__global__ __launch_bounds__(256) void fold_filters_multi_kernel(const long long* __restrict__ tab, int n_pairs) {
    // binary search of the pair this block belongs to
    int lo = 0, hi = n_pairs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[(int64_t)mid * FOLD_WORDS + 14] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* e = tab + (int64_t)lo * FOLD_WORDS;
    const float* w = reinterpret_cast<const float*>(e[0]);
    const float* gamma = reinterpret_cast<const float*>(e[1]);
    const float* beta = reinterpret_cast<const float*>(e[2]);
    const float* mean = reinterpret_cast<const float*>(e[3]);
    const float* var = reinterpret_cast<const float*>(e[4]);
    float* ws = reinterpret_cast<float*>(e[5]);
    float* wk = reinterpret_cast<float*>(e[6]);
    float* scale = reinterpret_cast<float*>(e[7]);
    float* shift = reinterpret_cast<float*>(e[8]);
    float* invstd = reinterpret_cast<float*>(e[9]);
    const int K = (int)e[10], C = (int)e[11], RS = (int)e[12];
    const float eps = __int_as_float((int)e[13]);
    const int blk = (int)((long long)blockIdx.x - e[14]);
    const int64_t total = (int64_t)K * C * RS;
    const int64_t crs = (int64_t)C * RS;
    const int64_t beg = (int64_t)blk * FOLD_CHUNK;
    for (int64_t i = beg + threadIdx.x; i < beg + FOLD_CHUNK && i < total; i += 256) {
        const int k = (int)(i / crs);
        const int rem = (int)(i - (int64_t)k * crs);
        const float is = rsqrtf(var[k] + eps);
        const float sc = gamma[k] * is;
        const float v = w[i] * sc;
        ws[i] = v;
        if (wk) {
            const int c = rem / RS, rs = rem - c * RS;
            wk[((int64_t)k * RS + rs) * C + c] = v;
        }
        if (rem == 0) {
            scale[k] = sc;
            shift[k] = beta[k] - mean[k] * sc;
            invstd[k] = is;
        }
    }
}

static int pick_slices(int N, int C, int HW, int* L) {
    const int64_t total = (int64_t)N * HW;
    int64_t S = rg::cdiv(2048, C);
    const int64_t maxS = rg::cdiv64(total, 4096);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    int64_t l = rg::cdiv64(total, S);
    l = (l + 3) & ~3ll;
    *L = (int)l;
    return (int)rg::cdiv64(total, l);
}

static unsigned grid_for(int64_t items) {
    int64_t g = rg::cdiv64(items, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" size_t rg_bn_workspace(int N, int C, int HW) {
    int L;
    const int S = pick_slices(N, C, HW, &L);
    return (size_t)C * S * 3 * sizeof(float);
}

// Batch statistics of x[N][C][HW]: mean[C], invstd[C] (=1/sqrt(biased var + eps)); updates the
// running statistics in place when they are given.
extern "C" int rg_bn_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var, int N,
                           int C, int HW, float eps, float momentum, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
    RG_REQUIRE(x && mean && invstd && N > 0 && C > 0 && HW > 0, "rg_bn_stats: bad arguments");
    RG_REQUIRE((int64_t)N * HW < (1ll << 31), "rg_bn_stats: N*HW exceeds 2^31");
    int L;
    const int S = pick_slices(N, C, HW, &L);
    if (!workspace || workspace_bytes < (size_t)C * S * 3 * sizeof(float)) {
        rg::set_error("rg_bn_stats: workspace too small");
        return RG_ERR_WORKSPACE;
    }
    float* part = static_cast<float*>(workspace);
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 4.0 * N * (double)C * HW);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(S, C), dim3(256), 0, stream, x, part, N, C, HW, L);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(rg::cdiv(C, 64)), dim3(64), 0, stream, part, C, S, eps, momentum,
                       mean, invstd, running_mean, running_var);
    return rg::check_launch("rg_bn_stats");
}

extern "C" int rg_bn_apply_fwd(const float* x, const float* mean, const float* stat, const float* gamma,
                               const float* beta, const float* residual, float* y, int N, int C, int HW,
                               int stat_is_var, float eps, int act, float slope, hipStream_t stream) {
    RG_REQUIRE(x && mean && stat && y && N > 0 && C > 0 && HW > 0, "rg_bn_apply_fwd: bad arguments");
    const int64_t total = (int64_t)N * C * HW;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, (residual ? 12.0 : 8.0) * total);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid_for((HW & 3) ? total : total / 4)), dim3(256), 0, stream, x, mean,
                       stat, gamma, beta, residual, y, total, C, HW, stat_is_var, eps, act, slope);
    return rg::check_launch("rg_bn_apply_fwd");
}

// Per-channel sums for the affine gradients: dbeta = sum(dy*act'(y)), dgamma = sum(dy*act'(y)*xhat).
extern "C" int rg_bn_bwd_reduce(const float* x, const float* dy, const float* y_act, const float* mean,
                                const float* stat, float* sum_dy, float* sum_dy_xhat, int N, int C, int HW,
                                int stat_is_var, float eps, int act, float slope, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(x && dy && mean && stat && sum_dy && sum_dy_xhat, "rg_bn_bwd_reduce: null tensor");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_bn_bwd_reduce: fused activation needs the forward output");
    int L;
    const int S = pick_slices(N, C, HW, &L);
    if (!workspace || workspace_bytes < (size_t)C * S * 2 * sizeof(float)) {
        rg::set_error("rg_bn_bwd_reduce: workspace too small");
        return RG_ERR_WORKSPACE;
    }
    float* part = static_cast<float*>(workspace);
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, (act ? 12.0 : 8.0) * N * (double)C * HW);
    hipLaunchKernelGGL(bn_bwd_reduce_partial_kernel, dim3(S, C), dim3(256), 0, stream, x, dy, y_act, mean, stat, part, N,
                       C, HW, L, stat_is_var, eps, act, slope);
    hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3(rg::cdiv(C, 64)), dim3(64), 0, stream, part, C, S, sum_dy,
                       sum_dy_xhat);
    return rg::check_launch("rg_bn_bwd_reduce");
}

// ---- InstanceNorm2d in one launch per direction -----------------------------------------------------------------------
// An instance (n, c) is HW contiguous floats: <= 32 KB for every map of the dual_gan / FD-GAN networks, so the second and third
// pass over it hit L1 / L2.  LANES = 16 / 32 / 64 lanes per instance (several instances per wave for the small maps, shuffles only),
// 256: one workgroup per instance.  Statistics: mean, then the centred sum of squares (two-pass, biased variance).
template <int LANES>
__device__ __forceinline__ float in_reduce(float v, float* red) {
    // LANES <= 64: butterfly inside the aligned lane group (several instances share a wave); 256: the whole workgroup
#pragma unroll
    for (int off = (LANES < 64 ? LANES : 64) / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (LANES <= 64) return v;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <int LANES>
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, int NC, int C, int HW, float eps,
                                                           int act, float slope) {
    __shared__ float red[4];
    const int inst = (int)blockIdx.x * (256 / LANES) + (int)threadIdx.x / LANES;
    if (inst >= NC) return;                                   // whole lane groups leave (LANES == 256: never taken, grid = NC)
    const int t = (int)threadIdx.x % LANES;
    constexpr int T = LANES;
    const int64_t base = (int64_t)inst * HW;
    const float* xp = x + base;
    const bool vec = (HW & 3) == 0;
    const int nv = HW >> 2;
    float s = 0.f;
    if (vec) {
        for (int i = t; i < nv; i += T) {
            const float4 v = reinterpret_cast<const float4*>(xp)[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = t; i < HW; i += T) s += xp[i];
    }
    const float mu = in_reduce<LANES>(s, red) / (float)HW;
    float q = 0.f;
    if (vec) {
        for (int i = t; i < nv; i += T) {
            const float4 v = reinterpret_cast<const float4*>(xp)[i];
            const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int i = t; i < HW; i += T) {
            const float a = xp[i] - mu;
            q += a * a;
        }
    }
    const float is = rsqrtf(in_reduce<LANES>(q, red) / (float)HW + eps);
    if (t == 0) {
        mean_out[inst] = mu;
        invstd_out[inst] = is;
    }
    const int c = inst % C;
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float sh = (beta ? beta[c] : 0.f) - mu * gs;
    float* yp = y + base;
    const float* rp = res ? res + base : nullptr;
    if (vec) {
        for (int i = t; i < nv; i += T) {
            const float4 v = reinterpret_cast<const float4*>(xp)[i];
            float4 o;
            o.x = v.x * gs + sh; o.y = v.y * gs + sh; o.z = v.z * gs + sh; o.w = v.w * gs + sh;
            if (rp) {
                const float4 r = reinterpret_cast<const float4*>(rp)[i];
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            o.x = rg_apply_act(o.x, act, slope); o.y = rg_apply_act(o.y, act, slope);
            o.z = rg_apply_act(o.z, act, slope); o.w = rg_apply_act(o.w, act, slope);
            reinterpret_cast<float4*>(yp)[i] = o;
        }
    } else {
        for (int i = t; i < HW; i += T) {
            float o = xp[i] * gs + sh;
            if (rp) o += rp[i];
            yp[i] = rg_apply_act(o, act, slope);
        }
    }
}

// g = dy * act'(y); s1 = sum g, s2 = sum g * xhat (written per instance for the affine gradients);
// dx = gamma * invstd * (g - s1 / HW - xhat * s2 / HW); dres = g
template <int LANES>
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ yact, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat,
                                                           float* __restrict__ sum_dx, int NC, int C, int HW, int act,
                                                           float slope) {
    __shared__ float red[4];
    const int inst = (int)blockIdx.x * (256 / LANES) + (int)threadIdx.x / LANES;
    if (inst >= NC) return;
    const int t = (int)threadIdx.x % LANES;
    constexpr int T = LANES;
    const int64_t base = (int64_t)inst * HW;
    const float* xp = x + base;
    const float* gp = dy + base;
    const float* yp = act != RG_ACT_NONE ? yact + base : nullptr;
    const float mu = mean[inst], is = invstd[inst];
    const bool vec = (HW & 3) == 0;
    const int nv = HW >> 2;
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        for (int i = t; i < nv; i += T) {
            float4 g = reinterpret_cast<const float4*>(gp)[i];
            if (yp) {
                const float4 yv = reinterpret_cast<const float4*>(yp)[i];
                g.x *= act_grad_from_out(yv.x, act, slope); g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope); g.w *= act_grad_from_out(yv.w, act, slope);
            }
            const float4 v = reinterpret_cast<const float4*>(xp)[i];
            s1 += (g.x + g.y) + (g.z + g.w);
            s2 += (g.x * (v.x - mu) + g.y * (v.y - mu)) + (g.z * (v.z - mu) + g.w * (v.w - mu));
        }
    } else {
        for (int i = t; i < HW; i += T) {
            float g = gp[i];
            if (yp) g *= act_grad_from_out(yp[i], act, slope);
            s1 += g;
            s2 += g * (xp[i] - mu);
        }
    }
    s1 = in_reduce<LANES>(s1, red);
    s2 = in_reduce<LANES>(s2, red) * is;
    if (t == 0) {
        sum_dy[inst] = s1;
        sum_dy_xhat[inst] = s2;
    }
    const float gs = (gamma ? gamma[inst % C] : 1.f) * is;
    const float a = s1 / (float)HW, b = s2 / (float)HW * is;
    float* dxp = dx ? dx + base : nullptr;
    float* drp = dres ? dres + base : nullptr;
    float sdx = 0.f;
    if (vec) {
        for (int i = t; i < nv; i += T) {
            float4 g = reinterpret_cast<const float4*>(gp)[i];
            if (yp) {
                const float4 yv = reinterpret_cast<const float4*>(yp)[i];
                g.x *= act_grad_from_out(yv.x, act, slope); g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope); g.w *= act_grad_from_out(yv.w, act, slope);
            }
            if (drp) reinterpret_cast<float4*>(drp)[i] = g;
            if (dxp) {
                const float4 v = reinterpret_cast<const float4*>(xp)[i];
                float4 o;
                o.x = gs * (g.x - a - (v.x - mu) * b); o.y = gs * (g.y - a - (v.y - mu) * b);
                o.z = gs * (g.z - a - (v.z - mu) * b); o.w = gs * (g.w - a - (v.w - mu) * b);
                reinterpret_cast<float4*>(dxp)[i] = o;
                sdx += (o.x + o.y) + (o.z + o.w);
            }
        }
    } else {
        for (int i = t; i < HW; i += T) {
            float g = gp[i];
            if (yp) g *= act_grad_from_out(yp[i], act, slope);
            if (drp) drp[i] = g;
            if (dxp) {
                const float o = gs * (g - a - (xp[i] - mu) * b);
                dxp[i] = o;
                sdx += o;
            }
        }
    }
    if (sum_dx) {                                                // uniform: the bias gradient of the convolution in front
        sdx = in_reduce<LANES>(sdx, red);
        if (t == 0) sum_dx[inst] = sdx;
    }
}

// ---- train-mode BatchNorm in one launch per direction, small per-channel extents ----------------------------------------------
// One workgroup per channel: its N rows of HW floats (stride C*HW) are <= 64 KB, so passes two and three hit L1 / L2.  Used when
// N*HW <= 16384 and the channel count alone fills the chip (layer3 / layer4 of the ResNets, the inner generator layers); the
// slice-parallel kernels above stay for the large maps.  Statistics as bn_stats (biased variance for the normalisation, unbiased
// for running_var), two-pass.
__device__ __forceinline__ float bn_block_sum(float v, float* red) {
    v = rg_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bn_train_fwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ res,
                                                                 float* __restrict__ y, float* __restrict__ mean_out,
                                                                 float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, int N, int C, int HW, float eps,
                                                                 float momentum, int act, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const bool vec = (HW & 3) == 0;
    const int rl = vec ? HW >> 2 : HW;                   // units (float4 or float) per row
    const int total = N * rl;
    const int64_t cs = (int64_t)C * HW;
    const float cnt = (float)N * (float)HW;
    float s = 0.f;
    for (int i = t; i < total; i += 256) {
        const int n = i / rl, v = i - n * rl;
        const float* p = x + (int64_t)n * cs + (int64_t)c * HW;
        if (vec) {
            const float4 a = reinterpret_cast<const float4*>(p)[v];
            s += (a.x + a.y) + (a.z + a.w);
        } else {
            s += p[v];
        }
    }
    const float mu = bn_block_sum(s, red) / cnt;
    float q = 0.f;
    for (int i = t; i < total; i += 256) {
        const int n = i / rl, v = i - n * rl;
        const float* p = x + (int64_t)n * cs + (int64_t)c * HW;
        if (vec) {
            const float4 a = reinterpret_cast<const float4*>(p)[v];
            const float d0 = a.x - mu, d1 = a.y - mu, d2 = a.z - mu, d3 = a.w - mu;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        } else {
            const float d = p[v] - mu;
            q += d * d;
        }
    }
    const float m2 = bn_block_sum(q, red);
    const float var = m2 / cnt;
    const float is = rsqrtf(var + eps);
    if (t == 0) {
        mean_out[c] = mu;
        invstd_out[c] = is;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (cnt > 1.f ? m2 / (cnt - 1.f) : var);
    }
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float sh = (beta ? beta[c] : 0.f) - mu * gs;
    for (int i = t; i < total; i += 256) {
        const int n = i / rl, v = i - n * rl;
        const int64_t off = (int64_t)n * cs + (int64_t)c * HW;
        if (vec) {
            const float4 a = reinterpret_cast<const float4*>(x + off)[v];
            float4 o;
            o.x = a.x * gs + sh; o.y = a.y * gs + sh; o.z = a.z * gs + sh; o.w = a.w * gs + sh;
            if (res) {
                const float4 r = reinterpret_cast<const float4*>(res + off)[v];
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            o.x = rg_apply_act(o.x, act, slope); o.y = rg_apply_act(o.y, act, slope);
            o.z = rg_apply_act(o.z, act, slope); o.w = rg_apply_act(o.w, act, slope);
            reinterpret_cast<float4*>(y + off)[v] = o;
        } else {
            float o = x[off + v] * gs + sh;
            if (res) o += res[off + v];
            y[off + v] = rg_apply_act(o, act, slope);
        }
    }
}

// ---- the same two kernels for maps whose rows are float4 multiples, with the loads taken out of the loops ---------------------------
// One workgroup per channel holds N*HW <= 16384 values = at most 16 float4 per thread.  The loops above issue one load per
// iteration behind an integer division and wait for it: with two workgroups per CU nothing hides that latency (a [32, 512, 16 x 8]
// layer: 15 us forward, 1 TB/s).  Here every thread computes its U offsets once (buffer resource over the tensor: offsets past
// the end return zeros / drop the store, so there are no bounds branches) and issues its loads together:
//   forward   x (and the residual) are loaded ONCE into registers; mean, centred sum of squares and the output come from them;
//   backward  two sweeps (the sums, then dx / dres) of U float4 per operand, eight units in flight per sweep.
// Same per-thread summation order and block reduction as the loop kernels: same statistics.
typedef __amdgpu_buffer_rsrc_t nrsrc_t;
constexpr unsigned NOOB = 0x80000000u;                  // tensors are < 2^31 bytes here (host check)
__device__ __forceinline__ nrsrc_t n_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 n_load4(nrsrc_t r, unsigned off) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void n_store4(nrsrc_t r, unsigned off, float4 v) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v w = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, w), r, off, 0, 0);
}

template <int U>
__device__ __forceinline__ void bn_unit_offsets(unsigned (&off)[U], int t, int total, int rl, int C, int c, int HW) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int i = t + 256 * j;
        const int n = i / rl, v = i - n * rl;            // U divisions per thread, once
        off[j] = i < total ? (unsigned)((n * C + c) * HW + 4 * v) * 4u : NOOB;
    }
}

template <int U>
__global__ __launch_bounds__(256) void bn_train_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ res,
                                                               float* __restrict__ y, float* __restrict__ mean_out,
                                                               float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, int N, int C, int HW, float eps,
                                                               float momentum, int act, float slope, unsigned bytes) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int rl = HW >> 2;
    const int total = N * rl;
    const float cnt = (float)N * (float)HW;
    const nrsrc_t rx = n_rsrc(x, bytes), ry = n_rsrc(y, bytes);
    const nrsrc_t rr = n_rsrc(res ? res : x, res ? bytes : 0u);
    unsigned off[U];
    bn_unit_offsets<U>(off, t, total, rl, C, c, HW);
    float4 a[U], r[U];
#pragma unroll
    for (int j = 0; j < U; ++j) a[j] = n_load4(rx, off[j]);
    if (res) {                                           // uniform; in flight behind x while the statistics are formed
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = n_load4(rr, off[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
    const float mu = bn_block_sum(s, red) / cnt;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const float d0 = a[j].x - mu, d1 = a[j].y - mu, d2 = a[j].z - mu, d3 = a[j].w - mu;
        const float e = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        q += off[j] != NOOB ? e : 0.f;
    }
    const float m2 = bn_block_sum(q, red);
    const float var = m2 / cnt;
    const float is = rsqrtf(var + eps);
    if (t == 0) {
        mean_out[c] = mu;
        invstd_out[c] = is;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (cnt > 1.f ? m2 / (cnt - 1.f) : var);
    }
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float sh = (beta ? beta[c] : 0.f) - mu * gs;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        float4 o;
        o.x = a[j].x * gs + sh; o.y = a[j].y * gs + sh; o.z = a[j].z * gs + sh; o.w = a[j].w * gs + sh;
        if (res) { o.x += r[j].x; o.y += r[j].y; o.z += r[j].z; o.w += r[j].w; }
        o.x = rg_apply_act(o.x, act, slope); o.y = rg_apply_act(o.y, act, slope);
        o.z = rg_apply_act(o.z, act, slope); o.w = rg_apply_act(o.w, act, slope);
        n_store4(ry, off[j], o);
    }
}

template <int U>
__global__ __launch_bounds__(256) void bn_train_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ yact, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               float* __restrict__ dx, float* __restrict__ dres,
                                                               float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat, int N,
                                                               int C, int HW, int act, float slope, unsigned bytes) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int rl = HW >> 2;
    const int total = N * rl;
    const float cnt = (float)N * (float)HW;
    const float mu = mean[c], is = invstd[c];
    const bool has_act = act != RG_ACT_NONE;
    const nrsrc_t rx = n_rsrc(x, bytes), rg = n_rsrc(dy, bytes), ry = n_rsrc(has_act ? yact : dy, has_act ? bytes : 0u);
    const nrsrc_t rdx = n_rsrc(dx ? dx : dres, dx ? bytes : 0u), rdr = n_rsrc(dres ? dres : dx, dres ? bytes : 0u);
    unsigned off[U];
    bn_unit_offsets<U>(off, t, total, rl, C, c, HW);
    constexpr int B = U < 8 ? U : 8;                     // units in flight per operand
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < U; j0 += B) {
        float4 g[B], a[B], yv[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            g[j] = n_load4(rg, off[j0 + j]);
            a[j] = n_load4(rx, off[j0 + j]);
            if (has_act) yv[j] = n_load4(ry, off[j0 + j]);
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (has_act) {
                g[j].x *= act_grad_from_out(yv[j].x, act, slope); g[j].y *= act_grad_from_out(yv[j].y, act, slope);
                g[j].z *= act_grad_from_out(yv[j].z, act, slope); g[j].w *= act_grad_from_out(yv[j].w, act, slope);
            }
            s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
            s2 += (g[j].x * (a[j].x - mu) + g[j].y * (a[j].y - mu)) + (g[j].z * (a[j].z - mu) + g[j].w * (a[j].w - mu));
        }
    }
    s1 = bn_block_sum(s1, red);
    s2 = bn_block_sum(s2, red) * is;
    if (t == 0) {
        sum_dy[c] = s1;
        sum_dy_xhat[c] = s2;
    }
    if (!dx && !dres) return;
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float a0 = s1 / cnt, b0 = s2 / cnt * is;
#pragma unroll
    for (int j0 = 0; j0 < U; j0 += B) {
        float4 g[B], a[B], yv[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            g[j] = n_load4(rg, off[j0 + j]);
            if (dx) a[j] = n_load4(rx, off[j0 + j]);
            if (has_act) yv[j] = n_load4(ry, off[j0 + j]);
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (has_act) {
                g[j].x *= act_grad_from_out(yv[j].x, act, slope); g[j].y *= act_grad_from_out(yv[j].y, act, slope);
                g[j].z *= act_grad_from_out(yv[j].z, act, slope); g[j].w *= act_grad_from_out(yv[j].w, act, slope);
            }
            if (dres) n_store4(rdr, off[j0 + j], g[j]);
            if (dx) {
                float4 o;
                o.x = gs * (g[j].x - a0 - (a[j].x - mu) * b0); o.y = gs * (g[j].y - a0 - (a[j].y - mu) * b0);
                o.z = gs * (g[j].z - a0 - (a[j].z - mu) * b0); o.w = gs * (g[j].w - a0 - (a[j].w - mu) * b0);
                n_store4(rdx, off[j0 + j], o);
            }
        }
    }
}

// ---- InstanceNorm2d with the instance in registers (HW % 4 == 0, at most 8 float4 per lane — every map of the networks here up to
// 128 x 64): the loop kernels above walk the instance three times (twice in the backward) with one load in flight per lane; here
// each lane loads its U units of every operand together, forms the statistics / sums from registers and writes the results, so
// the launch costs one memory round trip instead of 3 * U.  Same per-lane order and reductions as the loop kernels: same values.
template <int LANES, int U>
__global__ __launch_bounds__(256) void instnorm_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ res,
                                                               float* __restrict__ y, float* __restrict__ mean_out,
                                                               float* __restrict__ invstd_out, int NC, int C, int HW, float eps,
                                                               int act, float slope, unsigned bytes) {
    __shared__ float red[4];
    const int inst = (int)blockIdx.x * (256 / LANES) + (int)threadIdx.x / LANES;
    if (inst >= NC) return;
    const int t = (int)threadIdx.x % LANES;
    const int nv = HW >> 2;
    const nrsrc_t rx = n_rsrc(x, bytes), ry = n_rsrc(y, bytes), rr = n_rsrc(res ? res : x, res ? bytes : 0u);
    unsigned off[U];
#pragma unroll
    for (int j = 0; j < U; ++j) off[j] = t + LANES * j < nv ? ((unsigned)inst * (unsigned)HW + 4u * (unsigned)(t + LANES * j)) * 4u : NOOB;
    float4 a[U], r[U];
#pragma unroll
    for (int j = 0; j < U; ++j) a[j] = n_load4(rx, off[j]);
    if (res) {
#pragma unroll
        for (int j = 0; j < U; ++j) r[j] = n_load4(rr, off[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
    const float mu = in_reduce<LANES>(s, red) / (float)HW;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const float d0 = a[j].x - mu, d1 = a[j].y - mu, d2 = a[j].z - mu, d3 = a[j].w - mu;
        const float e = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        q += off[j] != NOOB ? e : 0.f;
    }
    const float is = rsqrtf(in_reduce<LANES>(q, red) / (float)HW + eps);
    if (t == 0) {
        mean_out[inst] = mu;
        invstd_out[inst] = is;
    }
    const int c = inst % C;
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float sh = (beta ? beta[c] : 0.f) - mu * gs;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        float4 o;
        o.x = a[j].x * gs + sh; o.y = a[j].y * gs + sh; o.z = a[j].z * gs + sh; o.w = a[j].w * gs + sh;
        if (res) { o.x += r[j].x; o.y += r[j].y; o.z += r[j].z; o.w += r[j].w; }
        o.x = rg_apply_act(o.x, act, slope); o.y = rg_apply_act(o.y, act, slope);
        o.z = rg_apply_act(o.z, act, slope); o.w = rg_apply_act(o.w, act, slope);
        n_store4(ry, off[j], o);
    }
}

template <int LANES, int U>
__global__ __launch_bounds__(256) void instnorm_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ yact, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               float* __restrict__ dx, float* __restrict__ dres,
                                                               float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat,
                                                               float* __restrict__ sum_dx, int NC, int C, int HW, int act,
                                                               float slope, unsigned bytes) {
    __shared__ float red[4];
    const int inst = (int)blockIdx.x * (256 / LANES) + (int)threadIdx.x / LANES;
    if (inst >= NC) return;
    const int t = (int)threadIdx.x % LANES;
    const int nv = HW >> 2;
    const bool has_act = act != RG_ACT_NONE;
    const nrsrc_t rx = n_rsrc(x, bytes), rg = n_rsrc(dy, bytes), ry = n_rsrc(has_act ? yact : dy, has_act ? bytes : 0u);
    const nrsrc_t rdx = n_rsrc(dx ? dx : dres, dx ? bytes : 0u), rdr = n_rsrc(dres ? dres : dx, dres ? bytes : 0u);
    const float mu = mean[inst], is = invstd[inst];
    unsigned off[U];
#pragma unroll
    for (int j = 0; j < U; ++j) off[j] = t + LANES * j < nv ? ((unsigned)inst * (unsigned)HW + 4u * (unsigned)(t + LANES * j)) * 4u : NOOB;
    float4 g[U], v[U], yv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        g[j] = n_load4(rg, off[j]);
        v[j] = n_load4(rx, off[j]);
        if (has_act) yv[j] = n_load4(ry, off[j]);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        if (has_act) {
            g[j].x *= act_grad_from_out(yv[j].x, act, slope); g[j].y *= act_grad_from_out(yv[j].y, act, slope);
            g[j].z *= act_grad_from_out(yv[j].z, act, slope); g[j].w *= act_grad_from_out(yv[j].w, act, slope);
        }
        s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s2 += (g[j].x * (v[j].x - mu) + g[j].y * (v[j].y - mu)) + (g[j].z * (v[j].z - mu) + g[j].w * (v[j].w - mu));
    }
    s1 = in_reduce<LANES>(s1, red);
    s2 = in_reduce<LANES>(s2, red) * is;
    if (t == 0) {
        sum_dy[inst] = s1;
        sum_dy_xhat[inst] = s2;
    }
    const float gs = (gamma ? gamma[inst % C] : 1.f) * is;
    const float a = s1 / (float)HW, b = s2 / (float)HW * is;
    float sdx = 0.f;
#pragma unroll
    for (int j = 0; j < U; ++j) {
        if (dres) n_store4(rdr, off[j], g[j]);
        if (dx) {
            float4 o;
            o.x = gs * (g[j].x - a - (v[j].x - mu) * b); o.y = gs * (g[j].y - a - (v[j].y - mu) * b);
            o.z = gs * (g[j].z - a - (v[j].z - mu) * b); o.w = gs * (g[j].w - a - (v[j].w - mu) * b);
            n_store4(rdx, off[j], o);
            sdx += off[j] != NOOB ? (o.x + o.y) + (o.z + o.w) : 0.f;
        }
    }
    if (sum_dx) {
        sdx = in_reduce<LANES>(sdx, red);
        if (t == 0) sum_dx[inst] = sdx;
    }
}

// out[c] = sum over n and the pixels of dy[n][c][.] — the bias gradient of a convolution / linear layer — in ONE launch when a
// channel holds at most 32768 values (one workgroup per channel; eight loads in flight per thread; fixed order).  The two-stage
// slice reduction (rg_bn_bwd_reduce on unit statistics) stays for the large maps.
template <bool VEC>
__global__ __launch_bounds__(256) void channel_sum_small_kernel(const float* __restrict__ dy, float* __restrict__ out, int N, int C,
                                                                int HW, unsigned bytes) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int rl = VEC ? HW >> 2 : HW;
    const int total = N * rl;
    const nrsrc_t rg = n_rsrc(dy, bytes);
    float s = 0.f;
    for (int i0 = t; i0 < total; i0 += 256 * 8) {
        unsigned off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + 256 * j;
            const int n = i / rl, v = i - n * rl;
            off[j] = i < total ? (unsigned)((n * C + c) * HW + (VEC ? 4 * v : v)) * 4u : NOOB;
        }
        if (VEC) {
            float4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = n_load4(rg, off[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (a[j].x + a[j].y) + (a[j].z + a[j].w);
        } else {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, off[j], 0, 0));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += a[j];
        }
    }
    s = bn_block_sum(s, red);
    if (t == 0) out[c] = s;
}

// units (float4) per thread of the register kernels for this geometry, 0 = use the loop kernels
static int bn_reg_units(int N, int C, int HW) {
    if ((HW & 3) || (int64_t)N * C * HW * 4 >= (1ll << 31)) return 0;
    const int total = N * (HW >> 2);
    const int per = (total + 255) / 256;
    if (per > 16) return 0;
    return per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : per <= 8 ? 8 : 16;
}

// g = dy * act'(y); sum_dy[c] = sum g, sum_dy_xhat[c] = sum g * xhat; dx = gamma * invstd * (g - mean(g) - xhat * mean(g xhat)); dres = g
__global__ __launch_bounds__(256) void bn_train_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ yact, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 float* __restrict__ dx, float* __restrict__ dres,
                                                                 float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat, int N,
                                                                 int C, int HW, int act, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const bool vec = (HW & 3) == 0;
    const int rl = vec ? HW >> 2 : HW;
    const int total = N * rl;
    const int64_t cs = (int64_t)C * HW;
    const float cnt = (float)N * (float)HW;
    const float mu = mean[c], is = invstd[c];
    const bool has_act = act != RG_ACT_NONE;
    float s1 = 0.f, s2 = 0.f;
    for (int i = t; i < total; i += 256) {
        const int n = i / rl, v = i - n * rl;
        const int64_t off = (int64_t)n * cs + (int64_t)c * HW;
        if (vec) {
            float4 g = reinterpret_cast<const float4*>(dy + off)[v];
            if (has_act) {
                const float4 yv = reinterpret_cast<const float4*>(yact + off)[v];
                g.x *= act_grad_from_out(yv.x, act, slope); g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope); g.w *= act_grad_from_out(yv.w, act, slope);
            }
            const float4 a = reinterpret_cast<const float4*>(x + off)[v];
            s1 += (g.x + g.y) + (g.z + g.w);
            s2 += (g.x * (a.x - mu) + g.y * (a.y - mu)) + (g.z * (a.z - mu) + g.w * (a.w - mu));
        } else {
            float g = dy[off + v];
            if (has_act) g *= act_grad_from_out(yact[off + v], act, slope);
            s1 += g;
            s2 += g * (x[off + v] - mu);
        }
    }
    s1 = bn_block_sum(s1, red);
    s2 = bn_block_sum(s2, red) * is;
    if (t == 0) {
        sum_dy[c] = s1;
        sum_dy_xhat[c] = s2;
    }
    if (!dx && !dres) return;
    const float gs = (gamma ? gamma[c] : 1.f) * is;
    const float a0 = s1 / cnt, b0 = s2 / cnt * is;
    for (int i = t; i < total; i += 256) {
        const int n = i / rl, v = i - n * rl;
        const int64_t off = (int64_t)n * cs + (int64_t)c * HW;
        if (vec) {
            float4 g = reinterpret_cast<const float4*>(dy + off)[v];
            if (has_act) {
                const float4 yv = reinterpret_cast<const float4*>(yact + off)[v];
                g.x *= act_grad_from_out(yv.x, act, slope); g.y *= act_grad_from_out(yv.y, act, slope);
                g.z *= act_grad_from_out(yv.z, act, slope); g.w *= act_grad_from_out(yv.w, act, slope);
            }
            if (dres) reinterpret_cast<float4*>(dres + off)[v] = g;
            if (dx) {
                const float4 a = reinterpret_cast<const float4*>(x + off)[v];
                float4 o;
                o.x = gs * (g.x - a0 - (a.x - mu) * b0); o.y = gs * (g.y - a0 - (a.y - mu) * b0);
                o.z = gs * (g.z - a0 - (a.z - mu) * b0); o.w = gs * (g.w - a0 - (a.w - mu) * b0);
                reinterpret_cast<float4*>(dx + off)[v] = o;
            }
        } else {
            float g = dy[off + v];
            if (has_act) g *= act_grad_from_out(yact[off + v], act, slope);
            if (dres) dres[off + v] = g;
            if (dx) dx[off + v] = gs * (g - a0 - (x[off + v] - mu) * b0);
        }
    }
}

// out_a[c] = sum_n a[n][c], out_b[c] = sum_n b[n][c] (fixed summation order): the affine gradients of an InstanceNorm from the per-(n,c) sums
// its backward reduction already produced.  Either pair may be NULL.
__global__ __launch_bounds__(256) void rows_sum_pair_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ oa, float* __restrict__ ob, int N, int C) {
    // 16 channels x 16 row lanes per workgroup: lane ny sums rows ny, ny + 16, ...; the 16 lane sums are added in lane order
    __shared__ float sa[16][17], sb[16][17];
    const int cx = threadIdx.x & 15, ny = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float x = 0.f, y = 0.f;
    if (c < C) {
        for (int n = ny; n < N; n += 16) {
            if (a) x += a[(int64_t)n * C + c];
            if (b) y += b[(int64_t)n * C + c];
        }
    }
    sa[ny][cx] = x;
    sb[ny][cx] = y;
    __syncthreads();
    if (ny == 0 && c < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ta += sa[i][cx];
            tb += sb[i][cx];
        }
        if (a) oa[c] = ta;
        if (b) ob[c] = tb;
    }
}

// 1 when rg_channel_sum runs as one launch for this geometry (else use rg_bn_bwd_reduce on unit statistics)
extern "C" size_t rg_channel_sum_ok(int N, int C, int HW) {
    return (int64_t)N * HW <= 32768 && (int64_t)N * C * HW * 4 < (1ll << 31) ? 1 : 0;
}

extern "C" int rg_channel_sum(const float* dy, float* out, int N, int C, int HW, hipStream_t stream) {
    RG_REQUIRE(dy && out && N > 0 && C > 0 && HW > 0, "rg_channel_sum: bad arguments");
    RG_REQUIRE(rg_channel_sum_ok(N, C, HW), "rg_channel_sum: %d x %d values per channel need the two-stage reduction", N, HW);
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 4.0 * N * (double)C * HW);
    const unsigned bytes = (unsigned)((int64_t)N * C * HW * 4);
    if ((HW & 3) == 0)
        hipLaunchKernelGGL(channel_sum_small_kernel<true>, dim3(C), dim3(256), 0, stream, dy, out, N, C, HW, bytes);
    else
        hipLaunchKernelGGL(channel_sum_small_kernel<false>, dim3(C), dim3(256), 0, stream, dy, out, N, C, HW, bytes);
    return rg::check_launch("rg_channel_sum");
}

extern "C" int rg_rows_sum_pair(const float* a, const float* b, float* out_a, float* out_b, int N, int C,
                                hipStream_t stream) {
    RG_REQUIRE((a || b) && (!a || out_a) && (!b || out_b) && N > 0 && C > 0, "rg_rows_sum_pair: bad arguments");
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 4.0 * ((a ? 1 : 0) + (b ? 1 : 0)) * (double)(N + 1) * C);
    hipLaunchKernelGGL(rows_sum_pair_kernel, dim3(rg::cdiv(C, 16)), dim3(256), 0, stream, a, b, out_a, out_b, N, C);
    return rg::check_launch("rg_rows_sum_pair");
}

// Train-mode BatchNorm, one launch per direction (see the kernels): rg_bn_train_fused_ok says whether the geometry qualifies;
// otherwise use rg_bn_stats + rg_bn_apply_fwd / rg_bn_bwd_reduce + rg_bn_bwd_apply.
// development switch: RG_BN_REG=0 keeps the loop kernels (A/B timing)
static const bool g_bn_reg = !(getenv("RG_BN_REG") && atoi(getenv("RG_BN_REG")) == 0);

extern "C" size_t rg_bn_train_fused_ok(int N, int C, int HW) {
    return (int64_t)N * HW <= 16384 && C >= 128 ? 1 : 0;
}

extern "C" int rg_bn_train_fwd_fused(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                     float* mean, float* invstd, float* running_mean, float* running_var, int N, int C, int HW,
                                     float eps, float momentum, int act, float slope, hipStream_t stream) {
    RG_REQUIRE(x && y && mean && invstd && N > 0 && C > 0 && HW > 0, "rg_bn_train_fwd_fused: bad arguments");
    RG_REQUIRE((int64_t)N * HW < (1ll << 30), "rg_bn_train_fwd_fused: N*HW too large");
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, (residual ? 12.0 : 8.0) * N * (double)C * HW);
    const unsigned bytes = (unsigned)((int64_t)N * C * HW * 4);
#define RG_BNF(U_)                                                                                                             \
    hipLaunchKernelGGL(bn_train_fwd_reg_kernel<U_>, dim3(C), dim3(256), 0, stream, x, gamma, beta, residual, y, mean, invstd,   \
                       running_mean, running_var, N, C, HW, eps, momentum, act, slope, bytes)
    switch (g_bn_reg ? bn_reg_units(N, C, HW) : 0) {
        case 1: RG_BNF(1); break;
        case 2: RG_BNF(2); break;
        case 4: RG_BNF(4); break;
        case 8: RG_BNF(8); break;
        case 16: RG_BNF(16); break;
        default:
            hipLaunchKernelGGL(bn_train_fwd_fused_kernel, dim3(C), dim3(256), 0, stream, x, gamma, beta, residual, y, mean, invstd,
                               running_mean, running_var, N, C, HW, eps, momentum, act, slope);
    }
#undef RG_BNF
    return rg::check_launch("rg_bn_train_fwd_fused");
}

extern "C" int rg_bn_train_bwd_fused(const float* x, const float* dy, const float* y_act, const float* mean, const float* invstd,
                                     const float* gamma, float* dx, float* dres, float* sum_dy, float* sum_dy_xhat, int N, int C,
                                     int HW, int act, float slope, hipStream_t stream) {
    RG_REQUIRE(x && dy && mean && invstd && sum_dy && sum_dy_xhat && N > 0 && C > 0 && HW > 0, "rg_bn_train_bwd_fused: bad arguments");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_bn_train_bwd_fused: fused activation needs the forward output");
    RG_REQUIRE((int64_t)N * HW < (1ll << 30), "rg_bn_train_bwd_fused: N*HW too large");
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, ((act ? 12.0 : 8.0) + (dx ? 4.0 : 0.0) + (dres ? 4.0 : 0.0)) * N * (double)C * HW);
    const unsigned bytes = (unsigned)((int64_t)N * C * HW * 4);
#define RG_BNB(U_)                                                                                                             \
    hipLaunchKernelGGL(bn_train_bwd_reg_kernel<U_>, dim3(C), dim3(256), 0, stream, x, dy, y_act, mean, invstd, gamma, dx, dres,  \
                       sum_dy, sum_dy_xhat, N, C, HW, act, slope, bytes)
    switch (g_bn_reg ? bn_reg_units(N, C, HW) : 0) {
        case 1: RG_BNB(1); break;
        case 2: RG_BNB(2); break;
        case 4: RG_BNB(4); break;
        case 8: RG_BNB(8); break;
        case 16: RG_BNB(16); break;
        default:
            hipLaunchKernelGGL(bn_train_bwd_fused_kernel, dim3(C), dim3(256), 0, stream, x, dy, y_act, mean, invstd, gamma, dx, dres,
                               sum_dy, sum_dy_xhat, N, C, HW, act, slope);
    }
#undef RG_BNB
    return rg::check_launch("rg_bn_train_bwd_fused");
}

// lanes per instance: at least ~2 load units (float4 when HW % 4 == 0) per lane, one workgroup above 2048 elements
static int in_lanes(int HW) {
    if (HW > 2048) return 256;
    const int units = (HW & 3) ? HW : HW >> 2;
    if (units <= 32) return 16;
    if (units <= 64) return 32;
    return 64;
}

// InstanceNorm2d forward: y = act(gamma[c] * (x - mean[n,c]) * invstd[n,c] + beta[c] + residual); mean / invstd [N*C] are kept for
// the backward.  One launch.
extern "C" int rg_instnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                               float* mean, float* invstd, int N, int C, int HW, float eps, int act, float slope,
                               hipStream_t stream) {
    RG_REQUIRE(x && y && mean && invstd && N > 0 && C > 0 && HW > 0, "rg_instnorm_fwd: bad arguments");
    RG_REQUIRE((int64_t)N * C < (1ll << 31), "rg_instnorm_fwd: N*C exceeds 2^31");
    const int NC = N * C;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, (residual ? 12.0 : 8.0) * (double)NC * HW);
    const int lanes = in_lanes(HW);
    const int per = (HW & 3) ? 0 : ((HW >> 2) + lanes - 1) / lanes;
    if (g_bn_reg && per >= 1 && per <= 8 && (int64_t)NC * HW * 4 < (1ll << 31)) {
        const unsigned bytes = (unsigned)((int64_t)NC * HW * 4);
        const int U = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
#define RG_INF(L_, U_)                                                                                                             \
    case L_ * 16 + U_:                                                                                                             \
        hipLaunchKernelGGL((instnorm_fwd_reg_kernel<L_, U_>), dim3(rg::cdiv(NC, 256 / L_)), dim3(256), 0, stream, x, gamma, beta,    \
                           residual, y, mean, invstd, NC, C, HW, eps, act, slope, bytes);                                          \
        break
        switch (lanes * 16 + U) {
            RG_INF(16, 1); RG_INF(16, 2); RG_INF(32, 1); RG_INF(32, 2); RG_INF(64, 1); RG_INF(64, 2); RG_INF(64, 4); RG_INF(64, 8);
            RG_INF(256, 1); RG_INF(256, 2); RG_INF(256, 4); RG_INF(256, 8);
            default: rg::set_error("rg_instnorm_fwd: no register kernel for %d lanes x %d units", lanes, U); return RG_ERR_INVALID;
        }
#undef RG_INF
        return rg::check_launch("rg_instnorm_fwd");
    }
    switch (lanes) {
        case 16: hipLaunchKernelGGL(instnorm_fwd_kernel<16>, dim3(rg::cdiv(NC, 16)), dim3(256), 0, stream, x, gamma, beta, residual, y,
                                    mean, invstd, NC, C, HW, eps, act, slope); break;
        case 32: hipLaunchKernelGGL(instnorm_fwd_kernel<32>, dim3(rg::cdiv(NC, 8)), dim3(256), 0, stream, x, gamma, beta, residual, y,
                                    mean, invstd, NC, C, HW, eps, act, slope); break;
        case 64: hipLaunchKernelGGL(instnorm_fwd_kernel<64>, dim3(rg::cdiv(NC, 4)), dim3(256), 0, stream, x, gamma, beta, residual, y,
                                    mean, invstd, NC, C, HW, eps, act, slope); break;
        default: hipLaunchKernelGGL(instnorm_fwd_kernel<256>, dim3(NC), dim3(256), 0, stream, x, gamma, beta, residual, y, mean,
                                    invstd, NC, C, HW, eps, act, slope);
    }
    return rg::check_launch("rg_instnorm_fwd");
}

// InstanceNorm2d backward in one launch: dx, dres (either may be NULL) and the per-instance sums [N*C] of g = dy*act'(y) and
// g*xhat (dbeta / dgamma = their sums over n: rg_rows_sum_pair); sum_dx (may be NULL) receives the per-instance sums of dx — the
// bias gradient of a convolution that feeds this layer, without another pass over dx.
extern "C" int rg_instnorm_bwd(const float* x, const float* dy, const float* y_act, const float* mean, const float* invstd,
                               const float* gamma, float* dx, float* dres, float* sum_dy, float* sum_dy_xhat, float* sum_dx, int N,
                               int C, int HW, int act, float slope, hipStream_t stream) {
    RG_REQUIRE(x && dy && mean && invstd && sum_dy && sum_dy_xhat && N > 0 && C > 0 && HW > 0, "rg_instnorm_bwd: bad arguments");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_instnorm_bwd: fused activation needs the forward output");
    RG_REQUIRE((int64_t)N * C < (1ll << 31), "rg_instnorm_bwd: N*C exceeds 2^31");
    const int NC = N * C;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, ((act ? 12.0 : 8.0) + (dx ? 4.0 : 0.0) + (dres ? 4.0 : 0.0)) * (double)NC * HW);
    const int lanes = in_lanes(HW);
    const int per = (HW & 3) ? 0 : ((HW >> 2) + lanes - 1) / lanes;
    if (g_bn_reg && per >= 1 && per <= 8 && (int64_t)NC * HW * 4 < (1ll << 31)) {
        const unsigned bytes = (unsigned)((int64_t)NC * HW * 4);
        const int U = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
#define RG_INB(L_, U_)                                                                                                             \
    case L_ * 16 + U_:                                                                                                             \
        hipLaunchKernelGGL((instnorm_bwd_reg_kernel<L_, U_>), dim3(rg::cdiv(NC, 256 / L_)), dim3(256), 0, stream, x, dy, y_act,     \
                           mean, invstd, gamma, dx, dres, sum_dy, sum_dy_xhat, sum_dx, NC, C, HW, act, slope, bytes);              \
        break
        switch (lanes * 16 + U) {
            RG_INB(16, 1); RG_INB(16, 2); RG_INB(32, 1); RG_INB(32, 2); RG_INB(64, 1); RG_INB(64, 2); RG_INB(64, 4); RG_INB(64, 8);
            RG_INB(256, 1); RG_INB(256, 2); RG_INB(256, 4); RG_INB(256, 8);
            default: rg::set_error("rg_instnorm_bwd: no register kernel for %d lanes x %d units", lanes, U); return RG_ERR_INVALID;
        }
#undef RG_INB
        return rg::check_launch("rg_instnorm_bwd");
    }
    switch (lanes) {
        case 16: hipLaunchKernelGGL(instnorm_bwd_kernel<16>, dim3(rg::cdiv(NC, 16)), dim3(256), 0, stream, x, dy, y_act, mean, invstd,
                                    gamma, dx, dres, sum_dy, sum_dy_xhat, sum_dx, NC, C, HW, act, slope); break;
        case 32: hipLaunchKernelGGL(instnorm_bwd_kernel<32>, dim3(rg::cdiv(NC, 8)), dim3(256), 0, stream, x, dy, y_act, mean, invstd,
                                    gamma, dx, dres, sum_dy, sum_dy_xhat, sum_dx, NC, C, HW, act, slope); break;
        case 64: hipLaunchKernelGGL(instnorm_bwd_kernel<64>, dim3(rg::cdiv(NC, 4)), dim3(256), 0, stream, x, dy, y_act, mean, invstd,
                                    gamma, dx, dres, sum_dy, sum_dy_xhat, sum_dx, NC, C, HW, act, slope); break;
        default: hipLaunchKernelGGL(instnorm_bwd_kernel<256>, dim3(NC), dim3(256), 0, stream, x, dy, y_act, mean, invstd, gamma, dx,
                                    dres, sum_dy, sum_dy_xhat, sum_dx, NC, C, HW, act, slope);
    }
    return rg::check_launch("rg_instnorm_bwd");
}

// dx (may be NULL) and dres (may be NULL; the gradient of a fused residual input = dy*act'(y)).
extern "C" int rg_bn_bwd_apply(const float* x, const float* dy, const float* y_act, const float* mean,
                               const float* stat, const float* gamma, const float* sum_dy, const float* sum_dy_xhat,
                               float* dx, float* dres, int N, int C, int HW, int train, int stat_is_var, float eps,
                               int act, float slope, hipStream_t stream) {
    RG_REQUIRE(dy && mean && stat && (dx || dres), "rg_bn_bwd_apply: null tensor");
    RG_REQUIRE(!train || (x && sum_dy && sum_dy_xhat), "rg_bn_bwd_apply: train mode needs x and the channel sums");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_bn_bwd_apply: fused activation needs the forward output");
    const int64_t total = (int64_t)N * C * HW;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 16.0 * total);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for((HW & 3) ? total : total / 4)), dim3(256), 0, stream, x, dy,
                       y_act, mean, stat, gamma, sum_dy, sum_dy_xhat, dx, dres, total, C, HW,
                       1.f / (float)((int64_t)N * HW), train, stat_is_var, eps, act, slope);
    return rg::check_launch("rg_bn_bwd_apply");
}

// Eval-mode (running statistics) backward in one pass: dx (may be NULL), dres (may be NULL) and, when sum_dy /
// sum_dy_xhat are given, the channel sums for dbeta / dgamma (FD-GAN trains the affine parameters of E's and D_id's
// frozen BatchNorms: FD/fdgan/model.py:72-85).
extern "C" int rg_bn_eval_bwd(const float* x, const float* dy, const float* y_act, const float* running_mean,
                              const float* running_var, const float* gamma, float* dx, float* dres, float* sum_dy,
                              float* sum_dy_xhat, int N, int C, int HW, float eps, int act, float slope,
                              void* workspace, size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(dy && running_mean && running_var && (dx || dres || sum_dy), "rg_bn_eval_bwd: null tensor");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_bn_eval_bwd: fused activation needs the forward output");
    const int want = (sum_dy && sum_dy_xhat) ? 1 : 0;
    RG_REQUIRE(!want || x, "rg_bn_eval_bwd: the channel sums need x");
    int L;
    const int S = pick_slices(N, C, HW, &L);
    if (want && (!workspace || workspace_bytes < (size_t)C * S * 2 * sizeof(float))) {
        rg::set_error("rg_bn_eval_bwd: workspace too small");
        return RG_ERR_WORKSPACE;
    }
    float* part = static_cast<float*>(workspace);
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, (want ? 12.0 : 8.0) * N * (double)C * HW + (dx ? 4.0 : 0.0) * N * (double)C * HW +
                                                         (dres ? 4.0 : 0.0) * N * (double)C * HW);
    hipLaunchKernelGGL(bn_eval_bwd_fused_kernel, dim3(S, C), dim3(256), 0, stream, x, dy, y_act, running_mean,
                       running_var, gamma, dx, dres, part, N, C, HW, L, eps, act, slope, want);
    if (want)
        hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3(rg::cdiv(C, 64)), dim3(64), 0, stream, part, C, S, sum_dy,
                           sum_dy_xhat);
    return rg::check_launch("rg_bn_eval_bwd");
}

// ---- entry points of the conv + frozen-BatchNorm fold (see the kernel comments above) ------------------------
extern "C" int rg_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          float eps, float* scale, float* shift, float* invstd, int C, hipStream_t stream) {
    RG_REQUIRE(running_mean && running_var && scale && shift && invstd && C > 0, "rg_bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(rg::cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, running_mean, running_var,
                       eps, scale, shift, invstd, C);
    return rg::check_launch("rg_bn_fold");
}

extern "C" int rg_act_bwd_sum(const float* dy, const float* y_act, float* g, float* sum_g, int N, int C, int HW, int act,
                              float slope, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    RG_REQUIRE(dy && (g || sum_g) && N > 0 && C > 0 && HW > 0, "rg_act_bwd_sum: bad arguments");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_act_bwd_sum: the activation gradient needs the forward output");
    int L;
    const int S = pick_slices(N, C, HW, &L);
    if (sum_g && (!workspace || workspace_bytes < (size_t)C * S * sizeof(float))) {
        rg::set_error("rg_act_bwd_sum: workspace too small");
        return RG_ERR_WORKSPACE;
    }
    const double el = (double)N * C * HW;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 4.0 * el * (1.0 + (act != RG_ACT_NONE ? 1.0 : 0.0) + (g ? 1.0 : 0.0)));
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(act_bwd_sum_kernel, dim3(S, C), dim3(256), 0, stream, dy, y_act, g, part, N, C, HW, L, act, slope,
                       sum_g ? 1 : 0);
    if (sum_g) hipLaunchKernelGGL(sum_slices_kernel, dim3(rg::cdiv(C, 64)), dim3(64), 0, stream, part, C, S, sum_g);
    return rg::check_launch("rg_act_bwd_sum");
}

extern "C" int rg_bn_fold_wgrad(const float* w, float* g, const float* scale, const float* invstd, const float* running_mean,
                                const float* sum_g, const float* partials, int n_slices, float* dbeta, float* dgamma, int K,
                                int M, hipStream_t stream) {
    RG_REQUIRE(w && g && scale && K > 0 && M > 0, "rg_bn_fold_wgrad: bad arguments");
    RG_REQUIRE(!dgamma || (invstd && running_mean && (sum_g || partials)), "rg_bn_fold_wgrad: dgamma needs invstd, mean and the sums");
    RG_REQUIRE(!partials || n_slices > 0, "rg_bn_fold_wgrad: partials need their slice count");
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 12.0 * K * (double)M);
    hipLaunchKernelGGL(bn_fold_wgrad_kernel, dim3(K), dim3(256), 0, stream, w, g, scale, invstd, running_mean, sum_g, partials,
                       n_slices, dbeta, dgamma, M);
    return rg::check_launch("rg_bn_fold_wgrad");
}

// slice partials only (no finalize launch): part[C][rg_bn_slices(N, C, HW)], consumed by rg_bn_fold_wgrad
extern "C" int rg_bn_slices(int N, int C, int HW) {
    int L;
    return pick_slices(N, C, HW, &L);
}

extern "C" int rg_act_bwd_partial(const float* dy, const float* y_act, float* g, float* part, int N, int C, int HW, int act,
                                  float slope, hipStream_t stream) {
    RG_REQUIRE(dy && part && N > 0 && C > 0 && HW > 0, "rg_act_bwd_partial: bad arguments");
    RG_REQUIRE(act == RG_ACT_NONE || y_act, "rg_act_bwd_partial: the activation gradient needs the forward output");
    int L;
    const int S = pick_slices(N, C, HW, &L);
    const double el = (double)N * C * HW;
    rg::ProfScope prof(rg::FAM_NORM, stream, 0.0, 4.0 * el * (1.0 + (act != RG_ACT_NONE ? 1.0 : 0.0) + (g ? 1.0 : 0.0)));
    hipLaunchKernelGGL(act_bwd_sum_kernel, dim3(S, C), dim3(256), 0, stream, dy, y_act, g, part, N, C, HW, L, act, slope, 1);
    return rg::check_launch("rg_act_bwd_partial");
}

extern "C" int rg_scale_rows(const float* w, const float* scale, float* out, int K, int M, hipStream_t stream) {
    RG_REQUIRE(w && scale && out && K > 0 && M > 0, "rg_scale_rows: bad arguments");
    const int64_t total = (int64_t)K * M;
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(total)), dim3(256), 0, stream, w, scale, out, M, total);
    return rg::check_launch("rg_scale_rows");
}

extern "C" int rg_fold_filters_multi(const void* table, int n_pairs, int total_blocks, hipStream_t stream) {
    RG_REQUIRE(table && n_pairs > 0 && total_blocks > 0, "rg_fold_filters_multi: bad arguments");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 12.0 * (double)total_blocks * FOLD_CHUNK);
    hipLaunchKernelGGL(fold_filters_multi_kernel, dim3(total_blocks), dim3(256), 0, stream,
                       static_cast<const long long*>(table), n_pairs);
    return rg::check_launch("rg_fold_filters_multi");
}

extern "C" int rg_fold_chunk(void) { return FOLD_CHUNK; }
