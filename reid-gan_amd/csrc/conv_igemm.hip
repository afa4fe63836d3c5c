// Implicit-GEMM 2-D convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient,
// fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fma chain).
//
// Replaces what cuDNN/ATen do for the reference's nn.Conv2d / nn.ConvTranspose2d layers:
//   ResNet-50 trunk           CC/clustercontrast/models/resnet_ibn_a.py:70-159 (layout pin), FD/reid/models/resnet.py:65-75
//   CustomPoseGenerator       FD/fdgan/networks.py:86-138  (Conv 4x4/2, (8,4) valid, ConvTranspose 4x4/2, (8,4))
//   NLayerDiscriminator       FD/fdgan/networks.py:206-232 (Conv 4x4/2, 4x4/1)
// ConvTranspose2d forward == dgrad of the conv with the same weight tensor; its dgrad == conv fwd.
//
// GEMM views (all tensors NCHW fp32, weights [K][C][KH][KW]):
//   fwd   : y[K, N*P*Q]      = w[K, C*KH*KW]          x im2col(x)[C*KH*KW, N*P*Q]
//   dgrad : dx[C, N*Hc*Wc]   = w^T[C, K*taps]         x gather(dy)[K*taps, N*Hc*Wc]   per stride-parity class
//   wgrad : dw[K, C*KH*KW]   = dy[K, N*P*Q]           x im2col(x)^T[N*P*Q, C*KH*KW]   split over N*P*Q
//
// Work decomposition: 256-thread workgroup = 4 wave64; block tile BM x BN x 16, each wave owns a
// (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles.  Operand tiles are staged global -> VGPR ->
// LDS, k-major ([16][BM+4] / [16][BN+4]) so that the MFMA operand reads (lane = row, lane>>5 = k)
// are bank-conflict free ds_read_b32; LDS is double buffered, one barrier per k-tile, and the
// next tile's global loads are issued before the current tile's MFMAs.  The flat tile id is
// remapped so that every XCD (private 4 MiB L2) works on a contiguous range of pixel tiles.
#include "rg_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int LPAD = 4;
constexpr int NT = 256;

struct FastDiv {
    unsigned mul;
    unsigned shr;
    unsigned d;
};

static FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d ? d : 1;
    if (f.d == 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    unsigned l = 0;
    while ((1ull << l) < f.d) ++l;  // ceil(log2 d)
    const unsigned p = 31 + l;
    f.mul = (unsigned)(((1ull << p) + f.d - 1) / f.d);
    f.shr = p - 32;
    return f;
}

__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return f.d == 1 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shr);
}

struct Epilogue {
    const float* scale;  // per output channel (GEMM row) or nullptr
    const float* shift;  // per output channel or nullptr
    const float* res;    // same shape as the output or nullptr
    int act;
    float slope;
};

struct ConvP {
    const float* x;   // fwd: input, dgrad: dy, wgrad: input
    const float* w;   // fwd/dgrad: weights, wgrad: dy
    float* y;         // fwd: y, dgrad: dx, wgrad: dw or workspace
    Epilogue ep;
    int N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q;
    int M, Ng, Kg;
    int a_vec4;
    int m_tiles, n_tiles;
    FastDiv d_rs, d_kw, d_pq, d_q;
    // wgrad
    int ktiles_per_split, splits;
};

struct DgradClass {
    int r0, s0, nrh, nrw, Hc, Wc, Ngc, Kgc, ntiles;
    FastDiv d_taps, d_nrw, d_hw, d_w;
};

struct DgradP {
    ConvP c;
    DgradClass cls[4];
};

template <int BM, int BN, int WM, int WN>
struct Tile {
    static constexpr int LDA = BM + LPAD;
    static constexpr int LDB = BN + LPAD;
    static constexpr int WTM = BM / WM;
    static constexpr int WTN = BN / WN;
    static constexpr int TM = WTM / 32;
    static constexpr int TN = WTN / 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA tile");
    // register staging sizes
    static constexpr int ACNT = (BM * BK / NT) < 4 ? 4 : (BM * BK / NT);
    static constexpr int BCNT = (BN * BK / NT) < 1 ? 1 : (BN * BK / NT);
};

// XCD-aware bijective remap of the flat block id: blocks b, b+8, b+16.. share an XCD (and its L2),
// so give each XCD a contiguous chunk of the tile space.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <typename T>
__device__ __forceinline__ void mma_tile(const float (*As)[T::LDA], const float (*Bs)[T::LDB],
                                         floatx16 (&acc)[T::TM][T::TN], int wm, int wn, int lane) {
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
        const int k = 2 * ks + kh;
        float a[T::TM], b[T::TN];
#pragma unroll
        for (int i = 0; i < T::TM; ++i) a[i] = As[k][wm * T::WTM + i * 32 + l32];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) b[j] = Bs[k][wn * T::WTN + j * 32 + l32];
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    __shared__ float As[2][BK][T::LDA];
    __shared__ float Bs[2][BK][T::LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // B (im2col gather): fixed pixel column per thread, lanes run along pixels (coalesced rows)
    constexpr int BKSTEP = NT / BN > 0 ? NT / BN : 1;
    const int bcol = tid % BN, bk0 = tid / BN;
    const int n = n0 + bcol;
    const bool bvalid = n < p.Ng;
    int img = 0, h0 = 0, w0 = 0;
    if (bvalid) {
        img = fdiv(n, p.d_pq);
        const int pq = n - img * p.P * p.Q;
        const int pp = fdiv(pq, p.d_q);
        const int qq = pq - pp * p.Q;
        h0 = pp * p.SH - p.PH;
        w0 = qq * p.SW - p.PW;
    }
    const int HW = p.H * p.W;
    const float* xb = p.x + (int64_t)img * p.C * HW;
    const int pix_off = h0 * p.W + w0;
    const int RS = p.KH * p.KW;

    float ra[T::ACNT], rb[T::BCNT];
    floatx16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        // A: weights [M][Kg], contiguous along k
        if (p.a_vec4) {
#pragma unroll
            for (int i = 0; i < T::ACNT / 4; ++i) {
                const int v = tid + NT * i;
                const int row = v >> 2, kq = (v & 3) * 4;
                const int m = m0 + row, k = kbase + kq;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (v < BM * 4 && m < p.M && k < p.Kg)
                    t = *reinterpret_cast<const float4*>(p.w + (int64_t)m * p.Kg + k);
                ra[4 * i + 0] = t.x;
                ra[4 * i + 1] = t.y;
                ra[4 * i + 2] = t.z;
                ra[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::ACNT; ++i) {
                const int e = tid + NT * i;
                const int kk = e & 15, row = e >> 4;
                const int m = m0 + row, k = kbase + kk;
                ra[i] = (e < BM * BK && m < p.M && k < p.Kg) ? p.w[(int64_t)m * p.Kg + k] : 0.f;
            }
        }
        // B: gather
#pragma unroll
        for (int i = 0; i < T::BCNT; ++i) {
            const int kk = bk0 + i * BKSTEP;
            const int k = kbase + kk;
            const int c = fdiv(k, p.d_rs);
            const int rs = k - c * RS;
            const int r = fdiv(rs, p.d_kw);
            const int s = rs - r * p.KW;
            const int h = h0 + r, w = w0 + s;
            const bool ok = bvalid && k < p.Kg && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            rb[i] = ok ? xb[c * HW + r * p.W + s + pix_off] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
        if (p.a_vec4) {
#pragma unroll
            for (int i = 0; i < T::ACNT / 4; ++i) {
                const int v = tid + NT * i;
                const int row = v >> 2, kq = (v & 3) * 4;
                if (v < BM * 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[buf][kq + j][row] = ra[4 * i + j];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::ACNT; ++i) {
                const int e = tid + NT * i;
                const int kk = e & 15, row = e >> 4;
                if (e < BM * BK) As[buf][kk][row] = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < T::BCNT; ++i) Bs[buf][bk0 + i * BKSTEP][bcol] = rb[i];
    };

    const int nk = (p.Kg + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane);
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int l32 = lane & 31, kh = lane >> 5;
    const int PQ = p.P * p.Q;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        if (nn >= p.Ng) continue;
        const int im = fdiv(nn, p.d_pq);
        const int pq = nn - im * PQ;
        const int64_t obase = (int64_t)im * p.K * PQ + pq;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) {
                    float v = acc[i][j][r];
                    if (p.ep.scale) v *= p.ep.scale[m];
                    if (p.ep.shift) v += p.ep.shift[m];
                    const int64_t o = obase + (int64_t)m * PQ;
                    if (p.ep.res) v += p.ep.res[o];
                    p.y[o] = rg_apply_act(v, p.ep.act, p.ep.slope);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// data gradient (also the forward of ConvTranspose2d), one GEMM per stride-parity class
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void conv_dgrad_kernel(const DgradP dp) {
    using T = Tile<BM, BN, WM, WN>;
    __shared__ float As[2][BK][T::LDA];
    __shared__ float Bs[2][BK][T::LDB];
    const ConvP& p = dp.c;
    const int ci = blockIdx.z;
    const DgradClass& cl = dp.cls[ci];
    const int ah = ci / p.SW, aw = ci % p.SW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int nwg = p.m_tiles * cl.ntiles;
    if ((int)blockIdx.x >= nwg) return;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // B: gather of dy, fixed pixel column per thread
    constexpr int BKSTEP = NT / BN > 0 ? NT / BN : 1;
    const int bcol = tid % BN, bk0 = tid / BN;
    const int n = n0 + bcol;
    const bool bvalid = n < cl.Ngc;
    int img = 0, hb = 0, wb = 0;
    if (bvalid) {
        img = fdiv(n, cl.d_hw);
        const int rem = n - img * cl.Hc * cl.Wc;
        const int hc = fdiv(rem, cl.d_w);
        const int wc = rem - hc * cl.Wc;
        hb = (ah + p.SH * hc + p.PH - cl.r0) / p.SH;
        wb = (aw + p.SW * wc + p.PW - cl.s0) / p.SW;
    }
    const int PQ = p.P * p.Q;
    const float* dyb = p.x + (int64_t)img * p.K * PQ;
    const int taps = cl.nrh * cl.nrw;

    // A: weights w[ko][c][r][s] with GEMM row m = c: lanes run along c
    constexpr int AKSTEP = NT / BM > 0 ? NT / BM : 1;
    constexpr int ACNT = (BM * BK / NT) < 1 ? 1 : (BM * BK / NT);
    const int acol = tid % BM, ak0 = tid / BM;
    const int am = m0 + acol;
    const bool avalid = am < p.M;
    const int RS = p.KH * p.KW;

    float ra[ACNT], rb[T::BCNT];
    floatx16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
#pragma unroll
        for (int i = 0; i < ACNT; ++i) {
            const int k = kbase + ak0 + i * AKSTEP;
            const int ko = fdiv(k, cl.d_taps);
            const int t = k - ko * taps;
            const int j = fdiv(t, cl.d_nrw);
            const int jj = t - j * cl.nrw;
            const int r = cl.r0 + p.SH * j, s = cl.s0 + p.SW * jj;
            const bool ok = avalid && k < cl.Kgc;
            ra[i] = ok ? p.w[((int64_t)ko * p.C + am) * RS + r * p.KW + s] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < T::BCNT; ++i) {
            const int k = kbase + bk0 + i * BKSTEP;
            const int ko = fdiv(k, cl.d_taps);
            const int t = k - ko * taps;
            const int j = fdiv(t, cl.d_nrw);
            const int jj = t - j * cl.nrw;
            const int pp = hb - j, qq = wb - jj;
            const bool ok = bvalid && k < cl.Kgc && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
            rb[i] = ok ? dyb[ko * PQ + pp * p.Q + qq] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACNT; ++i) As[buf][ak0 + i * AKSTEP][acol] = ra[i];
#pragma unroll
        for (int i = 0; i < T::BCNT; ++i) Bs[buf][bk0 + i * BKSTEP][bcol] = rb[i];
    };

    const int nk = (cl.Kgc + BK - 1) / BK;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane);
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    const int l32 = lane & 31, kh = lane >> 5;
    const int HW = p.H * p.W;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        if (nn >= cl.Ngc) continue;
        const int im = fdiv(nn, cl.d_hw);
        const int rem = nn - im * cl.Hc * cl.Wc;
        const int hc = fdiv(rem, cl.d_w);
        const int wc = rem - hc * cl.Wc;
        const int h = ah + p.SH * hc, w = aw + p.SW * wc;
        const int64_t obase = (int64_t)im * p.C * HW + h * p.W + w;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) {
                    float v = acc[i][j][r];
                    if (p.ep.scale) v *= p.ep.scale[m];
                    if (p.ep.shift) v += p.ep.shift[m];
                    const int64_t o = obase + (int64_t)m * HW;
                    if (p.ep.res) v += p.ep.res[o];
                    p.y[o] = rg_apply_act(v, p.ep.act, p.ep.slope);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: reduction over N*P*Q split across blockIdx.z, partials to a workspace
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    __shared__ float As[2][BK][T::LDA];
    __shared__ float Bs[2][BK][T::LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int mt = blockIdx.x % p.m_tiles, nt = blockIdx.x / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.z;

    const int kk = tid & 15, r0 = tid >> 4;  // lanes run along the reduction (pixel) axis
    constexpr int ACNT = BM / 16, BCNT = BN / 16;
    const int PQ = p.P * p.Q, HW = p.H * p.W, RS = p.KH * p.KW;

    // B columns (c, r, s) handled by this thread are fixed for the whole reduction
    int coff[BCNT], crs[BCNT];
#pragma unroll
    for (int i = 0; i < BCNT; ++i) {
        const int n = n0 + r0 + 16 * i;
        if (n < p.Ng) {
            const int c = fdiv(n, p.d_rs);
            const int rs = n - c * RS;
            const int r = fdiv(rs, p.d_kw);
            const int s = rs - r * p.KW;
            coff[i] = c * HW + r * p.W + s;
            crs[i] = (r << 16) | s;
        } else {
            coff[i] = 0;
            crs[i] = -1;
        }
    }

    float ra[ACNT], rb[BCNT];
    floatx16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_tile = [&](int kt) {
        const int g = kt * BK + kk;  // global output-pixel index n*P*Q + p*Q + q
        const bool gvalid = g < p.Kg;
        int img = 0, h0 = 0, w0 = 0, pq = 0;
        if (gvalid) {
            img = fdiv(g, p.d_pq);
            pq = g - img * PQ;
            const int pp = fdiv(pq, p.d_q);
            const int qq = pq - pp * p.Q;
            h0 = pp * p.SH - p.PH;
            w0 = qq * p.SW - p.PW;
        }
        const float* dyb = p.w + (int64_t)img * p.K * PQ + pq;
#pragma unroll
        for (int i = 0; i < ACNT; ++i) {
            const int m = m0 + r0 + 16 * i;
            ra[i] = (gvalid && m < p.M) ? dyb[(int64_t)m * PQ] : 0.f;
        }
        const float* xb = p.x + (int64_t)img * p.C * HW;
        const int pix_off = h0 * p.W + w0;
#pragma unroll
        for (int i = 0; i < BCNT; ++i) {
            const int r = crs[i] >> 16, s = crs[i] & 0xffff;
            const int h = h0 + r, w = w0 + s;
            const bool ok = gvalid && crs[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            rb[i] = ok ? xb[coff[i] + pix_off] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACNT; ++i) As[buf][kk][r0 + 16 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < BCNT; ++i) Bs[buf][kk][r0 + 16 * i] = rb[i];
    };

    const int nk_total = (p.Kg + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk_total) kt_end = nk_total;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane);
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    const int l32 = lane & 31, kh = lane >> 5;
    float* out = p.y + (int64_t)split * p.M * p.Ng;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        if (nn >= p.Ng) continue;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < p.M) out[(int64_t)m * p.Ng + nn] = acc[i][j][r];
            }
        }
    }
}

// dw[i] = sum_s ws[s][i]  (deterministic split-K combine; float4 when possible)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * n + i];
    out[i] = s;
}

static void fill_common(ConvP& p, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                        int P, int Q) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.KH = KH; p.KW = KW;
    p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW; p.P = P; p.Q = Q;
    p.d_rs = make_fastdiv(KH * KW);
    p.d_kw = make_fastdiv(KW);
    p.d_pq = make_fastdiv(P * Q);
    p.d_q = make_fastdiv(Q);
    p.a_vec4 = 0;
    p.ktiles_per_split = 0;
    p.splits = 1;
}

static int validate(const char* op, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                    int P, int Q) {
    RG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && KH > 0 && KW > 0, "%s: non-positive dimension", op);
    RG_REQUIRE(SH > 0 && SW > 0 && PH >= 0 && PW >= 0 && P > 0 && Q > 0, "%s: bad stride/pad/output size", op);
    RG_REQUIRE((P - 1) * SH - PH + KH - 1 >= 0 && (Q - 1) * SW - PW + KW - 1 >= 0, "%s: inconsistent geometry", op);
    // every output pixel must start inside the padded input
    RG_REQUIRE((int64_t)(P - 1) * SH - PH < H && (int64_t)(Q - 1) * SW - PW < W, "%s: output larger than input allows", op);
    RG_REQUIRE((int64_t)N * P * Q < (1ll << 31) && (int64_t)C * KH * KW < (1ll << 31) && (int64_t)N * H * W < (1ll << 31) &&
                   (int64_t)C * H * W < (1ll << 31) && (int64_t)K * P * Q < (1ll << 31),
               "%s: dimension product exceeds 2^31", op);
    return RG_OK;
}

// tile selection: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 32x256
static int pick_tile(int M, int64_t Ng) {
    if (M <= 32) return 3;
    const int64_t big = (int64_t)rg::cdiv(M, 128) * rg::cdiv64(Ng, 128);
    if (M <= 64) return (int64_t)rg::cdiv64(Ng, 128) >= 256 ? 1 : 2;
    if (big >= 384) return 0;
    return 2;
}

}  // namespace

#define RG_TILE_DISPATCH(tile, KERNEL, grid_expr, ...)                                   \
    switch (tile) {                                                                      \
        case 0: { constexpr int BM_ = 128, BN_ = 128; auto g = grid_expr;                \
                  hipLaunchKernelGGL((KERNEL<128, 128, 2, 2>), g, dim3(NT), 0, stream, __VA_ARGS__); } break; \
        case 1: { constexpr int BM_ = 64, BN_ = 128; auto g = grid_expr;                 \
                  hipLaunchKernelGGL((KERNEL<64, 128, 2, 2>), g, dim3(NT), 0, stream, __VA_ARGS__); } break;  \
        case 2: { constexpr int BM_ = 64, BN_ = 64; auto g = grid_expr;                  \
                  hipLaunchKernelGGL((KERNEL<64, 64, 2, 2>), g, dim3(NT), 0, stream, __VA_ARGS__); } break;   \
        default: { constexpr int BM_ = 32, BN_ = 256; auto g = grid_expr;                \
                  hipLaunchKernelGGL((KERNEL<32, 256, 1, 4>), g, dim3(NT), 0, stream, __VA_ARGS__); } break;  \
    }

static const int kTileBM[4] = {128, 64, 64, 32};
static const int kTileBN[4] = {128, 128, 64, 256};

extern "C" int rg_conv2d_fwd(const float* x, const float* w, float* y, int N, int C, int H, int W, int K, int KH, int KW,
                             int SH, int SW, int PH, int PW, int P, int Q, const float* scale, const float* shift,
                             const float* residual, int act, float slope, hipStream_t stream) {
    if (int e = validate("rg_conv2d_fwd", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(x && w && y, "rg_conv2d_fwd: null tensor");
    ConvP p;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = x; p.w = w; p.y = y;
    p.ep = Epilogue{scale, shift, residual, act, slope};
    p.M = K; p.Ng = N * P * Q; p.Kg = C * KH * KW;
    p.a_vec4 = (p.Kg % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
    const int tile = pick_tile(p.M, p.Ng);
    p.m_tiles = rg::cdiv(p.M, kTileBM[tile]);
    p.n_tiles = rg::cdiv(p.Ng, kTileBN[tile]);
    rg::ProfScope prof(rg::FAM_CONV_FWD, stream, 2.0 * p.M * (double)p.Ng * p.Kg);
    RG_TILE_DISPATCH(tile, conv_fwd_kernel, dim3(p.m_tiles * p.n_tiles), p);
    return rg::check_launch("rg_conv2d_fwd");
}

extern "C" int rg_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int K, int KH,
                               int KW, int SH, int SW, int PH, int PW, int P, int Q, const float* scale,
                               const float* shift, const float* residual, int act, float slope, hipStream_t stream) {
    if (int e = validate("rg_conv2d_dgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(dy && w && dx, "rg_conv2d_dgrad: null tensor");
    RG_REQUIRE(SH <= 2 && SW <= 2, "rg_conv2d_dgrad: stride > 2 not supported (got %d,%d)", SH, SW);
    DgradP dp;
    ConvP& p = dp.c;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = dy; p.w = w; p.y = dx;
    p.ep = Epilogue{scale, shift, residual, act, slope};
    p.M = C;
    int64_t ng_max = 0;
    double flops = 0.0;
    for (int ah = 0; ah < SH; ++ah)
        for (int aw = 0; aw < SW; ++aw) {
            DgradClass& cl = dp.cls[ah * SW + aw];
            cl.r0 = (ah + PH) % SH;
            cl.s0 = (aw + PW) % SW;
            cl.nrh = cl.r0 < KH ? (KH - cl.r0 + SH - 1) / SH : 0;
            cl.nrw = cl.s0 < KW ? (KW - cl.s0 + SW - 1) / SW : 0;
            cl.Hc = ah < H ? (H - ah + SH - 1) / SH : 0;
            cl.Wc = aw < W ? (W - aw + SW - 1) / SW : 0;
            cl.Ngc = N * cl.Hc * cl.Wc;
            cl.Kgc = K * cl.nrh * cl.nrw;
            cl.d_taps = make_fastdiv(cl.nrh * cl.nrw);
            cl.d_nrw = make_fastdiv(cl.nrw);
            cl.d_hw = make_fastdiv(cl.Hc * cl.Wc);
            cl.d_w = make_fastdiv(cl.Wc);
            if (cl.Ngc > ng_max) ng_max = cl.Ngc;
            flops += 2.0 * C * (double)cl.Ngc * cl.Kgc;
        }
    const int tile = pick_tile(p.M, ng_max * SH * SW);
    p.m_tiles = rg::cdiv(p.M, kTileBM[tile]);
    int nt_max = 0;
    for (int i = 0; i < SH * SW; ++i) {
        dp.cls[i].ntiles = rg::cdiv(dp.cls[i].Ngc, kTileBN[tile]);
        if (dp.cls[i].ntiles > nt_max) nt_max = dp.cls[i].ntiles;
    }
    p.n_tiles = nt_max;
    p.Ng = (int)ng_max;
    p.Kg = K * KH * KW;
    rg::ProfScope prof(rg::FAM_CONV_DGRAD, stream, flops);
    RG_TILE_DISPATCH(tile, conv_dgrad_kernel, dim3(p.m_tiles * nt_max, 1, SH * SW), dp);
    return rg::check_launch("rg_conv2d_dgrad");
}

namespace {
struct WgradPlan {
    int tile, m_tiles, n_tiles, splits, ktiles_per_split;
};
static WgradPlan plan_wgrad(int M, int Ng, int64_t Kg) {
    WgradPlan pl;
    pl.tile = (M <= 32) ? 3 : ((M <= 64 || Ng <= 64) ? 2 : 0);
    if (pl.tile == 0 && (int64_t)rg::cdiv(M, 128) * rg::cdiv(Ng, 128) < 64) pl.tile = 2;
    pl.m_tiles = rg::cdiv(M, kTileBM[pl.tile]);
    pl.n_tiles = rg::cdiv(Ng, kTileBN[pl.tile]);
    const int64_t nk = rg::cdiv64(Kg, BK);
    const int64_t mn = (int64_t)pl.m_tiles * pl.n_tiles;
    int64_t want = rg::cdiv64(1024, mn);  // aim at ~4 workgroups per CU
    if (want > nk / 4) want = nk / 4;     // keep >= 4 k-tiles per split
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;
    pl.ktiles_per_split = (int)rg::cdiv64(nk, want);
    pl.splits = (int)rg::cdiv64(nk, pl.ktiles_per_split);
    return pl;
}
}  // namespace

extern "C" size_t rg_conv2d_wgrad_workspace(int N, int C, int K, int KH, int KW, int P, int Q) {
    const WgradPlan pl = plan_wgrad(K, C * KH * KW, (int64_t)N * P * Q);
    if (pl.splits <= 1) return 0;
    return (size_t)pl.splits * (size_t)K * (size_t)C * KH * KW * sizeof(float);
}

extern "C" int rg_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH,
                               int KW, int SH, int SW, int PH, int PW, int P, int Q, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    if (int e = validate("rg_conv2d_wgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(x && dy && dw, "rg_conv2d_wgrad: null tensor");
    RG_REQUIRE(KH < 65536 && KW < 65536, "rg_conv2d_wgrad: filter too large");
    ConvP p;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = x; p.w = dy;
    p.ep = Epilogue{nullptr, nullptr, nullptr, 0, 0.f};
    p.M = K; p.Ng = C * KH * KW; p.Kg = N * P * Q;
    const WgradPlan pl = plan_wgrad(p.M, p.Ng, p.Kg);
    p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles;
    p.splits = pl.splits; p.ktiles_per_split = pl.ktiles_per_split;
    const size_t need = pl.splits > 1 ? (size_t)pl.splits * p.M * (size_t)p.Ng * sizeof(float) : 0;
    if (need > workspace_bytes || (need && !workspace)) {
        rg::set_error("rg_conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return RG_ERR_WORKSPACE;
    }
    p.y = pl.splits > 1 ? static_cast<float*>(workspace) : dw;
    const int tile = pl.tile;
    {
        rg::ProfScope prof(rg::FAM_CONV_WGRAD, stream, 2.0 * p.M * (double)p.Ng * p.Kg);
        RG_TILE_DISPATCH(tile, conv_wgrad_kernel, dim3(p.m_tiles * p.n_tiles, 1, pl.splits), p);
        if (int e = rg::check_launch("rg_conv2d_wgrad")) return e;
        if (pl.splits > 1) {
            const int64_t n = (int64_t)p.M * p.Ng;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rg::cdiv64(n, 256)), dim3(256), 0, stream,
                               static_cast<const float*>(workspace), dw, n, pl.splits);
        }
    }
    return rg::check_launch("rg_conv2d_wgrad(reduce)");
}
