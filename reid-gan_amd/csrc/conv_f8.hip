// FP8 implicit-GEMM convolution family for gfx950 (MI355X): forward, data gradient, weight gradient on
// v_mfma_scale_f32_32x32x64_f8f6f4 (OCP e4m3 activations / filters, e5m2 gradients, per-tensor scales, fp32
// accumulate).  BASELINE config 5: the dual_gan two-generator path (CC/dual_gan/models/DPTN_model.py:216-225) asks for
// "fp8 MFMA convs"; the reference itself computes these layers in fp32 through cuDNN (nn.Conv2d / nn.ConvTranspose2d in
// CC/dual_gan/models/base_function.py:236-443), so this family has a DECLARED tolerance against the fp32 oracle and an
// exact CPU emulation of its own arithmetic (oracle/ref_fp8.py) for parity.
//
// Design (not a dtype switch of conv_igemm.hip: fp8 MFMA wants 8 reduction-contiguous BYTES per lane, so the data layout,
// not the kernel, is what changes):
//   * operands are quantised ONCE by a transposing pass into layouts whose GEMM reduction axis is contiguous:
//       forward   y[k][pix]   = sum_{tap,c} W[k][tap][c]    * X[pix+tap][c]     X as [N][H*W][Cp]  ("NHWC"), W as [K][RS][Cp]
//       dgrad     dx[c][pix]  = sum_{tap,k} W^T[c][tap][k]  * DY[pix-tap][k]    DY as [N][P*Q][Kp],           W as [C][RS][Kp]
//       wgrad     dw[k][c,rs] = sum_{pix,n} DY[k][pix][n]   * X[c][pix+rs][n]   both as [C][H*W][Np] ("CHWN": the batch index is
//     the contiguous one, so a spatial filter shift never breaks the 16-byte alignment of a fragment);
//     channel / batch counts are padded to multiples of 16 with zeros, every staged access is one aligned 16-byte chunk whose
//     address comes from a per-chunk tap decode (raw buffer loads: chunks in the padding halo return 0);
//   * block tile BM x 128 x 64 bytes, 4 wave64 as 2 x 2, 32x32x64 MFMA tiles (one k-tile = one MFMA step per 32x32 block); LDS rows are 64 bytes with the 16-byte chunk
//     index XOR-swizzled by (row >> 2) & 3, which makes both the ds_write_b128 staging stores and the ds_read_b128 fragment
//     reads conflict-free; two ds_read_b128 per operand row feed one MFMA (lanes 0-31 take bytes 0-31 of the row, lanes 32-63
//     bytes 32-63, identically for both operands); double-buffered LDS, one barrier per k-tile, next tile's loads in flight during the MFMAs;
//   * per-tensor scaling state float[4] = {amax in use, amax being collected, dequantisation scale, format max}: the
//     quantiser clamps to the format range, collects the next amax with an integer atomicMax (order independent, deterministic),
//     and rg_f8_roll_scales() switches all states of a network in one launch (delayed scaling; rg_f8_amax + roll gives
//     just-in-time scaling for calibration and tests).
#include "rg_common.h"

#include <stdlib.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;
constexpr int BN = 128;
constexpr int ROWB = 64;                     // bytes of one LDS row = one k-tile
constexpr unsigned OOB = 0x80000000u;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ int4v bload16(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float4 bload4f(rsrc_t r, unsigned off) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float bloadf(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void bstoref(rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}

struct FastDiv {
    unsigned mul, shr, d;
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
    while ((1ull << l) < f.d) ++l;
    const unsigned p = 31 + l;
    f.mul = (unsigned)(((1ull << p) + f.d - 1) / f.d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    return f.d == 1 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shr);
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------------------------------------------
// quantisation
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ constexpr float f8_max(int fmt) { return fmt == 0 ? 448.f : 57344.f; }   // e4m3fn, e5m2 (OCP)

// state[1] = max(state[1], max |x|)
__global__ __launch_bounds__(256) void f8_amax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ state) {
    __shared__ float red[16];
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (; i + 3 < n; i += stride) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    for (; i < n; i += stride)
        for (int j = 0; j < 4 && i + j < n; ++j) m = fmaxf(m, fabsf(x[i + j]));
    m = rg_block_max(m, red);
    if (threadIdx.x == 0 && m > 0.f) atomicMax(reinterpret_cast<int*>(state + 1), __float_as_int(m));
}

// every state: {amax_cur, amax_next, dequant, fmax}: amax_cur <- amax_next (when something was collected), dequant = amax_cur / fmax
__global__ void f8_roll_kernel(float* __restrict__ states, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float* s = states + 4 * i;
    const float nxt = s[1];
    if (nxt > 0.f) s[0] = nxt;
    s[1] = 0.f;
    s[2] = s[0] > 0.f ? s[0] / s[3] : 1.f;
}

__device__ __forceinline__ unsigned pack4(float a, float b, float c, float d, int fmt) {
    int p = 0;
    if (fmt == 0) {
        p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    } else {
        p = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, p, false);
        p = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, p, true);
    }
    return (unsigned)p;
}

// out[b][l][r] (r < Rp, bytes) = fp8(clamp(in[b*bs + r*rs + l] * fmax / amax_cur)), zeros for r >= R; amax_next collected.
// grid (l tiles of 64, r tiles of 64, B); the tile goes through LDS so that reads run along l and writes along r.
__global__ __launch_bounds__(256) void f8_quantize_transpose_kernel(const float* __restrict__ in, unsigned char* __restrict__ out,
                                                                    float* __restrict__ state, float* __restrict__ scale_out,
                                                                    int fmt, int R, int L, int Rp, int64_t bs, int64_t rs) {
    __shared__ float tile[64][65];
    __shared__ float red[16];
    const int t = threadIdx.x;
    const int l0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int64_t base = (int64_t)blockIdx.z * bs;
    const float amax = state[0];
    const float fmax = f8_max(fmt);
    const float q = amax > 0.f ? fmax / amax : 1.f;
    if (scale_out && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) *scale_out = amax > 0.f ? amax / fmax : 1.f;
    float m = 0.f;
    {
        const int ll = t & 63, lr = t >> 6;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + lr + 4 * i;
            float v = 0.f;
            if (r < R && l0 + ll < L) v = in[base + (int64_t)r * rs + l0 + ll];
            m = fmaxf(m, fabsf(v));
            tile[lr + 4 * i][ll] = fminf(fmaxf(v * q, -fmax), fmax);
        }
    }
    __syncthreads();
    {
        const int ll = t >> 2, rq = t & 3;
        if (l0 + ll < L && r0 + rq * 16 < Rp) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = pack4(tile[rq * 16 + 4 * j][ll], tile[rq * 16 + 4 * j + 1][ll], tile[rq * 16 + 4 * j + 2][ll],
                             tile[rq * 16 + 4 * j + 3][ll], fmt);
            uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(out + ((int64_t)blockIdx.z * L + l0 + ll) * Rp + r0 + rq * 16) = v;
        }
    }
    m = rg_block_max(m, red);
    if (t == 0 && m > 0.f) atomicMax(reinterpret_cast<int*>(state + 1), __float_as_int(m));
}


// Both operand layouts of one tensor in ONE pass over the fp32 data: in[n][c][l] -> a[n][l][cp] and b[c][l][np] (cp / np: c / n
// padded to 16 with zero bytes).  Tile = 16 n x 16 c x 64 l, grid (l tiles, c tiles, n tiles).  Thread (lq, c) reads, in pass n,
// the float4 of pixels 4 lq .. 4 lq + 3 of row (n, c) (coalesced 256-byte runs) and packs it into one word of 4 fp8 bytes.
//   layout b (n contiguous): byte j of the thread's own 16 words IS the 16-byte run b[c][4 lq + j][n0 .. n0 + 15] — assembled with
//     byte permutes in registers, no LDS;
//   layout a (c contiguous): the words go through a 16 KiB LDS tile [n][lq][c] (word index XOR-swizzled: conflict-free stores),
//     thread (n, lq) reads its 16 words back with four ds_read_b128 and permutes them into the four runs a[n][4 lq + j][c0 .. c0 + 15].
// (The first version moved single bytes through LDS: 128 ds_read_u8 per thread; this one issues 16 ds_write_b32 + 4 ds_read_b128.)
__device__ __forceinline__ unsigned byte_of4(unsigned w0, unsigned w1, unsigned w2, unsigned w3, int j) {
    // {w0.byte[j], w1.byte[j], w2.byte[j], w3.byte[j]}; v_perm_b32 selects from the 8 bytes of (hi : lo), selector bytes 0-3 = lo
    const unsigned sel = (unsigned)j | ((unsigned)(4 + j) << 8);
    const unsigned t01 = __builtin_amdgcn_perm(w1, w0, sel);          // bytes 0, 1 = w0[j], w1[j]
    const unsigned t23 = __builtin_amdgcn_perm(w3, w2, sel);
    return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
}

// Gradient variant (yact != NULL and / or part != NULL): the tensor quantised is g = in * act'(yact) — the activation backward of the
// layer whose output gradient this is, never written out in fp32 — and part[tile][c] (tile = n tile * pixel tiles + pixel tile)
// receives the tile's sum of g per channel: the bias gradient is the column sum of `part` (rg_rows_sum_pair), fixed order.
__device__ __forceinline__ float f8_act_grad(float yv, int act, float slope) {      // `act` is uniform: selects, no branches
    const float neg = act == RG_ACT_LEAKY ? slope : 0.f;
    const float step = yv > 0.f ? 1.f : neg;
    return act == RG_ACT_TANH ? 1.f - yv * yv : (act == RG_ACT_NONE ? 1.f : step);
}

// Loads: every thread reads 16 float4 (one per sample of the tile) through a buffer resource that spans exactly the tile's samples —
// offsets outside it (samples beyond N, channels beyond C, pixels beyond L) return zeros, so the loads carry no branches and are
// issued eight samples at a time (the first version tested n / c / l per pass and waited for each load before the next: a
// 4-workgroup launch took 12 us).  VEC: L % 4 == 0 (one 16-byte load per sample), otherwise four dword loads.
template <bool VEC, bool HASY>
__global__ __launch_bounds__(256) void f8_quantize_dual_kernel(const float* __restrict__ in, unsigned char* __restrict__ a,
                                                               unsigned char* __restrict__ b, float* __restrict__ state,
                                                               float* __restrict__ scale_out, int fmt, int N, int C, int L,
                                                               int Cp, int Np, int xcd_groups, const float* __restrict__ yact,
                                                               int act, float slope, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) unsigned tile[16 * 16 * 16];        // word (n, lq, c) at ((n*16 + lq)*16 + (c ^ 4*(lq>>2)))
    __shared__ float red[16];
    const int t = threadIdx.x;
    // 1-D grid, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs, so id % 8 picks the XCD and all (c, n) tiles of one
    // pixel tile are consecutive workgroups of ONE XCD — the 16-byte pieces they write into the same 64 / 128-byte rows of `a`
    // (c tiles) and `b` (n tiles) meet in that XCD's L2 and leave it as full lines
    // (maps with fewer than 16 pixel tiles keep the plain order — pixel tile fastest — which spreads them over all XCDs)
    const int ct = Cp >> 4, nt = Np >> 4;
    const int per = ct * nt;
    int ltile, r;
    if (xcd_groups) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        ltile = (k / per) * 8 + xcd;
        r = k - (k / per) * per;
    } else {
        const int lt = (L + 63) >> 6;
        r = blockIdx.x / lt;
        ltile = blockIdx.x - r * lt;
    }
    const int l0 = ltile * 64, c0 = (r % ct) * 16, n0 = (r / ct) * 16;
    if (l0 >= L) return;                                 // whole workgroup: the pixel-tile count is padded to a multiple of 8
    const float amax = state[0];
    const float fmax = f8_max(fmt);
    const float q = amax > 0.f ? fmax / amax : 1.f;
    if (scale_out && t == 0 && blockIdx.x == 0) *scale_out = amax > 0.f ? amax / fmax : 1.f;
    float m = 0.f;
    const int lq = t & 15, cr = t >> 4;                  // this thread's pixel quad and channel of the tile
    const int c = c0 + cr, l = l0 + 4 * lq;
    const int nvalid = N - n0 < 16 ? N - n0 : 16;
    const unsigned sstride = (unsigned)C * (unsigned)L * 4u;              // bytes per sample; 16 of them < 2^31 (host check)
    const unsigned win = (unsigned)nvalid * sstride;
    const rsrc_t rin = make_rsrc(in + (int64_t)n0 * C * L, win);
    const rsrc_t ry = make_rsrc(HASY ? yact + (int64_t)n0 * C * L : in, HASY ? win : 0u);
    unsigned o0[4];                                                     // VEC: o0[0] only
#pragma unroll
    for (int j = 0; j < 4; ++j) o0[j] = (c < C && l + j < L) ? ((unsigned)c * (unsigned)L + (unsigned)(l + j)) * 4u : OOB;
    unsigned wn[16];                                     // word of sample n0 + pass
    float csum = 0.f;                                    // this thread's share of the channel sum (16 samples x 4 pixels)
    constexpr int HB = HASY ? 8 : 16;                    // samples whose loads are in flight together (64 VGPRs either way)
#pragma unroll
    for (int half = 0; half < 16 / HB; ++half) {
        float4 vv[HB], yy[HASY ? HB : 1];
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            const unsigned so = (unsigned)(half * HB + i) * sstride;
            if (VEC) {
                vv[i] = bload4f(rin, o0[0] + so);
                if (HASY) yy[i] = bload4f(ry, o0[0] + so);
            } else {
                vv[i] = make_float4(bloadf(rin, o0[0] + so), bloadf(rin, o0[1] + so), bloadf(rin, o0[2] + so), bloadf(rin, o0[3] + so));
                if (HASY) yy[i] = make_float4(bloadf(ry, o0[0] + so), bloadf(ry, o0[1] + so), bloadf(ry, o0[2] + so), bloadf(ry, o0[3] + so));
            }
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            const int pass = half * HB + i;
            float4 v = vv[i];
            if (HASY) {
                v.x *= f8_act_grad(yy[i].x, act, slope);
                v.y *= f8_act_grad(yy[i].y, act, slope);
                v.z *= f8_act_grad(yy[i].z, act, slope);
                v.w *= f8_act_grad(yy[i].w, act, slope);
            }
            csum += (v.x + v.y) + (v.z + v.w);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            wn[pass] = pack4(fminf(fmaxf(v.x * q, -fmax), fmax), fminf(fmaxf(v.y * q, -fmax), fmax),
                             fminf(fmaxf(v.z * q, -fmax), fmax), fminf(fmaxf(v.w * q, -fmax), fmax), fmt);
            if (a) tile[(pass * 16 + lq) * 16 + (cr ^ ((lq >> 2) << 2))] = wn[pass];
        }
    }
    if (part) {                                          // uniform; the 16 pixel-quad lanes of a channel are 16 consecutive lanes
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) csum += __shfl_xor(csum, o, 64);
        if (lq == 0 && c < C) part[((int64_t)(n0 >> 4) * ((L + 63) >> 6) + ltile) * C + c] = csum;
    }
    if (b && c < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (l + j >= L) break;
            const uint4 o = make_uint4(byte_of4(wn[0], wn[1], wn[2], wn[3], j), byte_of4(wn[4], wn[5], wn[6], wn[7], j),
                                       byte_of4(wn[8], wn[9], wn[10], wn[11], j), byte_of4(wn[12], wn[13], wn[14], wn[15], j));
            *reinterpret_cast<uint4*>(b + ((int64_t)c * L + l + j) * Np + n0) = o;
        }
    }
    if (a) {
        __syncthreads();
        const int nr = t >> 4;                           // (sample nr, pixel quad lq) of the tile
        if (n0 + nr < N) {
            uint4 w4[4];                                 // w4[k] = words of channels 4k .. 4k + 3
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w4[k] = *reinterpret_cast<const uint4*>(&tile[(nr * 16 + lq) * 16 + ((k ^ (lq >> 2)) << 2)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (l + j >= L) break;
                const uint4 o = make_uint4(byte_of4(w4[0].x, w4[0].y, w4[0].z, w4[0].w, j), byte_of4(w4[1].x, w4[1].y, w4[1].z, w4[1].w, j),
                                           byte_of4(w4[2].x, w4[2].y, w4[2].z, w4[2].w, j), byte_of4(w4[3].x, w4[3].y, w4[3].z, w4[3].w, j));
                *reinterpret_cast<uint4*>(a + ((int64_t)(n0 + nr) * L + l + j) * Cp + c0) = o;
            }
        }
    }
    m = rg_block_max(m, red);
    if (t == 0 && m > 0.f) atomicMax(reinterpret_cast<int*>(state + 1), __float_as_int(m));
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM core
// ---------------------------------------------------------------------------------------------------------------
struct F8Class {          // one stride-parity class of the data gradient (the single class of a stride-1 layer)
    int r0, s0, nrh, nrw, Hc, Wc, Ngc, Kc, ntiles;
    FastDiv d_nrw, d_hw, d_w;
};

struct F8P {
    const unsigned char* A;   // row operand of the GEMM rows m
    const unsigned char* B;   // row operand of the GEMM columns n
    unsigned a_bytes, b_bytes;
    float* out;               // fp32 result (NCHW activation, or dw / split-K partials for wgrad)
    unsigned out_bytes;
    const float* sa;          // dequantisation scales of the two operands (one device float each, written by the quantiser)
    const float* sb;
    const float* shift;       // per GEMM row (output channel) or null
    const float* res;         // like out or null
    int act;
    float slope;
    int M, Ng, Kc;            // rows, columns, 16-byte chunks along the reduction (fwd / wgrad; dgrad: per class)
    int m_tiles, n_tiles;
    int N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q;
    int Cq;                   // 16-byte chunks per pixel of the gathered operand (fwd: Cp/16, dgrad: Kp/16, wgrad: Np/16)
    FastDiv d_cq, d_kw, d_pq, d_q, d_rs;
    int ktiles_per_split, splits;      // wgrad
    F8Class cls[4];
};

typedef int int8v __attribute__((ext_vector_type(8)));

// One 32x32x64 step: each lane supplies 32 reduction-contiguous bytes of its row of A and of B (lanes 0-31: bytes 0-31 of the
// 64-byte k-tile, lanes 32-63: bytes 32-63).  v_mfma_scale_f32_32x32x64_f8f6f4 with both block scales at 2^0 (E8M0 127) is
// the plain fp8 product at twice the issue rate of the 32x32x16 forms (the per-tensor scales are applied in the epilogue).
// FA / FB: 0 = e4m3, 1 = e5m2 — the instruction's own format codes.
template <int FA, int FB>
__device__ __forceinline__ floatx16 mfma8(int8v a, int8v b, floatx16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FA, FB, 0, 127, 0, 127);
}

__device__ __forceinline__ unsigned lds_off(int row, int chunk) { return (unsigned)row * ROWB + (unsigned)((chunk ^ (row >> 2)) & 3) * 16u; }

// MODE 0 forward, 1 data gradient (class from the workgroup id, see below), 2 weight gradient (blockIdx.z = split)
// NB = 2: operands staged through registers (global -> VGPR -> ds_write), two LDS buffers, prefetch distance one k-tile.
// NB > 2: LDS-DMA ring (buffer_load ... lds, 16 bytes per lane): no staging registers, NB - 1 k-tiles in flight per workgroup.  A
//         k-tile is only 64 reduction bytes = one MFMA step per 32x32 block (~100 ns of matrix work per workgroup), so the loop is
//         bound by load LATENCY, not by issue: the ring plus 2-3 resident workgroups keeps ~10 tiles per CU in flight.
typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE, int BM, int FA, int FB, int NB>
__global__ __launch_bounds__(NT) void conv_f8_kernel(const F8P p) {
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int AR = BM / 64;              // A rows staged per thread (64 rows per pass)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NB][(BM + BN) * ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // Data gradient with stride: the SH*SW parity classes of one tile write INTERLEAVED pixels of the same output rows (every
    // second float).  Their workgroups are laid out back to back on one XCD (id % 8 = XCD, classes fastest), so those half-written
    // lines meet in one L2 and leave it whole; launched class by class (grid z) the same bytes cost twice the HBM write time.
    int cls_idx = 0, tile_pre = -1;
    if (MODE == 1) {
        const int ncls = p.SH * p.SW;
        if (ncls > 1) {
            const int slots = p.m_tiles * p.n_tiles;           // n_tiles = the largest class
            const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
            cls_idx = k % ncls;
            const int sl = k / ncls;
            const int q = slots >> 3, r = slots & 7;
            if (sl >= q + (xcd < r ? 1 : 0)) return;
            tile_pre = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + sl;
        }
    }
    const F8Class& cl = p.cls[cls_idx];
    const int ntl = MODE == 1 ? cl.ntiles : p.n_tiles;
    const int nwg = p.m_tiles * ntl;
    if (tile_pre < 0 && (int)blockIdx.x >= nwg) return;
    const int tile = tile_pre >= 0 ? tile_pre : xcd_remap(blockIdx.x, nwg);
    if (tile >= nwg) return;
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int Ng = MODE == 1 ? cl.Ngc : p.Ng;
    const int Kc = MODE == 1 ? cl.Kc : p.Kc;
    const rsrc_t ra = make_rsrc(p.A, p.a_bytes), rb = make_rsrc(p.B, p.b_bytes);
    const int RS = p.KH * p.KW, HW = p.H * p.W, PQ = p.P * p.Q;
    const int ah = MODE == 1 ? cls_idx / p.SW : 0, aw = MODE == 1 ? cls_idx % p.SW : 0;

    // ---- per-thread staging rows: chunk (tid & 3) of rows (tid >> 2) + 64 i ----
    const int ch = tid & 3, srow = tid >> 2;
    int arow[AR];
    bool aok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        arow[i] = m0 + srow + 64 * i;
        aok[i] = arow[i] < p.M;
    }
    // gathered operand B: decode the two rows this thread stages
    int b_img[2], b_y[2], b_x[2];
    bool bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + srow + 64 * i;
        bok[i] = n < Ng;
        b_img[i] = b_y[i] = b_x[i] = 0;
        if (bok[i]) {
            if (MODE == 0) {                       // column = output pixel (img, pp, qq): top-left input position
                const int img = fdiv(n, p.d_pq);
                const int pq = n - img * PQ;
                const int pp = fdiv(pq, p.d_q);
                b_img[i] = img;
                b_y[i] = pp * p.SH - p.PH;
                b_x[i] = (pq - pp * p.Q) * p.SW - p.PW;
            } else if (MODE == 1) {                // column = input pixel (img, hc, wc) of this class
                const int img = fdiv(n, cl.d_hw);
                const int rem = n - img * cl.Hc * cl.Wc;
                const int hc = fdiv(rem, cl.d_w);
                const int wc = rem - hc * cl.Wc;
                b_img[i] = img;
                b_y[i] = (ah + p.SH * hc + p.PH - cl.r0) / p.SH;
                b_x[i] = (aw + p.SW * wc + p.PW - cl.s0) / p.SW;
            } else {                               // column = (c, r, s) of the filter
                const int c = fdiv(n, p.d_rs);
                const int rs = n - c * RS;
                const int r = fdiv(rs, p.d_kw);
                b_img[i] = c;
                b_y[i] = r - p.PH;
                b_x[i] = rs - r * p.KW - p.PW;
            }
        }
    }

    int4v sa_[AR], sb_[2];
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // byte offsets (or OOB) of the 16-byte chunks this thread stages for k-tile kt; `chx` = its chunk of the 64-byte row
    auto tile_offsets = [&](int kt, int chx, unsigned (&oa)[AR], unsigned (&ob)[2]) {
        const int q = kt * 4 + chx;                // chunk index along the reduction
        const bool qok = q < Kc;
        const int t = fdiv(q, p.d_cq);             // fwd: filter tap; dgrad: tap of the class; wgrad: output pixel
        const int cc = q - t * p.Cq;
        if (MODE == 0) {
            const int r = fdiv(t, p.d_kw), s = t - r * p.KW;
#pragma unroll
            for (int i = 0; i < AR; ++i) oa[i] = (aok[i] && qok) ? (unsigned)(arow[i] * Kc + q) * 16u : OOB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int h = b_y[i] + r, w = b_x[i] + s;
                const bool ok = bok[i] && qok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                ob[i] = ok ? (unsigned)(((b_img[i] * p.H + h) * p.W + w) * p.Cq + cc) * 16u : OOB;
            }
        } else if (MODE == 1) {
            const int j = fdiv(t, cl.d_nrw), jj = t - j * cl.nrw;
            const int rs = (cl.r0 + p.SH * j) * p.KW + cl.s0 + p.SW * jj;
#pragma unroll
            for (int i = 0; i < AR; ++i) oa[i] = (aok[i] && qok) ? (unsigned)((arow[i] * RS + rs) * p.Cq + cc) * 16u : OOB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pp = b_y[i] - j, qq = b_x[i] - jj;
                const bool ok = bok[i] && qok && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
                ob[i] = ok ? (unsigned)(((b_img[i] * p.P + pp) * p.Q + qq) * p.Cq + cc) * 16u : OOB;
            }
        } else {
            const int pp = fdiv(t, p.d_q), qq = t - pp * p.Q;
#pragma unroll
            for (int i = 0; i < AR; ++i) oa[i] = (aok[i] && qok) ? (unsigned)((arow[i] * PQ + t) * p.Cq + cc) * 16u : OOB;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int h = pp * p.SH + b_y[i], w = qq * p.SW + b_x[i];
                const bool ok = bok[i] && qok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                ob[i] = ok ? (unsigned)((b_img[i] * HW + h * p.W + w) * p.Cq + cc) * 16u : OOB;
            }
        }
    };
    auto load_tile = [&](int kt) {
        unsigned oa[AR], ob[2];
        tile_offsets(kt, ch, oa, ob);
#pragma unroll
        for (int i = 0; i < AR; ++i) sa_[i] = bload16(ra, oa[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) sb_[i] = bload16(rb, ob[i]);
    };
    // LDS-DMA: lane l of wave w lands at (w*64 + l) * 16 bytes of the 4 KiB pass = row w*16 + l/4, physical chunk l%4, so the lane
    // fetches the LOGICAL chunk whose swizzled position that is (the XOR is an involution); out-of-range chunks arrive as zeros
    const int che = ch ^ ((srow >> 2) & 3);
    const unsigned lds_lane0 = (unsigned)__builtin_amdgcn_readfirstlane(wid) * 1024u;
    auto dma_tile = [&](int kt, int buf) {
        unsigned oa[AR], ob[2];
        tile_offsets(kt, che, oa, ob);
        unsigned char* base = lds[buf] + lds_lane0;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(base + i * 4096), 16, (int)oa[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(base + BM * ROWB + i * 4096), 16, (int)ob[i], 0, 0, 0);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<int4v*>(&lds[buf][lds_off(srow + 64 * i, ch)]) = sa_[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<int4v*>(&lds[buf][BM * ROWB + lds_off(srow + 64 * i, ch)]) = sb_[i];
    };

    const int nk = (Kc + 3) >> 2;
    int kt_begin = 0, kt_end = nk;
    if (MODE == 2) {
        kt_begin = (int)blockIdx.z * p.ktiles_per_split;
        kt_end = min(kt_begin + p.ktiles_per_split, nk);
    }
    const int l32 = lane & 31, lh = lane >> 5;
    auto compute_tile = [&](const unsigned char* As, const unsigned char* Bs) {
        int8v a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 32 + l32;
                const int4v lo = *reinterpret_cast<const int4v*>(As + lds_off(row, 2 * lh));
                const int4v hi = *reinterpret_cast<const int4v*>(As + lds_off(row, 2 * lh + 1));
                a[i] = int8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * WTN + j * 32 + l32;
                const int4v lo = *reinterpret_cast<const int4v*>(Bs + lds_off(row, 2 * lh));
                const int4v hi = *reinterpret_cast<const int4v*>(Bs + lds_off(row, 2 * lh + 1));
                b[j] = int8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma8<FA, FB>(a[i], b[j], acc[i][j]);
    };
    if (NB == 2) {
        if (kt_begin < kt_end) {
            load_tile(kt_begin);
            store_tile(0);
        }
        __syncthreads();
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool has_next = kt + 1 < kt_end;
            if (has_next) load_tile(kt + 1);
            compute_tile(lds[cur], lds[cur] + BM * ROWB);
            if (has_next) store_tile(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    } else {
        constexpr int LPT = AR + 2;                // DMA instructions per thread and k-tile
        const int nkt = kt_end - kt_begin;
#pragma unroll
        for (int sidx = 0; sidx < NB - 1; ++sidx)
            if (sidx < nkt) dma_tile(kt_begin + sidx, sidx);
        int buf = 0;
        for (int it = 0; it < nkt; ++it) {
            // tile `it` has landed once at most the younger tiles' DMAs (issued after it) are outstanding
            const int younger = min(NB - 2, nkt - 1 - it);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // every wave's part of tile `it` is in LDS; every wave is done reading tile it-1
            if (it + NB - 1 < nkt) {
                int nbuf = buf + NB - 1;
                if (nbuf >= NB) nbuf -= NB;
                dma_tile(kt_begin + it + NB - 1, nbuf);          // into the buffer tile it-1 used
            }
            compute_tile(lds[buf], lds[buf] + BM * ROWB);
            if (++buf == NB) buf = 0;
        }
    }

    // ---- epilogue ----
    const rsrc_t ro = make_rsrc(p.out, p.out_bytes);
    const int mrow0 = m0 + wm * WTM + 4 * lh;
    if (MODE == 2) {                               // raw partial sums [split][M][Ng]; the reduce kernel applies the scales
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nn = n0 + wn * WTN + j * 32 + l32;
            const unsigned ob = nn < Ng ? (unsigned)((((int64_t)blockIdx.z * p.M + mrow0) * Ng + nn) * 4) : OOB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                    bstoref(ro, (mrow0 + mo < p.M) ? ob + (unsigned)(mo * Ng) * 4u : OOB, acc[i][j][r]);
                }
        }
        return;
    }
    const float scale = p.sa[0] * p.sb[0];
    const bool has_shift = p.shift != nullptr, has_res = p.res != nullptr;       // uniform
    const rsrc_t rsh = make_rsrc(has_shift ? p.shift : p.sa, has_shift ? (unsigned)p.M * 4u : 0u);
    const rsrc_t rr = make_rsrc(has_res ? (const void*)p.res : (const void*)p.out, has_res ? p.out_bytes : 0u);
    const int PIX = MODE == 0 ? PQ : HW;
    // loads first, in batches (one wait per batch), then arithmetic and stores: a load -> wait -> store chain per element made the
    // first version of this epilogue cost more than the whole k-loop on the small DPTN layers
    float sh[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[i][r] = 0.f;
    if (has_shift) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sh[i][r] = bloadf(rsh, (unsigned)(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2)) * 4u);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = n0 + wn * WTN + j * 32 + l32;
        unsigned ob = OOB;
        if (nn < Ng) {
            if (MODE == 0) {
                const int img = fdiv(nn, p.d_pq);
                ob = (unsigned)((((int64_t)img * p.M + mrow0) * PQ + (nn - img * PQ)) * 4);
            } else {
                const int img = fdiv(nn, cl.d_hw);
                const int rem = nn - img * cl.Hc * cl.Wc;
                const int hc = fdiv(rem, cl.d_w);
                const int wc = rem - hc * cl.Wc;
                ob = (unsigned)((((int64_t)img * p.M + mrow0) * HW + (ah + p.SH * hc) * p.W + aw + p.SW * wc) * 4);
            }
        }
        unsigned off[TM][16];
        float rv[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                off[i][r] = (mrow0 + mo < p.M) ? ob + (unsigned)(mo * PIX) * 4u : OOB;
                rv[i][r] = 0.f;
            }
        if (has_res) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[i][r] = bloadf(rr, off[i][r]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bstoref(ro, off[i][r], rg_apply_act(acc[i][j][r] * scale + sh[i][r] + rv[i][r], p.act, p.slope));
    }
}

// dw[i] = scale_a * scale_b * sum_s partial[s][i]   (fixed summation tree: deterministic)
__global__ __launch_bounds__(256) void f8_splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n,
                                                               int splits, const float* __restrict__ sa, const float* __restrict__ sb) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int k = ty;
        for (; k + 4 < splits; k += 8) {
            s0 += ws[(int64_t)k * n + i];
            s1 += ws[(int64_t)(k + 4) * n + i];
        }
        if (k < splits) s0 += ws[(int64_t)k * n + i];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty != 0 || i >= n) return;
    out[i] = ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) * (sa[0] * sb[0]);
}

// n % 4 == 0: float4 outputs, four slab loads in flight; per element the summation tree of f8_splitk_reduce_kernel
__global__ __launch_bounds__(256) void f8_splitk_reduce_vec_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n,
                                                                   int splits, const float* __restrict__ sa,
                                                                   const float* __restrict__ sb) {
    __shared__ float4 red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * 64 + tx;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (i < n4) {
        int k = ty;
        for (; k + 12 < splits; k += 16) {
            const float4 a = w4[(int64_t)k * n4 + i], b = w4[(int64_t)(k + 4) * n4 + i];
            const float4 c = w4[(int64_t)(k + 8) * n4 + i], d = w4[(int64_t)(k + 12) * n4 + i];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s0.x += c.x; s0.y += c.y; s0.z += c.z; s0.w += c.w;
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        }
        for (; k + 4 < splits; k += 8) {
            const float4 a = w4[(int64_t)k * n4 + i], b = w4[(int64_t)(k + 4) * n4 + i];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (k < splits) {
            const float4 a = w4[(int64_t)k * n4 + i];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[ty][tx] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (ty != 0 || i >= n4) return;
    const float4 r0 = red[0][tx], r1 = red[1][tx], r2 = red[2][tx], r3 = red[3][tx];
    const float sc = sa[0] * sb[0];
    reinterpret_cast<float4*>(out)[i] = make_float4(((r0.x + r1.x) + (r2.x + r3.x)) * sc, ((r0.y + r1.y) + (r2.y + r3.y)) * sc,
                                                    ((r0.z + r1.z) + (r2.z + r3.z)) * sc, ((r0.w + r1.w) + (r2.w + r3.w)) * sc);
}

static bool fits(int64_t bytes) { return bytes > 0 && bytes < (1ll << 31); }

static void fill_geom(F8P& p, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.KH = KH; p.KW = KW;
    p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW; p.P = P; p.Q = Q;
    p.d_kw = make_fastdiv(KW);
    p.d_pq = make_fastdiv(P * Q);
    p.d_q = make_fastdiv(Q);
    p.d_rs = make_fastdiv(KH * KW);
    p.ktiles_per_split = 1 << 30;
    p.splits = 1;
    p.shift = nullptr;
    p.res = nullptr;
    p.act = 0;
    p.slope = 0.f;
}

static int check_geom(const char* op, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P,
                      int Q) {
    RG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 && PW >= 0 && P > 0 &&
                   Q > 0, "%s: bad dimension", op);
    RG_REQUIRE((int64_t)(P - 1) * SH - PH < H && (int64_t)(Q - 1) * SW - PW < W, "%s: output larger than input allows", op);
    RG_REQUIRE(fits((int64_t)N * C * H * W * 4) && fits((int64_t)N * K * P * Q * 4) && fits((int64_t)K * C * KH * KW * 4),
               "%s: every tensor must be smaller than 2 GiB (32-bit buffer offsets)", op);
    return RG_OK;
}

static int pad16(int v) { return (v + 15) / 16 * 16; }

#define RG_F8_LAUNCH_NB(MODE_, GRID_, NB_)                                                                                  \
    do {                                                                                                                   \
        if (bm == 128) {                                                                                                   \
            if (fa == 0 && fb == 0) hipLaunchKernelGGL((conv_f8_kernel<MODE_, 128, 0, 0, NB_>), GRID_, dim3(NT), 0, stream, p);  \
            else if (fa == 0) hipLaunchKernelGGL((conv_f8_kernel<MODE_, 128, 0, 1, NB_>), GRID_, dim3(NT), 0, stream, p);        \
            else hipLaunchKernelGGL((conv_f8_kernel<MODE_, 128, 1, 0, NB_>), GRID_, dim3(NT), 0, stream, p);                     \
        } else {                                                                                                           \
            if (fa == 0 && fb == 0) hipLaunchKernelGGL((conv_f8_kernel<MODE_, 64, 0, 0, NB_>), GRID_, dim3(NT), 0, stream, p);   \
            else if (fa == 0) hipLaunchKernelGGL((conv_f8_kernel<MODE_, 64, 0, 1, NB_>), GRID_, dim3(NT), 0, stream, p);         \
            else hipLaunchKernelGGL((conv_f8_kernel<MODE_, 64, 1, 0, NB_>), GRID_, dim3(NT), 0, stream, p);                      \
        }                                                                                                                  \
    } while (0)

// RG_F8_BM=64|128 pins the tile height of the forward / data-gradient GEMMs (A/B measurements); anything else is ignored
static int f8_bm_override() {
    static const int v = getenv("RG_F8_BM") ? atoi(getenv("RG_F8_BM")) : 0;
    return (v == 64 || v == 128) ? v : 0;
}

// RG_F8_NB=2 selects the register-staged two-buffer loop (the first version; kept for A/B measurements), default the 4-deep DMA ring
static int f8_ring() {
    static const int nb = getenv("RG_F8_NB") ? atoi(getenv("RG_F8_NB")) : 4;
    return nb == 2 ? 2 : 4;
}
#define RG_F8_LAUNCH(MODE_, GRID_)                      \
    do {                                                \
        if (f8_ring() == 2) RG_F8_LAUNCH_NB(MODE_, GRID_, 2); \
        else RG_F8_LAUNCH_NB(MODE_, GRID_, 4);          \
    } while (0)

}  // namespace

// ---- scaling states -----------------------------------------------------------------------------------------------
extern "C" int rg_f8_amax(const float* x, int64_t n, float* state, hipStream_t stream) {
    RG_REQUIRE(x && state && n > 0, "rg_f8_amax: bad arguments");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 4.0 * n);
    int64_t g = rg::cdiv64(n, 256 * 4 * 4);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(f8_amax_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, stream, x, n, state);
    return rg::check_launch("rg_f8_amax");
}

extern "C" int rg_f8_roll_scales(float* states, int count, hipStream_t stream) {
    RG_REQUIRE(states && count > 0, "rg_f8_roll_scales: bad arguments");
    hipLaunchKernelGGL(f8_roll_kernel, dim3(rg::cdiv(count, 256)), dim3(256), 0, stream, states, count);
    return rg::check_launch("rg_f8_roll_scales");
}

// out[b][l][r] (bytes, r padded to Rp = 16 * ceil(R / 16) with zeros) = fp8(in[b*bs + r*rs + l] * fmax / state[0]); fmt 0 e4m3, 1 e5m2
extern "C" int rg_f8_quantize(const float* in, void* out, float* state, float* scale_out, int fmt, int B, int R, int L,
                              int64_t bs, int64_t rs, hipStream_t stream) {
    RG_REQUIRE(in && out && state && (fmt == 0 || fmt == 1) && B > 0 && R > 0 && L > 0, "rg_f8_quantize: bad arguments");
    RG_REQUIRE(B <= 65535, "rg_f8_quantize: batch dimension %d exceeds the grid limit", B);
    const int Rp = pad16(R);
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, (double)B * L * (4.0 * R + Rp));
    hipLaunchKernelGGL(f8_quantize_transpose_kernel, dim3(rg::cdiv(L, 64), rg::cdiv(Rp, 64), B), dim3(256), 0, stream, in,
                       static_cast<unsigned char*>(out), state, scale_out, fmt, R, L, Rp, bs, rs);
    return rg::check_launch("rg_f8_quantize");
}

static void launch_quantize_dual(hipStream_t stream, unsigned wgs, const float* in, void* a, void* b, float* state, float* scale_out,
                                 int fmt, int N, int C, int L, int Cp, int Np, int xcd_groups, const float* yact, int act, float slope,
                                 float* part) {
    const bool vec = (L & 3) == 0;
#define RG_QD(V, Y)                                                                                                              \
    hipLaunchKernelGGL((f8_quantize_dual_kernel<V, Y>), dim3(wgs), dim3(256), 0, stream, in, static_cast<unsigned char*>(a),      \
                       static_cast<unsigned char*>(b), state, scale_out, fmt, N, C, L, Cp, Np, xcd_groups, yact, act, slope, part)
    if (vec && yact) RG_QD(true, true);
    else if (vec) RG_QD(true, false);
    else if (yact) RG_QD(false, true);
    else RG_QD(false, false);
#undef RG_QD
}

// Both layouts of in[N][C][L] in one pass: a [N][L][Cp] (may be NULL) and b [C][L][Np] (may be NULL); see rg_f8_quantize.
// Padding rows of `a` beyond C and of `b` beyond N are written as zeros only inside the 16-wide tiles that hold real data; callers
// allocate exactly Cp = 16*ceil(C/16), Np = 16*ceil(N/16), which those tiles cover.
extern "C" int rg_f8_quantize_dual(const float* in, void* a, void* b, float* state, float* scale_out, int fmt, int N, int C, int L,
                                   hipStream_t stream) {
    RG_REQUIRE(in && (a || b) && state && (fmt == 0 || fmt == 1) && N > 0 && C > 0 && L > 0, "rg_f8_quantize_dual: bad arguments");
    const int Cp = pad16(C), Np = pad16(N);
    const int lt = rg::cdiv(L, 64);
    const int xcd_groups = lt >= 16 ? 1 : 0;
    const int64_t wgs = (int64_t)(xcd_groups ? rg::cdiv(lt, 8) * 8 : lt) * (Cp / 16) * (Np / 16);
    RG_REQUIRE(wgs < (1ll << 31), "rg_f8_quantize_dual: tensor exceeds the grid limit");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, (double)L * (4.0 * N * C + (a ? (double)N * Cp : 0.0) + (b ? (double)C * Np : 0.0)));
    RG_REQUIRE((int64_t)C * L < (1ll << 24), "rg_f8_quantize_dual: C * L exceeds the 2^31-byte window of one 16-sample tile");
    launch_quantize_dual(stream, (unsigned)wgs, in, a, b, state, scale_out, fmt, N, C, L, Cp, Np, xcd_groups, nullptr, RG_ACT_NONE, 0.f,
                         nullptr);
    return rg::check_launch("rg_f8_quantize_dual");
}

// rows of `part` rg_f8_quantize_grad writes for [N][C][L]: one per (16-sample, 64-pixel) tile
extern "C" int rg_f8_grad_tiles(int N, int L) { return (pad16(N) / 16) * rg::cdiv(L, 64); }

// The output-gradient operand of a convolution's backward in one pass over dy: g = dy * act'(yact) (yact NULL: g = dy) quantised
// into a [N][L][Cp] and / or b [C][L][Np] as rg_f8_quantize_dual does, with the per-tile channel sums of g in
// part[rg_f8_grad_tiles(N, L)][C] (NULL: not wanted).  Neither g nor a separate reduction pass over it touches memory.
extern "C" int rg_f8_quantize_grad(const float* dy, const float* yact, int act, float slope, void* a, void* b, float* part,
                                   float* state, float* scale_out, int fmt, int N, int C, int L, hipStream_t stream) {
    RG_REQUIRE(dy && (a || b || part) && state && (fmt == 0 || fmt == 1) && N > 0 && C > 0 && L > 0, "rg_f8_quantize_grad: bad arguments");
    RG_REQUIRE(act == RG_ACT_NONE || yact, "rg_f8_quantize_grad: the activation backward needs the forward output");
    const int Cp = pad16(C), Np = pad16(N);
    const int lt = rg::cdiv(L, 64);
    const int xcd_groups = lt >= 16 ? 1 : 0;
    const int64_t wgs = (int64_t)(xcd_groups ? rg::cdiv(lt, 8) * 8 : lt) * (Cp / 16) * (Np / 16);
    RG_REQUIRE(wgs < (1ll << 31), "rg_f8_quantize_grad: tensor exceeds the grid limit");
    const bool has_act = act != RG_ACT_NONE;
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0,
                       (double)L * ((has_act ? 8.0 : 4.0) * N * C + (a ? (double)N * Cp : 0.0) + (b ? (double)C * Np : 0.0)));
    RG_REQUIRE((int64_t)C * L < (1ll << 24), "rg_f8_quantize_grad: C * L exceeds the 2^31-byte window of one 16-sample tile");
    launch_quantize_dual(stream, (unsigned)wgs, dy, a, b, state, scale_out, fmt, N, C, L, Cp, Np, xcd_groups, has_act ? yact : nullptr,
                         act, slope, part);
    return rg::check_launch("rg_f8_quantize_grad");
}

// ---- forward: y[N][K][P][Q] fp32 = act(sx * sw * conv(xq, wq) + shift + residual) ------------------------------------
// xq [N][H*W][Cp] (fmt_x: 0 e4m3 activations, 1 e5m2 — the data gradient of a transposed convolution), wq [K][KH*KW][Cp] e4m3
extern "C" int rg_conv2d_f8_fwd(const void* xq, const void* wq, const float* sx, const float* sw, int fmt_x, float* y, int N,
                                int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q,
                                const float* shift, const float* residual, int act, float slope, hipStream_t stream) {
    if (int e = check_geom("rg_conv2d_f8_fwd", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(xq && wq && sx && sw && y, "rg_conv2d_f8_fwd: null tensor");
    F8P p;
    fill_geom(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    const int Cp = pad16(C);
    p.A = static_cast<const unsigned char*>(wq);
    p.B = static_cast<const unsigned char*>(xq);
    p.a_bytes = (unsigned)((int64_t)K * KH * KW * Cp);
    p.b_bytes = (unsigned)((int64_t)N * H * W * Cp);
    p.out = y;
    p.out_bytes = (unsigned)((int64_t)N * K * P * Q * 4);
    p.sa = sw; p.sb = sx;
    p.shift = shift; p.res = residual; p.act = act; p.slope = slope;
    p.M = K; p.Ng = N * P * Q;
    p.Cq = Cp / 16;
    p.Kc = KH * KW * p.Cq;
    p.d_cq = make_fastdiv(p.Cq);
    static const int bm_env = f8_bm_override();
    const int bm = bm_env ? bm_env : (K <= 64 ? 64 : 128), fa = 0, fb = fmt_x;
    p.m_tiles = rg::cdiv(K, bm);
    p.n_tiles = rg::cdiv(p.Ng, BN);
    rg::ProfScope prof(rg::FAM_CONV_F8, stream, 2.0 * K * (double)p.Ng * C * KH * KW,
                       (double)N * H * W * Cp + (double)K * KH * KW * Cp + 4.0 * N * K * P * Q);
    const dim3 grid(p.m_tiles * p.n_tiles, 1, 1);
    RG_F8_LAUNCH(0, grid);
    return rg::check_launch("rg_conv2d_f8_fwd");
}

// ---- data gradient / transposed-convolution forward: dx[N][C][H][W] fp32 from dyq [N][P*Q][Kp] and wq_t [C][KH*KW][Kp] --------
extern "C" int rg_conv2d_f8_dgrad(const void* dyq, const void* wq_t, const float* sdy, const float* sw, int fmt_dy, float* dx,
                                  int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P,
                                  int Q, const float* shift, const float* residual, int act, float slope, hipStream_t stream) {
    if (int e = check_geom("rg_conv2d_f8_dgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(dyq && wq_t && sdy && sw && dx, "rg_conv2d_f8_dgrad: null tensor");
    RG_REQUIRE(SH <= 2 && SW <= 2, "rg_conv2d_f8_dgrad: stride > 2 not supported (got %d,%d)", SH, SW);
    F8P p;
    fill_geom(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    const int Kp = pad16(K);
    p.A = static_cast<const unsigned char*>(wq_t);
    p.B = static_cast<const unsigned char*>(dyq);
    p.a_bytes = (unsigned)((int64_t)C * KH * KW * Kp);
    p.b_bytes = (unsigned)((int64_t)N * P * Q * Kp);
    p.out = dx;
    p.out_bytes = (unsigned)((int64_t)N * C * H * W * 4);
    p.sa = sw; p.sb = sdy;
    p.shift = shift; p.res = residual; p.act = act; p.slope = slope;
    p.M = C;
    p.Cq = Kp / 16;
    p.d_cq = make_fastdiv(p.Cq);
    static const int bm_env = f8_bm_override();
    const int bm = bm_env ? bm_env : (C <= 64 ? 64 : 128), fa = 0, fb = fmt_dy;
    p.m_tiles = rg::cdiv(C, bm);
    int nt_max = 0;
    double flops = 0.0;
    for (int ah = 0; ah < SH; ++ah)
        for (int aw = 0; aw < SW; ++aw) {
            F8Class& cl = p.cls[ah * SW + aw];
            cl.r0 = (ah + PH) % SH;
            cl.s0 = (aw + PW) % SW;
            cl.nrh = cl.r0 < KH ? (KH - cl.r0 + SH - 1) / SH : 0;
            cl.nrw = cl.s0 < KW ? (KW - cl.s0 + SW - 1) / SW : 0;
            cl.Hc = ah < H ? (H - ah + SH - 1) / SH : 0;
            cl.Wc = aw < W ? (W - aw + SW - 1) / SW : 0;
            cl.Ngc = N * cl.Hc * cl.Wc;
            cl.Kc = cl.nrh * cl.nrw * p.Cq;
            cl.ntiles = rg::cdiv(cl.Ngc, BN);
            cl.d_nrw = make_fastdiv(cl.nrw);
            cl.d_hw = make_fastdiv(cl.Hc * cl.Wc);
            cl.d_w = make_fastdiv(cl.Wc);
            if (cl.ntiles > nt_max) nt_max = cl.ntiles;
            flops += 2.0 * C * (double)cl.Ngc * K * cl.nrh * cl.nrw;
        }
    p.Ng = 0; p.Kc = 0; p.n_tiles = nt_max;
    rg::ProfScope prof(rg::FAM_CONV_F8, stream, flops,
                       (double)N * P * Q * Kp + (double)C * KH * KW * Kp + 4.0 * N * C * H * W);
    const int slots = p.m_tiles * nt_max;
    const dim3 grid(SH * SW > 1 ? (unsigned)(8 * ((slots >> 3) + 1) * SH * SW) : (unsigned)slots, 1, 1);
    RG_F8_LAUNCH(1, grid);
    return rg::check_launch("rg_conv2d_f8_dgrad");
}

// ---- weight gradient: dw[K][C][KH][KW] fp32 from xq [C][H*W][Np] and dyq [K][P*Q][Np] (batch-contiguous layouts) --------------
namespace {
static void plan_f8_wgrad(int M, int Ng, int nk, int* bm, int* splits, int* per) {
    *bm = M <= 64 ? 64 : 128;
    const int tiles = rg::cdiv(M, *bm) * rg::cdiv(Ng, BN);
    int want = rg::cdiv(1024, tiles);
    if (want > nk / 8) want = nk / 8;          // >= 8 k-tiles (512 reduction bytes) per split
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    while (want > 1 && (int64_t)want * M * Ng * 4 >= (1ll << 31)) --want;
    *per = rg::cdiv(nk, want);
    *splits = rg::cdiv(nk, *per);
}
}  // namespace

extern "C" size_t rg_conv2d_f8_wgrad_workspace(int N, int C, int K, int KH, int KW, int P, int Q) {
    const int Np = pad16(N);
    const int nk = rg::cdiv(P * Q * (Np / 16), 4);
    int bm, splits, per;
    plan_f8_wgrad(K, C * KH * KW, nk, &bm, &splits, &per);
    return (size_t)splits * (size_t)K * (size_t)C * KH * KW * sizeof(float);
}

extern "C" int rg_conv2d_f8_wgrad(const void* xq, const void* dyq, const float* sx, const float* sdy, int fmt_x, int fmt_dy,
                                  float* dw, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                                  int P, int Q, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (int e = check_geom("rg_conv2d_f8_wgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(xq && dyq && sx && sdy && dw, "rg_conv2d_f8_wgrad: null tensor");
    RG_REQUIRE(fmt_x != fmt_dy || fmt_x == 0, "rg_conv2d_f8_wgrad: e5m2 x e5m2 is not instantiated");
    F8P p;
    fill_geom(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    const int Np = pad16(N);
    p.A = static_cast<const unsigned char*>(dyq);
    p.B = static_cast<const unsigned char*>(xq);
    p.a_bytes = (unsigned)((int64_t)K * P * Q * Np);
    p.b_bytes = (unsigned)((int64_t)C * H * W * Np);
    RG_REQUIRE(fits((int64_t)K * P * Q * Np) && fits((int64_t)C * H * W * Np), "rg_conv2d_f8_wgrad: operand exceeds 2 GiB");
    p.sa = sdy; p.sb = sx;
    p.M = K; p.Ng = C * KH * KW;
    p.Cq = Np / 16;
    p.Kc = P * Q * p.Cq;
    p.d_cq = make_fastdiv(p.Cq);
    const int nk = rg::cdiv(p.Kc, 4);
    int bm, splits, per;
    plan_f8_wgrad(p.M, p.Ng, nk, &bm, &splits, &per);
    const size_t need = (size_t)splits * p.M * (size_t)p.Ng * sizeof(float);
    if (!workspace || need > workspace_bytes) {
        rg::set_error("rg_conv2d_f8_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return RG_ERR_WORKSPACE;
    }
    p.splits = splits; p.ktiles_per_split = per;
    p.out = static_cast<float*>(workspace);
    p.out_bytes = (unsigned)need;
    p.m_tiles = rg::cdiv(p.M, bm);
    p.n_tiles = rg::cdiv(p.Ng, BN);
    const int fa = fmt_dy, fb = fmt_x;
    rg::ProfScope prof(rg::FAM_CONV_F8, stream, 2.0 * K * (double)C * KH * KW * N * P * Q,
                       (double)K * P * Q * Np + (double)C * H * W * Np + 4.0 * K * C * KH * KW);
    const dim3 grid(p.m_tiles * p.n_tiles, 1, splits);
    RG_F8_LAUNCH(2, grid);
    if (int e = rg::check_launch("rg_conv2d_f8_wgrad")) return e;
    const int64_t n = (int64_t)p.M * p.Ng;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
        hipLaunchKernelGGL(f8_splitk_reduce_vec_kernel, dim3((unsigned)rg::cdiv64(n >> 2, 64)), dim3(256), 0, stream,
                           static_cast<const float*>(workspace), dw, n, splits, sdy, sx);
    else
        hipLaunchKernelGGL(f8_splitk_reduce_kernel, dim3((unsigned)rg::cdiv64(n, 64)), dim3(256), 0, stream,
                           static_cast<const float*>(workspace), dw, n, splits, sdy, sx);
    return rg::check_launch("rg_conv2d_f8_wgrad(reduce)");
}
