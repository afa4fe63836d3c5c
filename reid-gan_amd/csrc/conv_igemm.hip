// Implicit-GEMM 2-D convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient, fp32 tensors.
// Arithmetic (RG_MATH 3, the default): every fp32 operand is split exactly into three bf16 pieces and each product is evaluated as
// six v_mfma_f32_32x32x16_bf16 partial products with fp32 accumulation (see "matrix arithmetic of one 16-deep k-tile" below);
// RG_MATH 1 builds the round-1/2 arithmetic on v_mfma_f32_32x32x2_f32 (a k-ordered fp32 fma chain).
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
// Loader variants (chosen on the host from the geometry):
//   * the reduction index k is wave-uniform in the gather loaders, so its (c,r,s) / (k,tap) decomposition runs on
//     the scalar unit (one readfirstlane); per lane only the bounds test and the address add remain;
//   * 1x1 / stride 1 / pad 0 layers (half of ResNet-50) load the pixel operand as float4 (16 B per lane);
//   * dgrad reads weights re-laid out as [K][KH*KW][C] (rg_weights_to_krsc, one tiny pass per layer per step) so the
//     weight operand is contiguous along the GEMM row and loads as float4 as well;
//   * layers whose M x N tile count cannot fill 256 CUs (8x4 and 16x8 maps at batch 32) split the reduction over
//     blockIdx.y; partial tiles go to a workspace and a finishing kernel sums them and applies the epilogue.
//
// Work decomposition: 256-thread workgroup = 4 wave64; block tile BM x BN x 16, each wave owns a
// (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles.  Operand tiles are staged global -> VGPR ->
// LDS, k-major ([16][BM+4] / [16][BN+4]) so that the MFMA operand reads (lane = row, lane>>5 = k)
// are bank-conflict free ds_read_b32; LDS is double buffered, one barrier per k-tile, and the
// next tile's global loads are issued before the current tile's MFMAs.  The flat tile id is
// remapped so that every XCD (private 4 MiB L2) works on a contiguous range of pixel tiles.
#include "rg_common.h"

#include <stdio.h>
#include <stdlib.h>
#include <array>
#include <atomic>
#include <map>
#include <mutex>

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
#ifndef RG_MATH
#define RG_MATH 3       // 3: split-bf16 arithmetic (three bf16 pieces per fp32 operand, six MFMA products); 1: fp32 MFMA (below)
#endif
#ifndef RG_WAVES
#if RG_MATH == 3
#define RG_WAVES 3      // the split fragments (3 x 4 registers per 32 x 16 operand block) need the 168-register budget
#else
#define RG_WAVES 4      // waves per SIMD the fwd / dgrad kernels are compiled for (register budget 512 / RG_WAVES; 4 = 128
                        // registers: 2-5 spilled dwords outside the k-loop, +0.8 % on the step over 3)
#endif
#endif
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
    const float* mask;   // same shape as the output or nullptr: after the residual add, v = mask > 0 ? v : 0 (the ReLU
                         // backward of the layer that produced this conv's input, whose output IS that input)
    float* rowsum;       // nullptr or [M][rowsum_cols]: per GEMM row, the sum of the FINAL values this wave stored (one
    int rowsum_cols;     // column per (class, n-tile, wave column) = the channel sums the BatchNorm fold of the layer
};                       // below needs, rg_bn_fold_wgrad `partials`), written in fixed order: deterministic

struct ConvP {
    const float* x;   // fwd: input, dgrad: dy, wgrad: input
    const float* w;   // fwd/dgrad: weights, wgrad: dy
    float* y;         // fwd: y, dgrad: dx, wgrad: dw or workspace
    Epilogue ep;
    int N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q;
    int M, Ng, Kg;
    int a_vec4;
    int wshift;          // wgrad, VEC instantiation: the im2col operand is a SHIFTED copy of x (stride 1): float4 loads at the tap's offset
    int m_tiles, n_tiles;
    FastDiv d_rs, d_kw, d_pq, d_q;
    // split-K (fwd / dgrad: partial tiles to `partial`; wgrad: to y)
    int ktiles_per_split, splits;
    float* partial;
    unsigned* arrive;    // nullptr, or one arrival counter per output tile (zero between launches): the LAST split of a tile to arrive
                         // sums the tile's partials and applies the epilogue itself — no finishing launch (splitk_arrive_finish)
    // buffer-resource sizes (bytes, < 2^31) of x / w / y / partial, and extra dividers for the (r,s)-major orders
    unsigned x_bytes, w_bytes, y_bytes, partial_bytes;
    FastDiv d_c, d_k;
};

struct DgradClass {
    int r0, s0, nrh, nrw, Hc, Wc, Ngc, Kgc, ntiles, poff;      // poff: first row-sum column block of the class
    int ktps, coff;                                            // split-K: k-tiles per split of THIS class, its first partial column
    FastDiv d_taps, d_nrw, d_hw, d_w;
};

struct DgradP {
    ConvP c;
    DgradClass cls[4];
    int ng_total;                                              // strided split-K: columns of one partial row (sum of the classes' Ngc)
};

template <int BM, int BN, int WM, int WN>
struct Tile {
    static constexpr int LDA = BM + LPAD;
    static constexpr int LDB = BN + LPAD;
    static constexpr int WTM = BM / WM;
    static constexpr int WTN = BN / WN;
    static constexpr int TBM = BM, TBN = BN, NTHREADS = 64 * WM * WN;
    static constexpr int TM = WTM / 32;
    static constexpr int TN = WTN / 32;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 waves per workgroup (8 for the plane-path kernels' 128 x 128 tile)");
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

// ---- matrix arithmetic of one 16-deep k-tile -------------------------------------------------------------------------------
// RG_MATH 3 (default): fp32 operands are split EXACTLY into three bf16 pieces each (x = hi + mid + lo with hi = bf16(x),
// mid = bf16(x - hi), lo = x - hi - mid, each rounded to nearest even: the residuals are exact fp32 subtractions and the last one
// has at most 8 significant bits, so it is a bf16 number; rounding rather than truncating keeps the residuals' signs independent
// of the operand's, so the dropped terms below do not add up to a bias — a truncating split underestimates every product by
// ~2^-24, measured as -4.7e-8 sum|a b| on same-sign data) and the product is evaluated as the six partial products whose weight
// is >= 2^-16 of the leading one
//     a*b ~ a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_mid b_mid + a_hi b_lo + a_lo b_hi)
// on v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in fp32, accumulation in fp32).  The three dropped products
// (mid*lo, lo*mid, lo*lo) are <= 2^-23 |a b| together: one fp32 rounding per product, i.e. the error model of the fp32 FMA chain
// the fp32 MFMA evaluates — measured against fp64 in tests/test_ops_gpu.py — at 6/16 of its matrix-pipe time (the bf16 MFMA
// issues 16x the FLOPs per cycle).  The split runs on the VALU after the fragment's ds_read_b32s (LDS tiles stay fp32, k-major,
// shared with RG_MATH 1), 4.5 VALU ops per element; small terms are accumulated first.
// RG_MATH 1: v_mfma_f32_32x32x2_f32 (a k-ordered fp32 fma chain), the round-1/2 arithmetic.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4r __attribute__((ext_vector_type(4)));

struct Split3 {
    int4r hi, mid, lo;      // 8 bf16 each: element j of the MFMA fragment = k index 8 * (lane >> 5) + j
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2r __attribute__((ext_vector_type(2)));
// two elements at a time: v_cvt_pk_bf16_f32 (round to nearest even) gives the packed pieces directly
__device__ __forceinline__ void split3_pair(float x0, float x1, int& hi, int& mid, int& lo) {
    const float2r x = {x0, x1};
    hi = __builtin_bit_cast(int, __builtin_convertvector(x, bf16x2));
    const float2r r = {x0 - __builtin_bit_cast(float, (unsigned)hi << 16), x1 - __builtin_bit_cast(float, (unsigned)hi & 0xffff0000u)};
    mid = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    const float2r l = {r[0] - __builtin_bit_cast(float, (unsigned)mid << 16), r[1] - __builtin_bit_cast(float, (unsigned)mid & 0xffff0000u)};
    lo = __builtin_bit_cast(int, __builtin_convertvector(l, bf16x2));
}

__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int h, m, l;
        split3_pair(x[2 * d], x[2 * d + 1], h, m, l);
        s.hi[d] = h; s.mid[d] = m; s.lo[d] = l;
    }
    return s;
}

__device__ __forceinline__ floatx16 mfma_bf16(const int4r& a, const int4r& b, const floatx16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc += A(32 x 16) * B(16 x 32) in split arithmetic, small terms first
__device__ __forceinline__ void mma_split3(const Split3& a, const Split3& b, floatx16& acc) {
    acc = mfma_bf16(a.lo, b.hi, acc);
    acc = mfma_bf16(a.hi, b.lo, acc);
    acc = mfma_bf16(a.mid, b.mid, acc);
    acc = mfma_bf16(a.mid, b.hi, acc);
    acc = mfma_bf16(a.hi, b.mid, acc);
    acc = mfma_bf16(a.hi, b.hi, acc);
}

#ifndef RG_PINSCHED
#define RG_PINSCHED 1
#endif
#if RG_PINSCHED
#define RG_PIN() __builtin_amdgcn_sched_barrier(0)      // keep the (MFMA, split pair) groups in source order
#else
#define RG_PIN()
#endif

#if RG_MATH == 3
// One 16-deep k-step of a (TM x 32) x (TN x 32) wave tile from fp32 operands in LDS: ra(i, q) / rb(j, q) read element q (k index
// 8 * (lane >> 5) + q) of this lane's row of A block i / column of B block j.
// Software-pipelined by hand: the matrix pipe and the VALU do not overlap across the waves of a SIMD here (the co-resident
// workgroups run in phase: PMC showed VALU-busy + MFMA-busy = kernel time), so each wave hides its own split work behind its own
// MFMAs: block order (0,0), (1,0), .., (0,1), .. needs one new fragment per block; while the six MFMAs of a block issue, the
// fragment of the NEXT block is split, one element pair (9 VALU instructions) behind each of the first four.
template <int TM, int TN, typename RA, typename RB>
__device__ __forceinline__ void mma_kstep(RA ra, RB rb, floatx16 (&acc)[TM][TN]) {
    float xa[TM][8], xb[TN][8];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) xa[i][q] = ra(i, q);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) xb[j][q] = rb(j, q);
    Split3 a[TM], b[TN];
    a[0] = split3(xa[0]);
    b[0] = split3(xb[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // fragment the next block needs first: a[i + 1] in the first column, b[j + 1] at the end of a column
            const bool na = (j == 0 && i + 1 < TM), nb = (i + 1 == TM && j + 1 < TN);
            const int ia = i + 1 < TM ? i + 1 : 0, jb = j + 1 < TN ? j + 1 : 0;
            auto pair = [&](int d) {
                int h, m, l;
                if (na) {
                    split3_pair(xa[ia][2 * d], xa[ia][2 * d + 1], h, m, l);
                    a[ia].hi[d] = h; a[ia].mid[d] = m; a[ia].lo[d] = l;
                } else if (nb) {
                    split3_pair(xb[jb][2 * d], xb[jb][2 * d + 1], h, m, l);
                    b[jb].hi[d] = h; b[jb].mid[d] = m; b[jb].lo[d] = l;
                }
            };
            floatx16& c = acc[i][j];
            c = mfma_bf16(a[i].lo, b[j].hi, c);  pair(0);  RG_PIN();
            c = mfma_bf16(a[i].hi, b[j].lo, c);  pair(1);  RG_PIN();
            c = mfma_bf16(a[i].mid, b[j].mid, c);  pair(2);  RG_PIN();
            c = mfma_bf16(a[i].mid, b[j].hi, c);  pair(3);  RG_PIN();
            c = mfma_bf16(a[i].hi, b[j].mid, c);
            c = mfma_bf16(a[i].hi, b[j].hi, c);
        }
}
#endif

// One 16-deep k-tile of MFMAs.  `hook(q)`, q = 0..3, is called behind the last matrix instructions: the kernels use it to write
// the NEXT tile's staged registers into the other LDS buffer, so those ds_writes (and the vmcnt wait in front of them) issue in
// the shadow of the MFMAs instead of after them.
template <typename T, typename Hook>
__device__ __forceinline__ void mma_tile(const float (*As)[T::LDA], const float (*Bs)[T::LDB],
                                         floatx16 (&acc)[T::TM][T::TN], int wm, int wn, int lane, Hook hook) {
    const int l32 = lane & 31, kh = lane >> 5;
#if RG_MATH == 3
    static_assert(BK == 16, "one bf16 MFMA k-step per LDS tile");
    mma_kstep<T::TM, T::TN>([&](int i, int q) { return As[8 * kh + q][wm * T::WTM + i * 32 + l32]; },
                            [&](int j, int q) { return Bs[8 * kh + q][wn * T::WTN + j * 32 + l32]; }, acc);
    hook(0); hook(1); hook(2); hook(3);
#else
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
        // stores of the next tile behind the LAST k-step: the global loads issued at the top of the tile get 7/8 of its
        // matrix work as cover before their first use
        if (ks == BK / 2 - 1) { hook(0); hook(1); hook(2); hook(3); }
    }
#endif
}

// true when element e of a CNT-element staging array belongs to quarter q (q < 0: every quarter)
__device__ __forceinline__ constexpr bool in_quarter(int e, int cnt, int q) { return q < 0 || (e * 4) / cnt == q; }


// ---- raw buffer access: 32-bit byte offsets from a wave-uniform base, hardware range check (a load beyond
// num_records returns 0, a store is dropped).  An invalid lane simply carries OOB as its offset: no branches.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef int int4v __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;      // every tensor is < 2^31 bytes (checked on the host)

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float4 bload4(rsrc_t r, unsigned off) {
    // bit_cast of the builtin's own 16-byte vector type (an implicit conversion to an ext_vector splats lane 0)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void bstore(rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, 0);
}
// AUX = 16: sc1, a write-through store (leaves the XCD's L2 for memory at once: what another XCD's workgroup may read in this launch)
template <int AUX>
__device__ __forceinline__ void bstore_aux(rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, AUX);
}
template <int I> struct AuxTag { static constexpr int value = I; };

template <typename T>
__device__ __forceinline__ void zero_acc(floatx16 (&acc)[T::TM][T::TN]) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// Fused epilogue y = mask(act(acc * scale[m] + shift[m] + res)) for a tile whose column j of the wave starts at byte
// offset ob[j] (OOB when outside) and whose GEMM rows are `rstride` bytes apart.  Branch-free: per-row scale / shift
// are broadcast buffer loads shared by the TN column blocks, residual / mask are buffer loads at the store offset
// issued RB at a time before their first use (OOB lanes read 0 and their stores are dropped by the hardware).
template <typename T, int ACT>
__device__ __forceinline__ void store_tile_epilogue(const ConvP& p, const floatx16 (&acc)[T::TM][T::TN], const unsigned (&ob)[T::TN],
                                                    unsigned rstride, int mrow0, int pc) {
    const rsrc_t ro = make_rsrc(p.y, p.y_bytes);
    const rsrc_t rr = make_rsrc(p.ep.res ? (const void*)p.ep.res : (const void*)p.y, p.ep.res ? p.y_bytes : 0u);
    const rsrc_t rm = make_rsrc(p.ep.mask ? (const void*)p.ep.mask : (const void*)p.y, p.ep.mask ? p.y_bytes : 0u);
    const rsrc_t rsc = make_rsrc(p.ep.scale ? p.ep.scale : p.ep.shift, p.ep.scale ? (unsigned)p.M * 4u : 0u);
    const rsrc_t rsh = make_rsrc(p.ep.shift ? p.ep.shift : p.ep.scale, p.ep.shift ? (unsigned)p.M * 4u : 0u);
    const bool has_scale = p.ep.scale != nullptr, has_res = p.ep.res != nullptr, has_mask = p.ep.mask != nullptr;
    constexpr int RB = 8;          // rows per batch: bounds the live VGPRs of the epilogue
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int rb = 0; rb < 16; rb += RB) {
            float sc[RB], sh[RB], rs[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) rs[q] = 0.f;
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = rb + q;
                const unsigned moff = (unsigned)(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2)) * 4u;
                sc[q] = has_scale ? bload(rsc, moff) : 1.f;      // zero-sized resources return 0 for every lane
                sh[q] = bload(rsh, moff);
            }
#pragma unroll
            for (int j = 0; j < T::TN; ++j) {
                // offsets are recomputed at each use (one mad + select) rather than kept live across the loads
                auto off_of = [&](int q) -> unsigned {
                    const int r = rb + q;
                    const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                    return (mrow0 + mo < p.M) ? ob[j] + (unsigned)mo * rstride : OOB;
                };
                float rv[RB];
                if (has_res) {
#pragma unroll
                    for (int q = 0; q < RB; ++q) rv[q] = bload(rr, off_of(q));
                }
                const bool colok = ob[j] != OOB;
                if (has_mask) {                      // mask folded into the residual registers: sign carries it
                    float mv[RB];
#pragma unroll
                    for (int q = 0; q < RB; ++q) mv[q] = bload(rm, off_of(q));
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        float v = acc[i][j][rb + q] * sc[q] + sh[q];
                        if (has_res) v += rv[q];
                        if (ACT == RG_ACT_RELU) v = fmaxf(v, 0.f);
                        if (ACT == RG_ACT_LEAKY) v = v > 0.f ? v : v * p.ep.slope;
                        if (ACT == RG_ACT_TANH) v = tanhf(v);
                        v = mv[q] > 0.f ? v : 0.f;
                        bstore(ro, off_of(q), v);
                        if (p.ep.rowsum) rs[q] += colok ? v : 0.f;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        float v = acc[i][j][rb + q] * sc[q] + sh[q];
                        if (has_res) v += rv[q];
                        if (ACT == RG_ACT_RELU) v = fmaxf(v, 0.f);
                        if (ACT == RG_ACT_LEAKY) v = v > 0.f ? v : v * p.ep.slope;
                        if (ACT == RG_ACT_TANH) v = tanhf(v);
                        bstore(ro, off_of(q), v);
                        if (p.ep.rowsum) rs[q] += colok ? v : 0.f;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the next batch's loads from being hoisted (VGPR pressure)
            }
            if (p.ep.rowsum) {                           // uniform: 32-lane butterfly per row, lane 0 of each half writes
                const int lane = threadIdx.x & 63;
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    float t = rs[q];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                    const int r = rb + q;
                    const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    if ((lane & 31) == 0 && m < p.M) p.ep.rowsum[(int64_t)m * p.ep.rowsum_cols + pc] = t;
                }
            }
        }
}

template <typename T>
__device__ __forceinline__ void store_tile_epilogue_any(const ConvP& p, const floatx16 (&acc)[T::TM][T::TN],
                                                        const unsigned (&ob)[T::TN], unsigned rstride, int mrow0, int pc) {
    switch (p.ep.act) {      // uniform
        case RG_ACT_RELU: store_tile_epilogue<T, RG_ACT_RELU>(p, acc, ob, rstride, mrow0, pc); break;
        case RG_ACT_LEAKY: store_tile_epilogue<T, RG_ACT_LEAKY>(p, acc, ob, rstride, mrow0, pc); break;
        case RG_ACT_TANH: store_tile_epilogue<T, RG_ACT_TANH>(p, acc, ob, rstride, mrow0, pc); break;
        default: store_tile_epilogue<T, RG_ACT_NONE>(p, acc, ob, rstride, mrow0, pc); break;
    }
}

// ---- split-K finish, four consecutive columns of one GEMM row (shared by conv_splitk_finish_vec_kernel and the in-kernel finish
// below, so that the two produce the same bits): left-to-right sum over the splits, then the epilogue ----
__device__ __forceinline__ float4 splitk_sum4(const float4* __restrict__ p4, int64_t sstride4, int64_t i, int splits) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        const float4 a = p4[(int64_t)s * sstride4 + i], b = p4[(int64_t)(s + 1) * sstride4 + i];
        const float4 c = p4[(int64_t)(s + 2) * sstride4 + i], d = p4[(int64_t)(s + 3) * sstride4 + i];
        v.x = (((v.x + a.x) + b.x) + c.x) + d.x; v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
        v.z = (((v.z + a.z) + b.z) + c.z) + d.z; v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; s < splits; ++s) {
        const float4 a = p4[(int64_t)s * sstride4 + i];
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    return v;
}

__device__ __forceinline__ void splitk_epilogue_store4(float* __restrict__ out, int64_t o, int m, float4 v, const Epilogue& ep) {
    if (ep.scale) { const float sc = ep.scale[m]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
    if (ep.shift) { const float sh = ep.shift[m]; v.x += sh; v.y += sh; v.z += sh; v.w += sh; }
    if (ep.res) {
        const float4 r = *reinterpret_cast<const float4*>(ep.res + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    v.x = rg_apply_act(v.x, ep.act, ep.slope); v.y = rg_apply_act(v.y, ep.act, ep.slope);
    v.z = rg_apply_act(v.z, ep.act, ep.slope); v.w = rg_apply_act(v.w, ep.act, ep.slope);
    if (ep.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(ep.mask + o);
        if (!(mk.x > 0.f)) v.x = 0.f;
        if (!(mk.y > 0.f)) v.y = 0.f;
        if (!(mk.z > 0.f)) v.z = 0.f;
        if (!(mk.w > 0.f)) v.w = 0.f;
    }
    *reinterpret_cast<float4*>(out + o) = v;
}

// Split-K without the finishing launch (p.arrive != nullptr; the host sets it only when Ng % 4 == 0, PIX % 4 == 0 and every pointer is
// 16-byte aligned).  The L2s of the eight XCDs are not coherent with each other inside a kernel and a CU's L1 is never refreshed by
// other CUs' stores, so the hand-off follows the counter form of the split-K seam: every workgroup of a tile stores its raw
// accumulators to partial[split] WRITE-THROUGH (sc1: no L2-wide release fence; __threadfence() here measured +27 us per launch),
// every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, and one lane counts the workgroup in
// on the tile's arrival counter (agent-scope atomic).  The workgroup that finds splits - 1 earlier arrivals is the last: one lane's
// agent-scope acquire (drops this CU's stale L1 lines), the wait for it, a barrier — then all waves read the tile's partials back in
// split order 0, 1, 2, ... (the finishing kernel's summation order: the result does not depend on which split came last) and write
// the finished outputs.  atomicInc wraps the counter to zero on that last arrival: clean for the next launch without a memset (the
// caller zeroes the counters once, rg_conv_splitk_arrivals).  Nobody waits for anybody: no workgroup can stall on one that has not
// been scheduled yet.
template <typename T>
__device__ __forceinline__ void splitk_arrive_finish(const ConvP& p, int m0, int n0, int Ng, int PIX, const FastDiv& d_pix) {
    __shared__ unsigned s_prev;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave: its write-through partial stores have left
    __syncthreads();                                        // ... for all waves (and nobody reads operand LDS any more)
    if (threadIdx.x == 0) {
        const unsigned prev = atomicInc(p.arrive + blockIdx.x, (unsigned)p.splits - 1u);
        if (prev == (unsigned)p.splits - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_prev = prev;
    }
    __syncthreads();
    if (s_prev != (unsigned)p.splits - 1u) return;          // uniform
    constexpr int C4 = T::TBN / 4, RSTEP = T::NTHREADS / C4;
    static_assert(T::NTHREADS % C4 == 0, "whole rows per pass");
    const int c4 = threadIdx.x % C4;
    const int n = n0 + 4 * c4;
    if (n >= Ng) return;
    const int ng4 = Ng >> 2;
    const int64_t sstride4 = (int64_t)p.M * ng4;
    const float4* p4 = reinterpret_cast<const float4*>(p.partial);
    const int im = fdiv(n, d_pix);
    const int pix = n - im * PIX;
    for (int r = threadIdx.x / C4; r < T::TBM; r += RSTEP) {
        const int m = m0 + r;
        if (m >= p.M) break;
        const float4 v = splitk_sum4(p4, sstride4, (int64_t)m * ng4 + (n >> 2), p.splits);
        splitk_epilogue_store4(p.y, ((int64_t)im * p.M + m) * PIX + pix, m, v, p.ep);
    }
}

// Epilogue for outputs laid out [img][M][PIX] with n = img*PIX + pix (fwd: PIX = P*Q; stride-1 dgrad: PIX = H*W).
// With split-K the raw accumulators go to partial[(split*M + m)*Ng + n] instead.  Buffer stores: one VALU add per
// element, lanes outside the tensor carry OOB and are dropped by the hardware.
template <typename T>
__device__ __forceinline__ void store_tile_nchw(const ConvP& p, const floatx16 (&acc)[T::TM][T::TN], int m0, int n0,
                                                int wm, int wn, int lane, int Ng, int PIX, const FastDiv& d_pix,
                                                int split, int pcol = 0) {
    const int l32 = lane & 31, kh = lane >> 5;
    const int mrow0 = m0 + wm * T::WTM + 4 * kh;
    const bool plain = !p.ep.scale && !p.ep.shift && !p.ep.res && !p.ep.mask && !p.ep.rowsum && p.ep.act == RG_ACT_NONE;
    if (p.partial || plain) {
        const rsrc_t ro = p.partial ? make_rsrc(p.partial, p.partial_bytes) : make_rsrc(p.y, p.y_bytes);
        const unsigned rstride = (p.partial ? (unsigned)Ng : (unsigned)PIX) * 4u;    // bytes between GEMM rows
        auto store_raw = [&](auto aux_tag) {
#pragma unroll
            for (int j = 0; j < T::TN; ++j) {
                const int nn = n0 + wn * T::WTN + j * 32 + l32;
                unsigned ob = OOB;
                if (nn < Ng) {
                    if (p.partial) {
                        ob = (unsigned)(((split * p.M + mrow0) * (int64_t)Ng + nn) * 4);
                    } else {
                        const int im = fdiv(nn, d_pix);
                        ob = (unsigned)((((int64_t)im * p.M + mrow0) * PIX + (nn - im * PIX)) * 4);
                    }
                }
#pragma unroll
                for (int i = 0; i < T::TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                        const unsigned off = (mrow0 + mo < p.M) ? ob + (unsigned)mo * rstride : OOB;
                        bstore_aux<decltype(aux_tag)::value>(ro, off, acc[i][j][r]);
                    }
            }
        };
        if (p.partial && p.arrive) {                            // uniform
            store_raw(AuxTag<16>());
            splitk_arrive_finish<T>(p, m0, n0, Ng, PIX, d_pix);
        } else {
            store_raw(AuxTag<0>());
        }
        return;
    }
    unsigned ob[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        ob[j] = OOB;
        if (nn < Ng) {
            const int im = fdiv(nn, d_pix);
            ob[j] = (unsigned)((((int64_t)im * p.M + mrow0) * PIX + (nn - im * PIX)) * 4);
        }
    }
    store_tile_epilogue_any<T>(p, acc, ob, (unsigned)PIX * 4u, mrow0, pcol);
}

// raw accumulators of a tile to partial[(split * M + m) * ncols + col0 + n] (strided data gradient with split-K: the classes'
// columns side by side, col0 = the class' first column)
template <typename T>
__device__ __forceinline__ void store_tile_partial_cols(const ConvP& p, const floatx16 (&acc)[T::TM][T::TN], int m0, int n0, int wm,
                                                        int wn, int lane, int Ng, int col0, int ncols, int split) {
    const int l32 = lane & 31, kh = lane >> 5;
    const int mrow0 = m0 + wm * T::WTM + 4 * kh;
    const rsrc_t ro = make_rsrc(p.partial, p.partial_bytes);
    const unsigned rstride = (unsigned)ncols * 4u;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        const unsigned ob = nn < Ng ? (unsigned)((((int64_t)split * p.M + mrow0) * ncols + col0 + nn) * 4) : OOB;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                bstore(ro, (mrow0 + mo < p.M && ob != OOB) ? ob + (unsigned)mo * rstride : OOB, acc[i][j][r]);
            }
    }
}

// A operand loader shared by fwd (weights [M][Kg], k contiguous): float4 along k (AVEC) or scalar.
template <int BM, bool AVEC>
struct ALoadK {
    static constexpr int NA = AVEC ? ((BM * 4 + NT - 1) / NT) : (BM * BK / NT);
    unsigned off[NA];
    int kq[NA];
    __device__ __forceinline__ void init(int tid, int m0, int M, int Kg) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + NT * i;
            const int row = AVEC ? (v >> 2) : (v >> 4);
            kq[i] = AVEC ? (v & 3) * 4 : (v & 15);
            const bool ok = (AVEC ? v < BM * 4 : v < BM * BK) && (m0 + row < M);
            off[i] = ok ? (unsigned)(((int64_t)(m0 + row) * Kg + kq[i]) * 4) : OOB;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// forward.  BMODE 0: generic gather, reduction order (c, r, s), weights [K][C][KH][KW]
//           BMODE 1: (r, s)-major order k' = rs*C + c, weights [K][KH*KW][C], C % 16 == 0: one bounds test per tile
//           BMODE 2: 1x1 / stride 1 / pad 0 with H*W % 4 == 0: pixel operand as float4
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BMODE, bool AVEC>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(RG_WAVES))) void conv_fwd_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    static_assert(BN >= 64, "the gather loader needs a wave-uniform k");
    __shared__ __attribute__((aligned(16))) float As[2][BK][T::LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][T::LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;
    const int HW = p.H * p.W;
    const int RS = p.KH * p.KW;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);

    ALoadK<BM, AVEC> al;
    al.init(tid, m0, p.M, p.Kg);

    // ---- B operand set-up ----
    constexpr int BKSTEP = NT / BN > 0 ? NT / BN : 1;
    const int bcol = tid % BN;
    const int bk0 = __builtin_amdgcn_readfirstlane(tid / BN);
    constexpr int BV = BN / 4;
    constexpr int BVSTEP = NT / BV;
    constexpr int BVCNT = (BV * BK + NT - 1) / NT;
    const int vcol = tid % BV, vrow0 = tid / BV;

    bool bvalid;
    int h0 = 0, w0 = 0, pixb = 0;     // pixb: element index of (img, c=0, h0, w0); may be "negative" inside padding
    unsigned bvoff = OOB;             // BMODE 2: byte offset of (img, k = vrow0, pix)
    if (BMODE == 2) {
        const int n = n0 + 4 * vcol;
        bvalid = n < p.Ng && vrow0 < BK;
        if (bvalid) {
            const int img = fdiv(n, p.d_pq);
            bvoff = (unsigned)((((int64_t)img * p.C + vrow0) * HW + (n - img * HW)) * 4);
        }
    } else {
        const int n = n0 + bcol;
        bvalid = n < p.Ng;
        if (bvalid) {
            const int img = fdiv(n, p.d_pq);
            const int pq = n - img * p.P * p.Q;
            const int pp = fdiv(pq, p.d_q);
            const int qq = pq - pp * p.Q;
            h0 = pp * p.SH - p.PH;
            w0 = qq * p.SW - p.PW;
            pixb = img * p.C * HW + h0 * p.W + w0;
        }
    }

    float ra[AVEC ? 4 * ALoadK<BM, AVEC>::NA : ALoadK<BM, AVEC>::NA];
    float rb[BMODE == 2 ? 1 : T::BCNT];
    float4 rbv[BMODE == 2 ? BVCNT : 1];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        const unsigned kb4 = (unsigned)kbase * 4u;
        const bool ktail = kbase + BK > p.Kg;                    // uniform; only the last tile of ragged Kg
#pragma unroll
        for (int i = 0; i < ALoadK<BM, AVEC>::NA; ++i) {
            unsigned o = al.off[i] + kb4;
            if (ktail && kbase + al.kq[i] >= p.Kg) o = OOB;
            if (AVEC) {
                const float4 t = bload4(rw, o);
                ra[4 * i + 0] = t.x; ra[4 * i + 1] = t.y; ra[4 * i + 2] = t.z; ra[4 * i + 3] = t.w;
            } else {
                ra[i] = bload(rw, o);
            }
        }
        if (BMODE == 2) {
            const unsigned kstride = (unsigned)HW * 4u;
#pragma unroll
            for (int i = 0; i < BVCNT; ++i) {
                unsigned o = bvoff + (unsigned)(kbase + i * BVSTEP) * kstride;
                if (ktail && kbase + vrow0 + i * BVSTEP >= p.Kg) o = OOB;
                rbv[i] = bload4(rx, o);
            }
        } else if (BMODE == 1) {
            const int rs = fdiv(kbase, p.d_c);                   // scalar: the whole tile shares (r, s)
            const int c0 = kbase - rs * p.C;
            const int r = fdiv(rs, p.d_kw);
            const int s = rs - r * p.KW;
            const int h = h0 + r, w = w0 + s;
            const bool ok = bvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const unsigned o0 = ok ? (unsigned)(pixb + r * p.W + s + (c0 + bk0) * HW) * 4u : OOB;
            const unsigned cstride = (unsigned)(BKSTEP * HW) * 4u;
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i) rb[i] = bload(rx, o0 + (unsigned)i * cstride);
        } else {
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i) {
                const int k = kbase + bk0 + i * BKSTEP;          // wave-uniform -> scalar unit
                const int c = fdiv(k, p.d_rs);
                const int rs = k - c * RS;
                const int r = fdiv(rs, p.d_kw);
                const int s = rs - r * p.KW;
                const int h = h0 + r, w = w0 + s;
                const bool ok = bvalid && k < p.Kg && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                rb[i] = bload(rx, ok ? (unsigned)(pixb + c * HW + r * p.W + s) * 4u : OOB);
            }
        }
    };
    auto store_tile = [&](int buf, int q) {
        constexpr int NA = ALoadK<BM, AVEC>::NA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + NT * i;
            if (AVEC) {
                const int row = v >> 2, kq = (v & 3) * 4;
                if (v < BM * 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (in_quarter(4 * i + j, 4 * NA, q)) As[buf][kq + j][row] = ra[4 * i + j];
                }
            } else {
                if (v < BM * BK && in_quarter(i, NA, q)) As[buf][v & 15][v >> 4] = ra[i];
            }
        }
        if (BMODE == 2) {
#pragma unroll
            for (int i = 0; i < BVCNT; ++i) {
                const int kk = vrow0 + i * BVSTEP;
                if (kk < BK && in_quarter(i, BVCNT, q)) *reinterpret_cast<float4*>(&Bs[buf][kk][4 * vcol]) = rbv[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i)
                if (in_quarter(i, T::BCNT, q)) Bs[buf][bk0 + i * BKSTEP][bcol] = rb[i];
        }
    };

    const int nk = (p.Kg + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk) kt_end = nk;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0, -1);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int q) {
            if (has_next) store_tile(cur ^ 1, q);
        });
        __syncthreads();
        cur ^= 1;
    }
    store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, p.Ng, p.P * p.Q, p.d_pq, split);
}

#include "conv_planes.h"

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 with TAP REUSE (forward, and the data gradient of such a layer, which is the same convolution with
// the filter taps flipped and the channel roles swapped).
// The generic kernels gather the pixel operand once per filter tap: nine L2 -> LDS passes over the same activations.  With a
// 128-pixel tile that is 12 KB of operands per 0.26 MFLOP, and at ~6 TB/s of L2 delivery the 64 / 128-channel layers of the trunk
// are bound by that traffic, not by the matrix pipe.  Here the reduction runs channel-block outer, tap inner: for every block of
// 16 reduction channels the workgroup loads ONE halo tile of the activations — its 128 output pixels (whole rows of one image, or
// whole small images) plus the one-pixel border, zero outside the image — and all nine taps read their pixel operand from that
// tile at a constant offset ((r-1) * (W+2) + (s-1)); only the filter operand (16 x BM floats) is fetched per tap.  Pixel-operand
// traffic drops 5-6 x and each barrier interval holds the same 8 MFMA k-steps as before but only the filter loads.
// Requirements (checked on the host): W in {4..64} a power of two, tile rows dividing H or whole images per tile, reduction
// channels % 16 == 0, filters in the [K][9][C] copy.
// ---------------------------------------------------------------------------------------------
struct HaloP {
    ConvP c;             // x: input tensor of the convolution being computed (fwd: x, dgrad: dy); w: [K][9][C] filters; y: output
    int Cred;            // channels of that input tensor (the reduction): fwd C, dgrad K
    int HP, Wh, slab;    // halo positions per tile, halo row length W + 2, positions per image of the tile (rows + 2) * Wh
    int cblocks, cb_per_split;
    FastDiv d_hp, d_slab, d_wh, d_hw, d_w;
};

// 128-row tiles: 184-194 registers; at the 168 of three waves per SIMD the compiler spilled 23-46 of them to scratch inside the
// loop (two waves per SIMD: forward +1 %, data gradient +4.5 % on the trunk's 3x3 layers, same box); 64-row tiles fit at three
template <int BM, bool DGRAD, int NH>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(BM == 128 ? 2 : 3))) void conv3x3_halo_kernel(const HaloP hp) {
    using T = Tile<BM, 128, 2, 2>;
    const ConvP& p = hp.c;
    __shared__ __attribute__((aligned(16))) float As[2][BK][T::LDA];
    __shared__ __attribute__((aligned(16))) float Hs[2][NH * NT];     // [16 channels][HP positions], flat

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, kh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * 128;
    const int split = blockIdx.y;
    const int HW = p.H * p.W, HP = hp.HP, Wh = hp.Wh;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);

    // tile origin: 128 % W == 0, so a tile starts at the beginning of a row (and covers whole rows / whole images)
    const int img0 = fdiv(n0, hp.d_hw);
    const int h0 = fdiv(n0 - img0 * HW, hp.d_w);

    // ---- pixel-operand positions of this lane inside the halo tile (two 32-pixel column blocks of the wave) ----
    int pos[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int pl = wn * T::WTN + j * 32 + l32;
        const int il = fdiv(pl, hp.d_hw);                      // 0 when the tile lies inside one image (H*W >= 128)
        const int rem = pl - il * HW;
        const int hl = fdiv(rem, hp.d_w);
        pos[j] = il * hp.slab + (hl + 1) * Wh + (rem - hl * p.W) + 1;
    }

    // ---- halo loader: element f = tid + 256 i of [16][HP]; byte offset of channel block 0, OOB outside the image ----
    unsigned hoff[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int f = tid + NT * i;
        const int ch = fdiv(f, hp.d_hp);
        const int ps = f - ch * HP;
        const int il = fdiv(ps, hp.d_slab);
        const int r2 = ps - il * hp.slab;
        const int hh = fdiv(r2, hp.d_wh);
        const int ww = r2 - hh * Wh - 1;
        const int img = img0 + il, h = h0 + hh - 1;
        const bool ok = ch < BK && img < p.N && (unsigned)h < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
        hoff[i] = ok ? (unsigned)((((int64_t)img * hp.Cred + ch) * p.H + h) * p.W + ww) * 4u : OOB;
    }
    const unsigned cbstride = (unsigned)(BK * HW) * 4u;        // bytes between channel blocks of the input tensor

    // ---- filter-operand loader ----
    constexpr int NA = BM / 64;                                // float4 per thread and tile (BM * 16 / 4 / 256)
    unsigned aoff[NA];
    int arow[NA], akq[NA];                                     // fwd: (row m, k quad); dgrad: (k row, m quad)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int v = tid + NT * i;
        if (!DGRAD) {
            arow[i] = v >> 2;
            akq[i] = (v & 3) * 4;
            aoff[i] = (m0 + arow[i] < p.M) ? (unsigned)(((int64_t)(m0 + arow[i]) * 9 * hp.Cred + akq[i]) * 4) : OOB;
        } else {
            arow[i] = v / (BM / 4);
            akq[i] = (v - arow[i] * (BM / 4)) * 4;
            aoff[i] = (m0 + akq[i] < p.M) ? (unsigned)(((int64_t)arow[i] * 9 * p.M + m0 + akq[i]) * 4) : OOB;
        }
    }
    float4 ra[NA];
    float hv[NH];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_a = [&](int cb, int t) {
        // fwd: w[m][t][cb*16 + kq..]; dgrad: w[cb*16 + kk][8 - t][m..] (the flipped tap of the transposed filter)
        const unsigned kb4 = DGRAD ? (unsigned)(((int64_t)cb * BK * 9 + (8 - t)) * p.M) * 4u
                                   : (unsigned)(t * hp.Cred + cb * BK) * 4u;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = bload4(rw, aoff[i] == OOB ? OOB : aoff[i] + kb4);
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!DGRAD) {
                As[buf][akq[i] + 0][arow[i]] = ra[i].x;
                As[buf][akq[i] + 1][arow[i]] = ra[i].y;
                As[buf][akq[i] + 2][arow[i]] = ra[i].z;
                As[buf][akq[i] + 3][arow[i]] = ra[i].w;
            } else {
                *reinterpret_cast<float4*>(&As[buf][arow[i]][akq[i]]) = ra[i];
            }
        }
    };
    auto load_h = [&](int cb) {
        const unsigned o = (unsigned)cb * cbstride;
#pragma unroll
        for (int i = 0; i < NH; ++i) hv[i] = bload(rx, hoff[i] == OOB ? OOB : hoff[i] + o);
    };
    auto store_h = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NH; ++i) Hs[buf][tid + NT * i] = hv[i];
    };

    const int cb_begin = split * hp.cb_per_split;
    int cb_end = cb_begin + hp.cb_per_split;
    if (cb_end > hp.cblocks) cb_end = hp.cblocks;
    if (cb_begin < cb_end) {
        load_h(cb_begin);
        load_a(cb_begin, 0);
        store_h(0);
        store_a(0);
    }
    __syncthreads();
    int ab = 0, hb = 0;
    for (int cb = cb_begin; cb < cb_end; ++cb) {
        const bool more_cb = cb + 1 < cb_end;
        for (int t = 0; t < 9; ++t) {
            const bool last = !more_cb && t == 8;
            if (t == 0 && more_cb) load_h(cb + 1);             // lands during the nine taps of this block
            if (!last) load_a(t == 8 ? cb + 1 : cb, t == 8 ? 0 : t + 1);
            const int r = (t * 11) >> 5;                       // t / 3 for t < 9
            const int toff = (r - 1) * Wh + (t - 3 * r - 1);
            const float* hsb = Hs[hb];
#if RG_MATH == 3
            mma_kstep<T::TM, T::TN>([&](int i, int q) { return As[ab][8 * kh + q][wm * T::WTM + i * 32 + l32]; },
                                    [&](int j, int q) { return hsb[(8 * kh + q) * HP + pos[j] + toff]; }, acc);
#else
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int k = 2 * ks + kh;
                float a[T::TM], b[T::TN];
#pragma unroll
                for (int i = 0; i < T::TM; ++i) a[i] = As[ab][k][wm * T::WTM + i * 32 + l32];
#pragma unroll
                for (int j = 0; j < T::TN; ++j) b[j] = hsb[k * HP + pos[j] + toff];
#pragma unroll
                for (int i = 0; i < T::TM; ++i)
#pragma unroll
                    for (int j = 0; j < T::TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
#endif
            if (!last) store_a(ab ^ 1);
            if (t == 8 && more_cb) store_h(hb ^ 1);
            __syncthreads();
            ab ^= 1;
        }
        hb ^= 1;
    }
    store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, p.Ng, HW, hp.d_hw, split, nt * 2 + wn);
}

// ---------------------------------------------------------------------------------------------
// data gradient (also the forward of ConvTranspose2d), one GEMM per stride-parity class.
// MODE 0: weights [K][C][KH][KW], reduction order (ko, tap), scalar loads (any geometry)
// MODE 1: weights [K][KH*KW][C] (== the original tensor for 1x1), tap-major order k' = tap*K + ko, K % 16 == 0 and
//         C % 4 == 0: weight operand float4 along C, one bounds test per tile for dy
// MODE 2: MODE 1 layout + 1x1 / stride 1 / pad 0 with P*Q % 4 == 0: dy loads as float4 too (any K)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(RG_WAVES))) void conv_dgrad_kernel(const DgradP dp) {
    using T = Tile<BM, BN, WM, WN>;
    static_assert(BN >= 64, "the gather loader needs a wave-uniform k");
    __shared__ __attribute__((aligned(16))) float As[2][BK][T::LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][T::LDB];
    const ConvP& p = dp.c;
    const int ci = blockIdx.z;
    const DgradClass& cl = dp.cls[ci];
    const int ah = ci / p.SW, aw = ci % p.SW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int nwg = p.m_tiles * cl.ntiles;
    if ((int)blockIdx.x >= nwg) return;
    if (p.partial && (p.SH > 1 || p.SW > 1) && cl.Kgc <= 0) return;      // strided split-K: the finisher writes tap-less classes itself
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;
    const int PQ = p.P * p.Q;
    const int RS = p.KH * p.KW;
    const int taps = cl.nrh * cl.nrw;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rdy = make_rsrc(p.x, p.x_bytes);

    // ---- B operand (dy) ----
    constexpr int BKSTEP = NT / BN > 0 ? NT / BN : 1;
    const int bcol = tid % BN;
    const int bk0 = __builtin_amdgcn_readfirstlane(tid / BN);
    constexpr int BV = BN / 4;
    constexpr int BVSTEP = NT / BV;
    constexpr int BVCNT = (BV * BK + NT - 1) / NT;
    const int vcol = tid % BV, vrow0 = tid / BV;
    bool bvalid;
    int hb = 0, wb = 0, imgb = 0;
    unsigned bvoff = OOB;
    if (MODE == 2) {
        const int n = n0 + 4 * vcol;
        bvalid = n < cl.Ngc && vrow0 < BK;
        if (bvalid) {
            const int img = fdiv(n, cl.d_hw);
            bvoff = (unsigned)((((int64_t)img * p.K + vrow0) * PQ + (n - img * PQ)) * 4);
        }
    } else {
        const int n = n0 + bcol;
        bvalid = n < cl.Ngc;
        if (bvalid) {
            const int img = fdiv(n, cl.d_hw);
            const int rem = n - img * cl.Hc * cl.Wc;
            const int hc = fdiv(rem, cl.d_w);
            const int wc = rem - hc * cl.Wc;
            hb = (ah + p.SH * hc + p.PH - cl.r0) / p.SH;
            wb = (aw + p.SW * wc + p.PW - cl.s0) / p.SW;
            imgb = img * p.K * PQ;
        }
    }

    // ---- A operand (weights), GEMM row m = input channel c ----
    constexpr int AKSTEP = NT / BM > 0 ? NT / BM : 1;
    constexpr int ACNT0 = (BM * BK / NT) < 1 ? 1 : (BM * BK / NT);
    const int acol = tid % BM, ak0 = tid / BM;
    constexpr int AV = BM / 4;
    constexpr int AVSTEP = NT / AV;
    constexpr int AVCNT = (AV * BK + NT - 1) / NT;
    const int avcol = tid % AV, avrow0 = tid / AV;
    // MODE 1/2: byte offset of (row k' = avrow0, m) inside one tap block of the [K][RS][C] tensor, or OOB
    const unsigned avoff = (MODE != 0 && m0 + 4 * avcol < p.M && avrow0 < BK)
                               ? (unsigned)(((int64_t)avrow0 * RS * p.C + m0 + 4 * avcol) * 4) : OOB;

    float ra[MODE == 0 ? ACNT0 : 1];
    float4 rav[MODE == 0 ? 1 : AVCNT];
    float rb[MODE == 2 ? 1 : T::BCNT];
    float4 rbv[MODE == 2 ? BVCNT : 1];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        const bool ktail = kbase + BK > cl.Kgc;
        if (MODE == 0) {
            const int am = m0 + acol;
#pragma unroll
            for (int i = 0; i < ACNT0; ++i) {
                const int k = kbase + ak0 + i * AKSTEP;
                const int ko = fdiv(k, cl.d_taps);
                const int t = k - ko * taps;
                const int j = fdiv(t, cl.d_nrw);
                const int jj = t - j * cl.nrw;
                const int r = cl.r0 + p.SH * j, s = cl.s0 + p.SW * jj;
                const bool ok = am < p.M && k < cl.Kgc;
                ra[i] = bload(rw, ok ? (unsigned)((((int64_t)ko * p.C + am) * RS + r * p.KW + s) * 4) : OOB);
            }
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i) {
                const int k = kbase + bk0 + i * BKSTEP;          // wave-uniform -> scalar unit
                const int ko = fdiv(k, cl.d_taps);
                const int t = k - ko * taps;
                const int j = fdiv(t, cl.d_nrw);
                const int jj = t - j * cl.nrw;
                const int pp = hb - j, qq = wb - jj;
                const bool ok = bvalid && k < cl.Kgc && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
                rb[i] = bload(rdy, ok ? (unsigned)(imgb + ko * PQ + pp * p.Q + qq) * 4u : OOB);
            }
            return;
        }
        // tap-major order: the whole tile shares one filter tap (scalar decode)
        const int tap = fdiv(kbase, p.d_k);
        const int ko0 = kbase - tap * p.K;
        const int j = fdiv(tap, cl.d_nrw);
        const int jj = tap - j * cl.nrw;
        const int rs = (cl.r0 + p.SH * j) * p.KW + cl.s0 + p.SW * jj;
        {
            const unsigned tbase = (unsigned)(((int64_t)ko0 * RS + rs) * p.C * 4);
            const unsigned kstride = (unsigned)(AVSTEP * RS * p.C) * 4u;
#pragma unroll
            for (int i = 0; i < AVCNT; ++i) {
                unsigned o = avoff + tbase + (unsigned)i * kstride;
                if (ktail && kbase + avrow0 + i * AVSTEP >= cl.Kgc) o = OOB;
                rav[i] = bload4(rw, o);
            }
        }
        if (MODE == 2) {
            const unsigned kstride = (unsigned)PQ * 4u;
#pragma unroll
            for (int i = 0; i < BVCNT; ++i) {
                unsigned o = bvoff + (unsigned)(kbase + i * BVSTEP) * kstride;
                if (ktail && kbase + vrow0 + i * BVSTEP >= cl.Kgc) o = OOB;
                rbv[i] = bload4(rdy, o);
            }
        } else {
            const int pp = hb - j, qq = wb - jj;
            const bool ok = bvalid && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
            const unsigned o0 = ok ? (unsigned)(imgb + (ko0 + bk0) * PQ + pp * p.Q + qq) * 4u : OOB;
            const unsigned kstride = (unsigned)(BKSTEP * PQ) * 4u;
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i) rb[i] = bload(rdy, o0 + (unsigned)i * kstride);
        }
    };
    auto store_tile = [&](int buf, int q) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < ACNT0; ++i)
                if (in_quarter(i, ACNT0, q)) As[buf][ak0 + i * AKSTEP][acol] = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < AVCNT; ++i) {
                const int kk = avrow0 + i * AVSTEP;
                if (kk < BK && in_quarter(i, AVCNT, q)) *reinterpret_cast<float4*>(&As[buf][kk][4 * avcol]) = rav[i];
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < BVCNT; ++i) {
                const int kk = vrow0 + i * BVSTEP;
                if (kk < BK && in_quarter(i, BVCNT, q)) *reinterpret_cast<float4*>(&Bs[buf][kk][4 * vcol]) = rbv[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < T::BCNT; ++i)
                if (in_quarter(i, T::BCNT, q)) Bs[buf][bk0 + i * BKSTEP][bcol] = rb[i];
        }
    };

    const int nk = (cl.Kgc + BK - 1) / BK;
    const int kt_begin = split * cl.ktps;
    int kt_end = kt_begin + cl.ktps;
    if (kt_end > nk) kt_end = nk;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0, -1);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int q) {
            if (has_next) store_tile(cur ^ 1, q);
        });
        __syncthreads();
        cur ^= 1;
    }

    if (p.SH == 1 && p.SW == 1) {       // one class: output pixels are contiguous, shared epilogue (+ split-K)
        store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, cl.Ngc, p.H * p.W, cl.d_hw, split, (cl.poff + nt) * WN + wn);
        return;
    }
    if (p.partial) {                    // strided split-K: raw accumulators to partial[split][m][coff + n] (conv_splitk_finish_strided_kernel)
        store_tile_partial_cols<T>(p, acc, m0, n0, wm, wn, lane, cl.Ngc, cl.coff, dp.ng_total, split);
        return;
    }
    // strided classes: pixel (hc, wc) of the class lands on (ah + SH*hc, aw + SW*wc); same fused epilogue
    const int l32 = lane & 31, kh = lane >> 5;
    const int HW = p.H * p.W;
    const int mrow0 = m0 + wm * T::WTM + 4 * kh;
    unsigned ob[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        ob[j] = OOB;
        if (nn < cl.Ngc) {
            const int im = fdiv(nn, cl.d_hw);
            const int rem = nn - im * cl.Hc * cl.Wc;
            const int hc = fdiv(rem, cl.d_w);
            const int wc = rem - hc * cl.Wc;
            const int h = ah + p.SH * hc, w = aw + p.SW * wc;
            ob[j] = (unsigned)((((int64_t)im * p.C + mrow0) * HW + h * p.W + w) * 4);
        }
    }
    store_tile_epilogue_any<T>(p, acc, ob, (unsigned)HW * 4u, mrow0, (cl.poff + nt) * WN + wn);
}

// ---------------------------------------------------------------------------------------------
// 1x1 / stride 1 / pad 0 data gradient with LDS-DMA staging.  Both operands are k-major in memory exactly as the LDS tile wants
// them: the filter tile [16 ko][BM c] is 16 rows of W[K][C], the gradient tile [16 ko][128 pixels] 16 channel rows of dy — so
// `buffer_load_dwordx4 ... lds` moves them global -> LDS with no staging registers, no ds_write and no VALU (lane l of wave w lands at
// (w * 64 + l) * 16 bytes of a 4 KiB pass = row pass*R + (w*64 + l) / (ROWS/4), 16-byte column (w*64 + l) % (ROWS/4): lane-linear).
// Ring of NB = 3 LDS tiles: two k-tiles in flight per workgroup behind the one being multiplied, counted `s_waitcnt vmcnt`, ONE raw
// s_barrier per k-tile (a __syncthreads would drain the DMAs).  The unpadded k-major rows keep the fragment reads conflict-free
// (lane = row, consecutive floats).  Same epilogue, same split-K as conv_dgrad_kernel<MODE 2>, whose launches it replaces when the
// planner picks a 128-pixel tile.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;

template <int BM>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(RG_WAVES))) void conv1x1_dma_kernel(const DgradP dp) {
    using T = Tile<BM, 128, 2, 2>;
    constexpr int NB = 3;
    constexpr int ATILE = BK * BM * 4, BTILE = BK * 128 * 4;          // bytes
    constexpr int APASS = ATILE / 4096, BPASS = BTILE / 4096;         // 4 KiB passes (256 lanes x 16 B) per tile
    constexpr int LPT = APASS + BPASS;                                // DMA instructions per thread and k-tile
    static_assert(APASS >= 1 && ATILE % 4096 == 0, "filter tile is a whole number of DMA passes");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NB][ATILE + BTILE];
    const ConvP& p = dp.c;
    const DgradClass& cl = dp.cls[0];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, kh = lane >> 5;
    const int nwg = p.m_tiles * cl.ntiles;
    if ((int)blockIdx.x >= nwg) return;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * 128;
    const int split = blockIdx.y;
    const int PQ = p.P * p.Q;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rdy = make_rsrc(p.x, p.x_bytes);

    // ---- DMA source offsets of this lane (k row 0 of the pass) ----
    constexpr int ALANES = BM / 4, AROWS = 256 / ALANES;              // lanes per filter row, filter rows per pass
    const int acol = (tid % ALANES) * 4, arow = tid / ALANES;
    const unsigned aoff = (m0 + acol < p.M) ? (unsigned)((arow * p.C + m0 + acol) * 4) : OOB;      // W[ko][c]: row stride C
    const int bcol = (tid & 31) * 4, brow = tid >> 5;                 // 32 lanes per 128-pixel row, 8 rows per pass
    unsigned boff = OOB;
    {
        const int n = n0 + bcol;
        if (n < cl.Ngc) {
            const int img = fdiv(n, cl.d_hw);
            boff = (unsigned)((((int64_t)img * p.K + brow) * PQ + (n - img * PQ)) * 4);
        }
    }
    const unsigned lds_lane0 = (unsigned)__builtin_amdgcn_readfirstlane(wid) * 1024u;

    auto dma_tile = [&](int kt, int buf) {
        const int kbase = kt * BK;
        unsigned char* base = lds[buf] + lds_lane0;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int k = kbase + arow + i * AROWS;
            const unsigned o = (aoff != OOB && k < p.K) ? aoff + (unsigned)((kbase + i * AROWS) * p.C) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(base + i * 4096), 16, (int)o, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
            const int k = kbase + brow + i * 8;
            const unsigned o = (boff != OOB && k < p.K) ? boff + (unsigned)((kbase + i * 8) * PQ) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t*)(base + ATILE + i * 4096), 16, (int)o, 0, 0, 0);
        }
    };

    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);
    const int nk = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk) kt_end = nk;
    const int nkt = kt_end > kt_begin ? kt_end - kt_begin : 0;
#pragma unroll
    for (int sidx = 0; sidx < NB - 1; ++sidx)
        if (sidx < nkt) dma_tile(kt_begin + sidx, sidx);
    int buf = 0;
    for (int it = 0; it < nkt; ++it) {
        // tile `it` has landed once at most the younger tile's DMAs (issued after it) are outstanding
        if (it + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's part of tile `it` is in LDS; every wave is done reading tile it-1
        if (it + NB - 1 < nkt) {
            int nbuf = buf + NB - 1;
            if (nbuf >= NB) nbuf -= NB;
            dma_tile(kt_begin + it + NB - 1, nbuf);          // into the buffer tile it-1 used
        }
        const float* As = reinterpret_cast<const float*>(lds[buf]);
        const float* Bs = reinterpret_cast<const float*>(lds[buf] + ATILE);
        mma_kstep<T::TM, T::TN>([&](int i, int q) { return As[(8 * kh + q) * BM + wm * T::WTM + i * 32 + l32]; },
                                [&](int j, int q) { return Bs[(8 * kh + q) * 128 + wn * T::WTN + j * 32 + l32]; }, acc);
        if (++buf == NB) buf = 0;
    }
    store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, cl.Ngc, p.H * p.W, cl.d_hw, split, (cl.poff + nt) * 2 + wn);
}

// Data gradient for layers with <= 4 input channels (the RGB stem, FD/reid/models/resnet.py via torchvision conv1;
// the generator's 64 -> 3 output ConvTranspose, FD/fdgan/networks.py:133-138).  A 32-row MFMA tile would be > 87 %
// padding there, so this is a direct VALU kernel: one thread per input pixel of one stride-parity class (uniform
// tap set per block), all C channels in registers, the filter bank [K][KH*KW][4] staged once in LDS (broadcast
// float4 reads), dy read coalesced along the row.
template <int CMAX>
__global__ __launch_bounds__(256) void conv_dgrad_smallc_kernel(const DgradP dp) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [K][RS][CMAX]
    const ConvP& p = dp.c;
    const int ci = blockIdx.z;
    const DgradClass& cl = dp.cls[ci];
    const int ah = ci / p.SW, aw = ci % p.SW;
    const int RS = p.KH * p.KW, PQ = p.P * p.Q, HW = p.H * p.W;
    for (int i = threadIdx.x; i < p.K * RS * CMAX; i += blockDim.x) {
        const int c = i % CMAX, t = i / CMAX;          // t = ko*RS + rs
        const int ko = t / RS, rs = t - ko * RS;
        wl[i] = c < p.C ? p.w[((int64_t)ko * p.C + c) * RS + rs] : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= cl.Ngc) return;
    const int img = fdiv(n, cl.d_hw);
    const int rem = n - img * cl.Hc * cl.Wc;
    const int hc = fdiv(rem, cl.d_w);
    const int wc = rem - hc * cl.Wc;
    const int h = ah + p.SH * hc, w = aw + p.SW * wc;
    const int hb = (h + p.PH - cl.r0) / p.SH, wb = (w + p.PW - cl.s0) / p.SW;
    const float* dyb = p.x + (int64_t)img * p.K * PQ;
    float acc[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
    for (int j = 0; j < cl.nrh; ++j) {
        const int pp = hb - j;
        if ((unsigned)pp >= (unsigned)p.P) continue;
        for (int jj = 0; jj < cl.nrw; ++jj) {
            const int qq = wb - jj;
            if ((unsigned)qq >= (unsigned)p.Q) continue;
            const int rs = (cl.r0 + p.SH * j) * p.KW + cl.s0 + p.SW * jj;
            const float* src = dyb + pp * p.Q + qq;
            const float* wrow = wl + rs * CMAX;
#pragma unroll 4
            for (int ko = 0; ko < p.K; ++ko) {
                const float v = src[(int64_t)ko * PQ];
                const float4 wv = *reinterpret_cast<const float4*>(wrow + (int64_t)ko * RS * CMAX);
                acc[0] += v * wv.x;
                if (CMAX > 1) acc[1] += v * wv.y;
                if (CMAX > 2) acc[2] += v * wv.z;
                if (CMAX > 3) acc[3] += v * wv.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        if (c < p.C) {
            float v = acc[c];
            if (p.ep.scale) v *= p.ep.scale[c];
            if (p.ep.shift) v += p.ep.shift[c];
            const int64_t o = ((int64_t)img * p.C + c) * HW + h * p.W + w;
            if (p.ep.res) v += p.ep.res[o];
            v = rg_apply_act(v, p.ep.act, p.ep.slope);
            if (p.ep.mask && !(p.ep.mask[o] > 0.f)) v = 0.f;
            p.y[o] = v;
        }
    }
}

// Register-tiled form of the small-C data gradient for <= 4 taps per axis and class (7x7 / 2, 4x4 / 2, 3x3 / 1):
// lanes run along a class row (coalesced loads of dy), each thread owns PX consecutive class ROWS of one column, so
// one broadcast filter read serves PX pixels and the PX + NRH - 1 gradient rows are loaded once per channel and
// slide across the vertical taps in registers.  The plain kernel above issues one global load and one LDS read per
// 3 FMAs and is bound by the load path.  Branch-free inner loops: rows / columns outside dy are buffer loads with an
// out-of-range offset (-> 0).
template <int PX, int NRH, int NRW>
__device__ __forceinline__ void smallc_px_accumulate(const ConvP& p, const DgradClass& cl, const float* wl, int img,
                                                     int hb0, int wb, float (&acc)[PX][3]) {
    constexpr int WR = PX + NRH - 1;
    const int RS = p.KH * p.KW, PQ = p.P * p.Q;
    const rsrc_t rdy = make_rsrc(p.x, p.x_bytes);
    // pixel i (class row hc0 + i), tap (j, jj) reads dy[pp = hb0 + i - j][q = wb - jj]: window row r = i + NRH-1 - j
    const unsigned imgoff = (unsigned)img * (unsigned)p.K * (unsigned)PQ * 4u;
    unsigned off[WR][NRW];
#pragma unroll
    for (int r = 0; r < WR; ++r) {
        const int pp = hb0 - (NRH - 1) + r;
        const unsigned rowoff = (unsigned)pp < (unsigned)p.P ? imgoff + (unsigned)(pp * p.Q) * 4u : OOB;
#pragma unroll
        for (int jj = 0; jj < NRW; ++jj) {
            const int q = wb - jj;
            const unsigned qo = (unsigned)q < (unsigned)p.Q ? (unsigned)q * 4u : OOB;
            off[r][jj] = ((rowoff | qo) & OOB) ? OOB : rowoff + qo;
        }
    }
    const float* wbase = wl + (cl.r0 * p.KW + cl.s0) * 4;
#pragma unroll 2
    for (int ko = 0; ko < p.K; ++ko) {
        float v[WR][NRW];
        const unsigned koff = (unsigned)ko * (unsigned)PQ * 4u;          // an out-of-range offset stays out of range
#pragma unroll
        for (int r = 0; r < WR; ++r)
#pragma unroll
            for (int jj = 0; jj < NRW; ++jj) v[r][jj] = bload(rdy, off[r][jj] + koff);
#pragma unroll
        for (int j = 0; j < NRH; ++j)
#pragma unroll
            for (int jj = 0; jj < NRW; ++jj) {
                const float4 wv = *reinterpret_cast<const float4*>(wbase + (ko * RS + p.SH * j * p.KW + p.SW * jj) * 4);
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    acc[i][0] += v[i + NRH - 1 - j][jj] * wv.x;
                    acc[i][1] += v[i + NRH - 1 - j][jj] * wv.y;
                    acc[i][2] += v[i + NRH - 1 - j][jj] * wv.z;
                }
            }
    }
}

template <int PX>
__global__ __launch_bounds__(256) void conv_dgrad_smallc_px_kernel(const DgradP dp) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [K][RS][4]
    const ConvP& p = dp.c;
    // block id = 8*ncls*a + 8*ci + x -> pixel region 8a + x of class ci: the classes of one region read the same rows
    // of dy, so they run back to back on the same XCD (ids are dealt round-robin over the 8 XCDs) and share its L2
    const int ncls = p.SH * p.SW;
    const int ci = (blockIdx.x >> 3) % ncls;
    const int region = (int)(blockIdx.x / (8 * ncls)) * 8 + (blockIdx.x & 7);
    const DgradClass& cl = dp.cls[ci];
    const int ah = ci / p.SW, aw = ci % p.SW;
    const int RS = p.KH * p.KW, HW = p.H * p.W;
    const int Hg = (cl.Hc + PX - 1) / PX;
    if (cl.Hc <= 0 || cl.Wc <= 0 || region * (int)blockDim.x >= p.N * Hg * cl.Wc) return;   // uniform
    for (int i = threadIdx.x; i < p.K * RS * 4; i += blockDim.x) {
        const int c = i & 3, t = i >> 2;               // t = ko*RS + rs
        wl[i] = c < p.C ? p.w[(int64_t)(t / RS) * p.C * RS + c * RS + (t % RS)] : 0.f;
    }
    __syncthreads();
    const int n = region * blockDim.x + threadIdx.x;
    if (n >= p.N * Hg * cl.Wc) return;
    const int img = n / (Hg * cl.Wc);
    const int rem = n - img * Hg * cl.Wc;
    const int hg = rem / cl.Wc;
    const int wc = rem - hg * cl.Wc;
    const int hc0 = hg * PX;
    const int hb0 = (ah + p.SH * hc0 + p.PH - cl.r0) / p.SH;
    const int wb = (aw + p.SW * wc + p.PW - cl.s0) / p.SW;
    float acc[PX][3];
#pragma unroll
    for (int i = 0; i < PX; ++i) acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
#define RG_SMALLC_CASE(NRH_, NRW_) \
    case NRH_ * 8 + NRW_: smallc_px_accumulate<PX, NRH_, NRW_>(p, cl, wl, img, hb0, wb, acc); break
    switch (cl.nrh * 8 + cl.nrw) {                      // uniform per block
        RG_SMALLC_CASE(1, 1); RG_SMALLC_CASE(1, 2); RG_SMALLC_CASE(1, 3); RG_SMALLC_CASE(1, 4);
        RG_SMALLC_CASE(2, 1); RG_SMALLC_CASE(2, 2); RG_SMALLC_CASE(2, 3); RG_SMALLC_CASE(2, 4);
        RG_SMALLC_CASE(3, 1); RG_SMALLC_CASE(3, 2); RG_SMALLC_CASE(3, 3); RG_SMALLC_CASE(3, 4);
        RG_SMALLC_CASE(4, 1); RG_SMALLC_CASE(4, 2); RG_SMALLC_CASE(4, 3); RG_SMALLC_CASE(4, 4);
        default: break;                                 // no tap reaches this class: zeros (+ epilogue)
    }
#undef RG_SMALLC_CASE
    const int w = aw + p.SW * wc;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        if (hc0 + i >= cl.Hc) break;
        const int h = ah + p.SH * (hc0 + i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c < p.C) {
                float v = acc[i][c];
                if (p.ep.scale) v *= p.ep.scale[c];
                if (p.ep.shift) v += p.ep.shift[c];
                const int64_t o = ((int64_t)img * p.C + c) * HW + h * p.W + w;
                if (p.ep.res) v += p.ep.res[o];
                v = rg_apply_act(v, p.ep.act, p.ep.slope);
                if (p.ep.mask && !(p.ep.mask[o] > 0.f)) v = 0.f;
                p.y[o] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One-output-channel convolutions (the PatchGAN heads: FD/fdgan/networks.py:225-226 Conv(512 -> 1, 4, 1, 1),
// CC/dual_gan/models/networks.py:946 ResDiscriminator's final conv): a 32-row MFMA tile would be 97 % padding and the
// layer is a 30 MB read with 0.2 GFLOP, so these are direct VALU kernels bound by the read of x.  Lanes run along the
// output pixels (coalesced rows of x, every element re-used KH*KW times out of L1); the input channels (forward) or
// the pixels (weight gradient) are sliced across blockIdx.y and the slices are summed by the ordinary split-K
// finishing kernels (same partial layout [slice][M = 1][n]), in fixed order: deterministic.
// ---------------------------------------------------------------------------------------------
struct ThinP {
    const float* x;
    const float* a;      // fwd: w [1][C][KH][KW]; wgrad: dy [N][1][P][Q]
    float* partial;
    int N, C, H, W, P, Q, SH, SW, PH, PW, per_slice;
    unsigned x_bytes;
    FastDiv d_pq, d_q;
};

template <int KH, int KW, int KO>
__global__ __launch_bounds__(256) void conv_fwd_k1_kernel(const ThinP t) {
    const int Ng = t.N * t.P * t.Q;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= Ng) return;
    const int img = fdiv(pix, t.d_pq);
    const int pq = pix - img * t.P * t.Q;
    const int pp = fdiv(pq, t.d_q), qq = pq - pp * t.Q;
    const int h0 = pp * t.SH - t.PH, w0 = qq * t.SW - t.PW;
    const rsrc_t rx = make_rsrc(t.x, t.x_bytes);
    unsigned off[KH][KW];
#pragma unroll
    for (int r = 0; r < KH; ++r)
#pragma unroll
        for (int s = 0; s < KW; ++s) {
            const int h = h0 + r, w = w0 + s;
            off[r][s] = ((unsigned)h < (unsigned)t.H && (unsigned)w < (unsigned)t.W)
                            ? (unsigned)((img * t.C * t.H + h) * t.W + w) * 4u : OOB;
        }
    const int c0 = blockIdx.y * t.per_slice;
    const int c1 = min(c0 + t.per_slice, t.C);
    const unsigned cstride = (unsigned)(t.H * t.W) * 4u;
    float acc[KO];
#pragma unroll
    for (int k = 0; k < KO; ++k) acc[k] = 0.f;
#pragma unroll 2
    for (int c = c0; c < c1; ++c) {
        const float* wc = t.a + c * (KH * KW);          // uniform: scalar loads; output channel k at + k*C*KH*KW
        const unsigned co = (unsigned)c * cstride;      // an out-of-range offset stays out of range
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const float xv = bload(rx, off[r][s] + co);
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[k] += xv * wc[k * t.C * (KH * KW) + r * KW + s];
            }
    }
#pragma unroll
    for (int k = 0; k < KO; ++k) t.partial[((int64_t)blockIdx.y * KO + k) * Ng + pix] = acc[k];
}

// grid (C, slices): block (c, s) reduces pixels [s*per_slice, (s+1)*per_slice) for the KH*KW taps of channel c and the KO
// output channels; partial layout [slice][KO][C][KH*KW]
template <int KH, int KW, int KO>
__global__ __launch_bounds__(256) void conv_wgrad_k1_kernel(const ThinP t) {
    constexpr int RS = KH * KW;
    __shared__ float red[4][KO * RS];
    const int Ng = t.N * t.P * t.Q;
    const int PQ = t.P * t.Q;
    const int c = blockIdx.x;
    const int beg = blockIdx.y * t.per_slice;
    const int end = min(beg + t.per_slice, Ng);
    const rsrc_t rx = make_rsrc(t.x, t.x_bytes);
    float acc[KO][KH][KW];
#pragma unroll
    for (int k = 0; k < KO; ++k)
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) acc[k][r][s] = 0.f;
    for (int pix = beg + threadIdx.x; pix < end; pix += 256) {
        const int img = fdiv(pix, t.d_pq);
        const int pq = pix - img * PQ;
        const int pp = fdiv(pq, t.d_q), qq = pq - pp * t.Q;
        const int h0 = pp * t.SH - t.PH, w0 = qq * t.SW - t.PW;
        float g[KO];
#pragma unroll
        for (int k = 0; k < KO; ++k) g[k] = t.a[(img * KO + k) * PQ + pq];
        const int base = (img * t.C + c) * t.H;
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const int h = h0 + r, w = w0 + s;
                const bool ok = (unsigned)h < (unsigned)t.H && (unsigned)w < (unsigned)t.W;
                const float xv = bload(rx, ok ? (unsigned)((base + h) * t.W + w) * 4u : OOB);
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[k][r][s] += g[k] * xv;
            }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KO; ++k)
#pragma unroll
        for (int r = 0; r < KH; ++r)
#pragma unroll
            for (int s = 0; s < KW; ++s) {
                const float v = rg_wave_sum(acc[k][r][s]);
                if (lane == 0) red[wid][(k * KH + r) * KW + s] = v;
            }
    __syncthreads();
    if (threadIdx.x < KO * RS) {
        const int k = threadIdx.x / RS, tap = threadIdx.x - k * RS;
        t.partial[(((int64_t)blockIdx.y * KO + k) * t.C + c) * RS + tap] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// Weight gradient of 3x3 / stride 1 / pad <= 1 layers with Q % 4 == 0 (the Output blocks: 64 -> 3 on a full-resolution map): four
// output pixels of a row per thread.  Their six input columns per filter row are one 16-byte load plus the two neighbours — 9 load
// instructions for four pixels instead of 36 (206 -> 164 us at 128 x 64 x 128 x 64; the same idea made the forward kernel slower).
__device__ __forceinline__ void thin_px4_offsets(const ThinP& t, int pix, unsigned (&ol)[3], unsigned (&om)[3], unsigned (&orr)[3],
                                                 int& img, int& pq) {
    const int PQ = t.P * t.Q;
    img = fdiv(pix, t.d_pq);
    pq = pix - img * PQ;
    const int pp = fdiv(pq, t.d_q), qq = pq - pp * t.Q;
    const int h0 = pp - t.PH, w0 = qq - t.PW;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int h = h0 + r;
        const bool hok = (unsigned)h < (unsigned)t.H;
        const unsigned row = (unsigned)((img * t.C * t.H + h) * t.W) * 4u;      // channel 0; + c * H * W * 4 per channel
        ol[r] = (hok && w0 >= 0) ? row + (unsigned)w0 * 4u : OOB;
        om[r] = hok ? row + (unsigned)(w0 + 1) * 4u : OOB;                     // columns w0 + 1 .. w0 + 4: inside the row (host check)
        orr[r] = (hok && w0 + 5 < t.W) ? row + (unsigned)(w0 + 5) * 4u : OOB;
    }
}

template <int KO>
__global__ __launch_bounds__(256) void conv_wgrad_k1_px4_kernel(const ThinP t) {
    __shared__ float red[4][KO * 9];
    const int Ng = t.N * t.P * t.Q;
    const int PQ = t.P * t.Q;
    const int c = blockIdx.x;
    const int beg = blockIdx.y * t.per_slice;                      // per_slice % 4 == 0 (host)
    const int end = min(beg + t.per_slice, Ng);
    const rsrc_t rx = make_rsrc(t.x, t.x_bytes);
    const unsigned co = (unsigned)c * (unsigned)(t.H * t.W) * 4u;
    float acc[KO][3][3];
#pragma unroll
    for (int k = 0; k < KO; ++k)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[k][r][s] = 0.f;
    for (int pix = beg + threadIdx.x * 4; pix < end; pix += 1024) {
        unsigned ol[3], om[3], orr[3];
        int img, pq;
        thin_px4_offsets(t, pix, ol, om, orr, img, pq);
        float g[KO][4];
#pragma unroll
        for (int k = 0; k < KO; ++k) {
            const float4 gv = *reinterpret_cast<const float4*>(t.a + (int64_t)(img * KO + k) * PQ + pq);
            g[k][0] = gv.x; g[k][1] = gv.y; g[k][2] = gv.z; g[k][3] = gv.w;
        }
        float v[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            v[r][0] = bload(rx, ol[r] + co);
            const float4 m = bload4(rx, om[r] + co);
            v[r][1] = m.x; v[r][2] = m.y; v[r][3] = m.z; v[r][4] = m.w;
            v[r][5] = bload(rx, orr[r] + co);
        }
#pragma unroll
        for (int k = 0; k < KO; ++k)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[k][r][s] += g[k][j] * v[r][j + s];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KO; ++k)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float v = rg_wave_sum(acc[k][r][s]);
                if (lane == 0) red[wid][(k * 3 + r) * 3 + s] = v;
            }
    __syncthreads();
    if (threadIdx.x < KO * 9) {
        const int k = threadIdx.x / 9, tap = threadIdx.x - k * 9;
        t.partial[(((int64_t)blockIdx.y * KO + k) * t.C + c) * 9 + tap] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// out[(img*M + m)*PIX + pix] = act((sum_s partial[s][m][n]) * scale[m] + shift[m] + res), n = img*PIX + pix
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ out, int M, int Ng, int PIX,
                                                                 FastDiv d_pix, int splits, Epilogue ep) {
    const int64_t total = (int64_t)M * Ng;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / Ng);
        const int n = (int)(i - (int64_t)m * Ng);
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += partial[(int64_t)s * total + i];
        const int im = fdiv(n, d_pix);
        const int pix = n - im * PIX;
        const int64_t o = ((int64_t)im * M + m) * PIX + pix;
        if (ep.scale) v *= ep.scale[m];
        if (ep.shift) v += ep.shift[m];
        if (ep.res) v += ep.res[o];
        v = rg_apply_act(v, ep.act, ep.slope);
        if (ep.mask && !(ep.mask[o] > 0.f)) v = 0.f;
        out[o] = v;
    }
}

// The same for Ng % 4 == 0 and PIX % 4 == 0 (every layer of the networks here): four consecutive columns per thread — they stay in
// one image and one row, so partials, residual, mask and output move as float4 — and the partial loads of four splits are in flight
// together.  Same left-to-right sum over the splits per element: same values as the scalar kernel.
__global__ __launch_bounds__(256) void conv_splitk_finish_vec_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                     int M, int Ng, int PIX, FastDiv d_pix, FastDiv d_ng4,
                                                                     int splits, Epilogue ep) {
    const int ng4 = Ng >> 2;
    const int64_t total4 = (int64_t)M * ng4;
    const int64_t sstride4 = total4;                    // float4 units between splits
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = fdiv((int)i, d_ng4);
        const int n = ((int)i - m * ng4) << 2;
        const float4 v = splitk_sum4(p4, sstride4, i, splits);
        const int im = fdiv(n, d_pix);
        const int pix = n - im * PIX;
        splitk_epilogue_store4(out, ((int64_t)im * M + m) * PIX + pix, m, v, ep);
    }
}

// Strided data gradient with split-K: partial[s][m][col] holds the classes' columns side by side; column col of class ci is pixel
// (img, hc, wc) of that stride-parity class = output pixel (ah + SH hc, aw + SW wc).  One thread per OUTPUT element (m, img, h, w):
// its class and column follow from (h, w), the stores are coalesced along w and the partial loads are SW interleaved unit-stride
// streams (one per column parity).  Same epilogue as the one-class finisher.
__global__ __launch_bounds__(256) void conv_splitk_finish_strided_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                         const DgradP dp, int splits, FastDiv d_hw, FastDiv d_w,
                                                                         FastDiv d_nhw) {
    const ConvP& p = dp.c;
    const int ncols = dp.ng_total;
    const int HW = p.H * p.W, NHW = p.N * HW;
    const int64_t total = (int64_t)p.M * NHW, slab = (int64_t)p.M * ncols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = fdiv((int)i, d_nhw);
        const int r = (int)i - m * NHW;
        const int im = fdiv(r, d_hw);
        const int hw = r - im * HW;
        const int h = fdiv(hw, d_w), w = hw - h * p.W;
        const int ah = h % p.SH, aw = w % p.SW;               // SH, SW <= 2
        const DgradClass& cl = dp.cls[ah * p.SW + aw];
        const int col = cl.coff + (im * cl.Hc + h / p.SH) * cl.Wc + w / p.SW;
        const float* pp = partial + (int64_t)m * ncols + col;
        float v = 0.f;
        if (cl.Kgc > 0)                                       // classes without filter taps (1x1 / stride 2: three of four) hold no partials
            for (int s = 0; s < splits; ++s) v += pp[(int64_t)s * slab];
        const int64_t o = ((int64_t)im * p.M + m) * HW + hw;
        if (p.ep.scale) v *= p.ep.scale[m];
        if (p.ep.shift) v += p.ep.shift[m];
        if (p.ep.res) v += p.ep.res[o];
        v = rg_apply_act(v, p.ep.act, p.ep.slope);
        if (p.ep.mask && !(p.ep.mask[o] > 0.f)) v = 0.f;
        out[o] = v;
    }
}

// w[K][C][RS] -> wt[K][RS][C]
__global__ void weights_to_krsc_kernel(const float* __restrict__ w, float* __restrict__ wt, int64_t total, int C, int RS) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int rs = (int)(t % RS);
        const int64_t k = t / RS;
        wt[i] = w[(k * C + c) * RS + rs];
    }
}

// The re-layout of EVERY filter of a network in one launch (the per-filter launches were 57 per step of the joint trainer, 5 us
// each, after every optimizer step).  Table in device memory, 6 int64 words per filter: w, wt, K, C, RS, first block; KRSC_CHUNK
// elements per workgroup.
constexpr int KRSC_CHUNK = 2048;
__global__ __launch_bounds__(256) void weights_to_krsc_multi_kernel(const long long* __restrict__ tab, int count) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[(int64_t)mid * 6 + 5] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long long* e = tab + (int64_t)lo * 6;
    const float* w = reinterpret_cast<const float*>(e[0]);
    float* wt = reinterpret_cast<float*>(e[1]);
    const int C = (int)e[3], RS = (int)e[4];
    const int64_t total = e[2] * C * RS;
    const int64_t beg = ((long long)blockIdx.x - e[5]) * KRSC_CHUNK;
    for (int64_t i = beg + threadIdx.x; i < beg + KRSC_CHUNK && i < total; i += 256) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int rs = (int)(t % RS);
        const int64_t k = t / RS;
        wt[i] = w[(k * C + c) * RS + rs];
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dw[K][C*KH*KW] = dy[K][N*P*Q] x im2col(x)^T; the reduction (output pixels) is split across
// blockIdx.z, partial tiles go to a workspace and a second kernel sums them.  Lanes run along the reduction axis
// (16 consecutive output pixels: coalesced rows of dy and x); each thread owns fixed GEMM rows / columns.
// Column order: (c, r, s) as in the checkpoint layout, or — p.a_vec4 != 0, C % 16 == 0 — (r, s)-major n' = rs*C + c,
// which lets a whole 16..128-column tile share one filter tap (one padding test per k-tile instead of one per
// element); the finishing kernel then writes dw back in [K][C][KH][KW] order.
// ---------------------------------------------------------------------------------------------
// VEC: 1x1 / stride 1 / pad 0 with P*Q % 4 == 0 — both operands are [rows][pixels] with the reduction axis
// contiguous, so each lane loads 4 consecutive pixels of one row (float4) instead of 4 scalar loads.
// VECA: only the dy operand that way (any filter, P*Q % 4 == 0); the im2col operand keeps the scalar gather.
template <int BM, int BN, int WM, int WN, bool VEC, bool VECA>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    __shared__ float As[2][BK][T::LDA];
    __shared__ float Bs[2][BK][T::LDB];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // Workgroups of one split read the same slice of dy and x (every tile row shares dy rows, every tile column x columns),
    // workgroups of different splits share nothing: with a multiple of 8 splits, split s lives entirely on XCD s % 8 (block
    // ids are dealt round-robin over the XCDs in x-then-z order), so each slice is pulled into ONE L2 instead of all eight
    // (measured before: 4.8x the algorithmic bytes fetched).  Otherwise: the tile remap inside each split.
    int tile_id, split;
    if ((gridDim.z & 7) == 0) {
        const unsigned lin = blockIdx.z * gridDim.x + blockIdx.x;
        const unsigned xcd = lin & 7, idx = lin >> 3;
        split = (int)(xcd + 8 * (idx / gridDim.x));
        tile_id = (int)(idx % gridDim.x);
    } else {
        split = blockIdx.z;
        tile_id = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int mt = tile_id % p.m_tiles, nt = tile_id / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const rsrc_t rdy = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);

    const int kk = tid & 15, r0 = tid >> 4;  // lanes run along the reduction (pixel) axis
    constexpr int ACNT = BM / 16, BCNT = BN / 16;
    const int PQ = p.P * p.Q, HW = p.H * p.W, RS = p.KH * p.KW;
    const bool rsc = p.a_vec4 != 0;
    // no padding and every window inside the image: no bounds test at all (all 1x1 layers)
    const bool nopad = p.PH == 0 && p.PW == 0 && (p.P - 1) * p.SH + p.KH <= p.H && (p.Q - 1) * p.SW + p.KW <= p.W;
    // (r,s)-major and the whole column tile inside one tap: one bounds test per lane per k-tile
    const bool same_rs = rsc && (fdiv(n0, p.d_c) == fdiv(min(n0 + BN, p.Ng) - 1, p.d_c));

    // per-thread GEMM rows (dy channels) and columns (c, r, s): fixed for the whole reduction
    unsigned aoff[ACNT];          // byte offset of row m inside one image of dy, or OOB
#pragma unroll
    for (int i = 0; i < ACNT; ++i) {
        const int m = m0 + r0 + 16 * i;
        aoff[i] = m < p.M ? (unsigned)m * (unsigned)PQ * 4u : OOB;
    }
    int coff[BCNT], crs[BCNT];    // element offset c*HW + r*W + s, packed (r,s) or -1
#pragma unroll
    for (int i = 0; i < BCNT; ++i) {
        const int n = n0 + r0 + 16 * i;
        if (n < p.Ng) {
            int c, rs;
            if (rsc) {
                rs = fdiv(n, p.d_c);
                c = n - rs * p.C;
            } else {
                c = fdiv(n, p.d_rs);
                rs = n - c * RS;
            }
            const int r = fdiv(rs, p.d_kw);
            const int s = rs - r * p.KW;
            coff[i] = c * HW + r * p.W + s;
            crs[i] = (r << 16) | s;
        } else {
            coff[i] = 0;
            crs[i] = -1;
        }
    }

    float ra[ACNT < 4 ? 4 : ACNT], rb[BCNT < 4 ? 4 : BCNT];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    // VEC: thread v owns row (v >> 2) + 64*i and the pixel quad (v & 3)*4 of every k-tile
    constexpr int AVN = (BM * 4 + NT - 1) / NT, BVN = (BN * 4 + NT - 1) / NT;
    const int vrow = tid >> 2, vkq = (tid & 3) * 4;
    unsigned avoff[AVN], bvoff[BVN];
#pragma unroll
    for (int i = 0; i < AVN; ++i) {
        const int m = m0 + vrow + 64 * i;
        avoff[i] = (vrow + 64 * i < BM && m < p.M) ? (unsigned)m * (unsigned)PQ * 4u : OOB;
    }
    int vrr[BVN], vss[BVN];               // p.wshift: tap offset (r - PH, s - PW) of the row's column n = (c, r, s)
#pragma unroll
    for (int i = 0; i < BVN; ++i) {
        const int n = n0 + vrow + 64 * i;
        int c = n;
        vrr[i] = vss[i] = 0;
        if (p.wshift) {
            c = fdiv(n, p.d_rs);
            const int rs = n - c * RS;
            const int r = fdiv(rs, p.d_kw);
            vrr[i] = r - p.PH;
            vss[i] = rs - r * p.KW - p.PW;
        }
        bvoff[i] = (vrow + 64 * i < BN && n < p.Ng) ? (unsigned)c * (unsigned)HW * 4u : OOB;
    }

    auto load_tile = [&](int kt) {
        if (VEC || VECA) {
            const int g = kt * BK + vkq;
            const bool gvalid = g < p.Kg;
            const int img = gvalid ? fdiv(g, p.d_pq) : 0;
            const int pq = g - img * PQ;
            const unsigned ab = gvalid ? (unsigned)(img * p.K * PQ + pq) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < AVN; ++i) {
                const float4 t = bload4(rdy, ((ab | avoff[i]) & OOB) ? OOB : ab + avoff[i]);
                ra[4 * i + 0] = t.x; ra[4 * i + 1] = t.y; ra[4 * i + 2] = t.z; ra[4 * i + 3] = t.w;
            }
            if (VEC && p.wshift) {
                // stride-1 filter tap (r, s): the four output pixels (pp, q0 .. q0 + 3) read x at (pp + r - PH, q0 + s - PW ..), four
                // CONSECUTIVE floats (Q % 4 == 0 keeps a quad inside one row).  One column may fall off either end of the row
                // (|s - PW| <= 1): the load is moved one element inwards and the vector shifted, so every address stays inside
                // the row (dword-aligned dwordx4 buffer loads are legal; nothing relies on partial out-of-range returns)
                const int pp = fdiv(pq, p.d_q);
                const int q0 = pq - pp * p.Q;
#pragma unroll
                for (int i = 0; i < BVN; ++i) {
                    const int hh = pp + vrr[i], wb = q0 + vss[i];
                    const bool ok = gvalid && bvoff[i] != OOB && (unsigned)hh < (unsigned)p.H;
                    const bool neg = wb < 0, over = wb + 3 >= p.W;
                    const int e = img * p.C * HW + hh * p.W + wb + (neg ? 1 : 0) - (over ? 1 : 0);
                    const float4 t = bload4(rx, ok ? (unsigned)e * 4u + bvoff[i] : OOB);
                    rb[4 * i + 0] = neg ? 0.f : (over ? t.y : t.x);
                    rb[4 * i + 1] = neg ? t.x : (over ? t.z : t.y);
                    rb[4 * i + 2] = neg ? t.y : (over ? t.w : t.z);
                    rb[4 * i + 3] = neg ? t.z : (over ? 0.f : t.w);
                }
                return;
            }
            if (VEC) {
                const unsigned bb = gvalid ? (unsigned)(img * p.C * HW + pq) * 4u : OOB;
#pragma unroll
                for (int i = 0; i < BVN; ++i) {
                    const float4 t = bload4(rx, ((bb | bvoff[i]) & OOB) ? OOB : bb + bvoff[i]);
                    rb[4 * i + 0] = t.x; rb[4 * i + 1] = t.y; rb[4 * i + 2] = t.z; rb[4 * i + 3] = t.w;
                }
                return;
            }
        }
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
        if (!VECA) {
            const unsigned abase = gvalid ? (unsigned)(img * p.K * PQ + pq) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < ACNT; ++i) ra[i] = bload(rdy, (abase | aoff[i]) & OOB ? OOB : abase + aoff[i]);
        }
        const int xb = img * p.C * HW + h0 * p.W + w0;      // element index of (img, 0, h0, w0); may sit in the padding
        if (nopad) {
#pragma unroll
            for (int i = 0; i < BCNT; ++i)
                rb[i] = bload(rx, (gvalid && crs[i] >= 0) ? (unsigned)(xb + coff[i]) * 4u : OOB);
        } else if (same_rs) {
            const int r = crs[0] >> 16, s = crs[0] & 0xffff;
            const int h = h0 + r, w = w0 + s;
            const bool ok = gvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
#pragma unroll
            for (int i = 0; i < BCNT; ++i) rb[i] = bload(rx, (ok && crs[i] >= 0) ? (unsigned)(xb + coff[i]) * 4u : OOB);
        } else {
#pragma unroll
            for (int i = 0; i < BCNT; ++i) {
                const int r = crs[i] >> 16, s = crs[i] & 0xffff;
                const int h = h0 + r, w = w0 + s;
                const bool ok = gvalid && crs[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                rb[i] = bload(rx, ok ? (unsigned)(xb + coff[i]) * 4u : OOB);
            }
        }
    };
    auto store_tile = [&](int buf, int q) {
        if (VEC || VECA) {
#pragma unroll
            for (int i = 0; i < AVN; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vrow + 64 * i < BM && in_quarter(4 * i + j, 4 * AVN, q)) As[buf][vkq + j][vrow + 64 * i] = ra[4 * i + j];
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < BVN; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (vrow + 64 * i < BN && in_quarter(4 * i + j, 4 * BVN, q)) Bs[buf][vkq + j][vrow + 64 * i] = rb[4 * i + j];
            return;
        }
        if (!VECA) {
#pragma unroll
            for (int i = 0; i < ACNT; ++i)
                if (in_quarter(i, ACNT, q)) As[buf][kk][r0 + 16 * i] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BCNT; ++i)
            if (in_quarter(i, BCNT, q)) Bs[buf][kk][r0 + 16 * i] = rb[i];
    };

    const int nk_total = (p.Kg + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk_total) kt_end = nk_total;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0, -1);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);
        mma_tile<T>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int q) {
            if (has_next) store_tile(cur ^ 1, q);
        });
        __syncthreads();
        cur ^= 1;
    }

    // partial (or final) tile: [split][M][Ng], columns contiguous
    const int l32 = lane & 31, kh = lane >> 5;
    const rsrc_t ro = make_rsrc(p.y, p.y_bytes);
    const int mrow0 = m0 + wm * T::WTM + 4 * kh;
    const unsigned rstride = (unsigned)p.Ng * 4u;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int nn = n0 + wn * T::WTN + j * 32 + l32;
        const unsigned ob = nn < p.Ng ? (unsigned)((((int64_t)split * p.M + mrow0) * p.Ng + nn) * 4) : OOB;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = i * 32 + (r & 3) + 8 * (r >> 2);
                bstore(ro, (mrow0 + mo < p.M) ? ob + (unsigned)mo * rstride : OOB, acc[i][j][r]);
            }
    }
}

// dw[i] = sum_s ws[s][i]; with rsc != 0 the partial columns are (r,s)-major (n' = rs*C + c) and are written back in
// the checkpoint order [K][C][RS].  Deterministic (fixed summation tree).  64 outputs x 4 split lanes per
// workgroup: small filter tensors with hundreds of splits stay parallel instead of one long serial chain per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            int64_t n, int splits, int rsc, int C, int RS) {
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
    const float s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (!rsc) {
        out[i] = s;
        return;
    }
    const int64_t crs = (int64_t)C * RS;
    const int64_t m = i / crs;
    const int np = (int)(i - m * crs);
    const int rs = np / C, c = np - rs * C;
    out[m * crs + (int64_t)c * RS + rs] = s;
}

// n % 4 == 0: 64 float4 outputs x 4 split lanes per workgroup, four slab loads in flight per thread; per element the same summation
// tree as splitk_reduce_kernel (lane ty adds splits ty, ty + 8, ... and ty + 4, ty + 12, ... in two chains): same values.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n,
                                                                int splits, int rsc, int C, int RS) {
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
    const float4 v = make_float4((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                                 (r0.w + r1.w) + (r2.w + r3.w));
    if (!rsc) {
        reinterpret_cast<float4*>(out)[i] = v;
        return;
    }
    const int64_t crs = (int64_t)C * RS;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t e = i * 4 + j;
        const int64_t m = e / crs;
        const int np = (int)(e - m * crs);
        const int rs = np / C, c = np - rs * C;
        out[m * crs + (int64_t)c * RS + rs] = vv[j];
    }
}

// Split-K reduction of the weight gradient of a convolution whose frozen-statistics BatchNorm is folded into it (norm.hip, "conv +
// frozen-statistics BatchNorm"): ONE workgroup per filter row k sums the slabs of its row (float4 columns, the summation tree of
// splitk_reduce_vec_kernel: same G bit for bit), and finishes the fold while G is in registers — dgamma[k] = invstd (sum_m W G - mean
// sum g), dbeta[k] = sum g (from the slice partials), dW = scale[k] G.  Replaces splitk_reduce_vec_kernel + bn_fold_wgrad_kernel:
// one launch, one write and one read of G less per folded layer (106 layers per FD-GAN step).  M % 4 == 0, 16-byte aligned buffers.
__global__ __launch_bounds__(256) void splitk_reduce_fold_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 const float* __restrict__ w, int M, int64_t n, int splits,
                                                                 const float* __restrict__ scale, const float* __restrict__ invstd,
                                                                 const float* __restrict__ mean, const float* __restrict__ sum_g,
                                                                 const float* __restrict__ part, int S, float* __restrict__ dbeta,
                                                                 float* __restrict__ dgamma) {
    __shared__ float4 red[4][64];
    __shared__ float redf[16];
    const int k = blockIdx.x;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int M4 = M >> 2;
    const int64_t n4 = n >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(ws);
    const float4* f4 = reinterpret_cast<const float4*>(w);
    float4* o4 = reinterpret_cast<float4*>(out);
    const float sc = scale[k];
    float dot = 0.f;
    for (int c0 = 0; c0 < M4; c0 += 64) {
        const int c = c0 + tx;
        const int64_t i = (int64_t)k * M4 + c;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (c < M4) {
            int q = ty;
            for (; q + 12 < splits; q += 16) {
                const float4 a = w4[(int64_t)q * n4 + i], b = w4[(int64_t)(q + 4) * n4 + i];
                const float4 cc = w4[(int64_t)(q + 8) * n4 + i], d = w4[(int64_t)(q + 12) * n4 + i];
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
                s0.x += cc.x; s0.y += cc.y; s0.z += cc.z; s0.w += cc.w;
                s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            }
            for (; q + 4 < splits; q += 8) {
                const float4 a = w4[(int64_t)q * n4 + i], b = w4[(int64_t)(q + 4) * n4 + i];
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            }
            if (q < splits) {
                const float4 a = w4[(int64_t)q * n4 + i];
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            }
        }
        red[ty][tx] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
        __syncthreads();
        if (ty == 0 && c < M4) {
            const float4 r0 = red[0][tx], r1 = red[1][tx], r2 = red[2][tx], r3 = red[3][tx];
            const float4 v = make_float4((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                                         (r0.w + r1.w) + (r2.w + r3.w));
            if (dgamma) {
                const float4 wv = f4[i];
                dot += (wv.x * v.x + wv.y * v.y) + (wv.z * v.z + wv.w * v.w);
            }
            o4[i] = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
        }
        __syncthreads();
    }
    float sg = 0.f;
    if (part) {                       // channel sum of g from slice / tile partials (the tree of bn_fold_wgrad_kernel)
        float t = 0.f;
        for (int s = threadIdx.x; s < S; s += 256) t += part[(int64_t)k * S + s];
        sg = rg_block_sum(t, redf);
        if (dbeta && threadIdx.x == 0) dbeta[k] = sg;
    } else if (sum_g) {
        sg = sum_g[k];
    }
    if (dgamma) {
        const float t = rg_block_sum(dot, redf);
        if (threadIdx.x == 0) dgamma[k] = invstd[k] * (t - mean[k] * sg);
    }
}

// algorithmic HBM bytes of one conv launch: one read of each operand + one write of the result (fp32)
#define ALG_BYTES (4.0 * ((double)N * C * H * W + (double)K * C * KH * KW + (double)N * K * P * Q))

static bool fits_buffer(int64_t elems) { return elems > 0 && elems * 4 < (1ll << 31); }

static void fill_common(ConvP& p, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                        int P, int Q) {
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.KH = KH; p.KW = KW;
    p.SH = SH; p.SW = SW; p.PH = PH; p.PW = PW; p.P = P; p.Q = Q;
    p.d_rs = make_fastdiv(KH * KW);
    p.d_kw = make_fastdiv(KW);
    p.d_pq = make_fastdiv(P * Q);
    p.d_q = make_fastdiv(Q);
    p.a_vec4 = 0;
    p.wshift = 0;
    p.ktiles_per_split = 1 << 30;
    p.splits = 1;
    p.partial = nullptr;
    p.arrive = nullptr;
    p.x_bytes = p.w_bytes = p.y_bytes = p.partial_bytes = 0;
    p.d_c = make_fastdiv(C);
    p.d_k = make_fastdiv(K);
}



static int validate(const char* op, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                    int P, int Q) {
    RG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && KH > 0 && KW > 0, "%s: non-positive dimension", op);
    RG_REQUIRE(SH > 0 && SW > 0 && PH >= 0 && PW >= 0 && P > 0 && Q > 0, "%s: bad stride/pad/output size", op);
    RG_REQUIRE((P - 1) * SH - PH + KH - 1 >= 0 && (Q - 1) * SW - PW + KW - 1 >= 0, "%s: inconsistent geometry", op);
    // every output pixel must start inside the padded input
    RG_REQUIRE((int64_t)(P - 1) * SH - PH < H && (int64_t)(Q - 1) * SW - PW < W, "%s: output larger than input allows", op);
    RG_REQUIRE((int64_t)N * P * Q < (1ll << 31) && (int64_t)C * KH * KW < (1ll << 31) && (int64_t)N * H * W < (1ll << 31) &&
                   (int64_t)C * H * W < (1ll << 31) && (int64_t)K * P * Q < (1ll << 31) &&
                   (int64_t)K * KH * KW < (1ll << 31),
               "%s: dimension product exceeds 2^31", op);
    RG_REQUIRE(fits_buffer((int64_t)N * C * H * W) && fits_buffer((int64_t)N * K * P * Q) &&
                   fits_buffer((int64_t)K * C * KH * KW),
               "%s: every tensor must be smaller than 2 GiB (32-bit buffer offsets)", op);
    return RG_OK;
}

static const int kTileBM[4] = {128, 64, 64, 32};
static const int kTileBN[4] = {128, 128, 64, 256};

struct GemmPlan {
    int tile, m_tiles, n_tiles, splits, ktiles_per_split;
};

// Development override: RG_CONV_FORCE="tile,splits" (tile 0..3 or -1, splits >= 1 or -1) pins the choice.
static int g_force_tile = -2, g_force_splits = -2;
static void force_override(int* tile, int* splits) {
    if (g_force_tile == -2) {
        g_force_tile = -1;
        g_force_splits = -1;
        const char* e = getenv("RG_CONV_FORCE");
        if (e) sscanf(e, "%d,%d", &g_force_tile, &g_force_splits);
    }
    if (g_force_tile >= 0 && g_force_tile <= 3) *tile = g_force_tile;
    if (g_force_splits >= 1) *splits = g_force_splits;
}

// Tile + split-K choice for the fwd / dgrad GEMMs (M x Ng outputs, Kg reduction), allow_split = single-class output.
// A small cost model instead of thresholds: 256 CUs, up to 4 co-resident workgroups per CU sharing each SIMD's
// matrix pipe.  Measured pipe utilisation vs co-residency (PMC, round 1): ~0.55 alone, ~0.7 with two, ~0.85 with
// three or more.  Work that does not divide into 256-wide rounds leaves CUs idle (wave quantisation), so the split
// count is chosen to land on full rounds; split-K pays for its partial tiles (write + read in the finishing kernel).
static GemmPlan plan_gemm(int M, int64_t Ng, int64_t Kg, bool allow_split) {
    const int64_t nk = rg::cdiv64(Kg > 0 ? Kg : 1, BK);
    int cand[2], ncand = 0;
    if (M <= 32) cand[ncand++] = 3;
    else if (M <= 64) { if (Ng >= 128) cand[ncand++] = 1; cand[ncand++] = 2; }
    else { if (Ng >= 128) cand[ncand++] = 0; cand[ncand++] = 2; }
    // utilisation of a SIMD's matrix pipe with r co-resident workgroups per CU, relative efficiency of each tile
    // shape (loads + LDS traffic per MFMA), fixed per-workgroup cost (first-tile latency, epilogue) in cycles
    static const double util[5] = {1.0, 0.60, 0.85, 0.90, 0.92};
    static const double tile_eff[4] = {1.0, 0.92, 0.85, 0.85};
    const double fixed = 6000.0;
    double best = 1e300;
    GemmPlan pl;
    pl.tile = cand[0];
    int best_s = 1;
    for (int ci = 0; ci < ncand; ++ci) {
        const int t = cand[ci];
        const int64_t tiles = (int64_t)rg::cdiv(M, kTileBM[t]) * rg::cdiv64(Ng, kTileBN[t]);
        const double mfma_per_ktile = (kTileBM[t] / 64.0) * (kTileBN[t] / 64.0) * 8.0 * 64.0 / tile_eff[t];
        const int smax = allow_split ? 16 : 1;
        for (int s = 1; s <= smax; ++s) {
            if (s > 1 && (nk / s < 8 || (int64_t)s * M * Ng * 4 >= (1ll << 31))) break;
            const int64_t kt = rg::cdiv64(nk, s);
            const int64_t B = tiles * rg::cdiv64(nk, kt);
            const double bt = kt * mfma_per_ktile + fixed;
            const int64_t r = rg::cdiv64(B, 256);
            double time = r <= 4 ? r * bt / util[r] : (double)B / 256.0 * bt / util[4];
            if (s > 1) time += 2400.0 + (double)(s + 1) * M * (double)Ng * 4.0 / 3.0e12 * 2.0e9;   // finishing kernel
            if (time < best * 0.97) {        // prefer bigger tiles / fewer splits unless clearly better
                best = time;
                pl.tile = t;
                best_s = s;
            }
        }
    }
    int splits = best_s;
    force_override(&pl.tile, &splits);
    if (!allow_split) splits = 1;
    pl.m_tiles = rg::cdiv(M, kTileBM[pl.tile]);
    pl.n_tiles = (int)rg::cdiv64(Ng, kTileBN[pl.tile]);
    pl.ktiles_per_split = (int)rg::cdiv64(nk, splits);
    pl.splits = (int)rg::cdiv64(nk, pl.ktiles_per_split);
    return pl;
}

// The cost model above was fitted to the round-1..3 kernels; against the eight-wave plane kernels it is off by one tile size or
// one split step on about a third of the ResNet geometries (tools/sweep_tiles.py, round 4: forward -6 %, data gradient -4 % with the
// best forced plan per layer).  So the PLAN is a measured choice too: the model's plan first, then every other tile of the shape
// class with 1 / 2 / 3 / 4 / 6 / 8 splits that keeps >= 4 k-tiles per split and <= 1 536 workgroups; an alternative has to beat the
// model's plan by 3 % (choose_impl).  RG_CONV_TUNE_PLAN=0 / RG_CONV_TUNE=0 / a forced plan / a GEMM under 1 GFLOP: the model's plan only.
static bool tune_enabled();
static const double kPlanTuneMinFlop = 1.0e9;      // below 1 GFLOP a launch takes ~10 us whatever the plan: not worth ~100 timed launches
static bool plan_tune_enabled() {
    static const int env = getenv("RG_CONV_TUNE_PLAN") ? atoi(getenv("RG_CONV_TUNE_PLAN")) : 1;
    return env != 0;
}
static GemmPlan make_plan(int M, int64_t Ng, int64_t nk, int tile, int splits) {
    GemmPlan pl;
    pl.tile = tile;
    pl.m_tiles = rg::cdiv(M, kTileBM[tile]);
    pl.n_tiles = (int)rg::cdiv64(Ng, kTileBN[tile]);
    pl.ktiles_per_split = (int)rg::cdiv64(nk, splits);
    pl.splits = (int)rg::cdiv64(nk, pl.ktiles_per_split);
    return pl;
}
static int plan_candidates(int M, int64_t Ng, int64_t Kg, GemmPlan* out, int max_out, bool only_bn128 = false) {
    out[0] = plan_gemm(M, Ng, Kg, true);
    int n = 1;
    if (g_force_tile >= 0 || g_force_splits >= 1 || !plan_tune_enabled() || !tune_enabled() || M <= 32) return n;
    if (2.0 * M * (double)Ng * (double)Kg < kPlanTuneMinFlop) return n;      // launch-bound sizes: nothing to choose between
    const int64_t nk = rg::cdiv64(Kg > 0 ? Kg : 1, BK);
    int tiles[3], nt = 0;
    if (Ng >= 128) {
        if (M > 64) tiles[nt++] = 0;
        tiles[nt++] = 1;
    }
    if (!only_bn128) tiles[nt++] = 2;
    static const int ss[6] = {1, 2, 3, 4, 6, 8};
    for (int ti = 0; ti < nt && n < max_out; ++ti)
        for (int si = 0; si < 6 && n < max_out; ++si) {
            const int t = tiles[ti], sp = ss[si];
            if (only_bn128 && sp > 1) continue;
            if (sp > 1 && (nk / sp < 4 || (int64_t)sp * M * Ng * 4 >= (1ll << 31))) continue;
            const int64_t wgs = (int64_t)rg::cdiv(M, kTileBM[t]) * rg::cdiv64(Ng, kTileBN[t]);
            if (sp > 1 && wgs * sp > 1536) continue;
            const GemmPlan pl = make_plan(M, Ng, nk, t, sp);
            bool dup = false;
            for (int i = 0; i < n; ++i) dup = dup || (out[i].tile == pl.tile && out[i].splits == pl.splits);
            if (!dup) out[n++] = pl;
        }
    return n;
}
static size_t plans_workspace(const GemmPlan* pl, int n, int M, int64_t Ng) {
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        const size_t b = pl[i].splits > 1 ? (size_t)pl[i].splits * M * (size_t)Ng * sizeof(float) : 0;
        if (b > need) need = b;
    }
    return need;
}

static unsigned finish_grid(int64_t n) {
    int64_t g = rg::cdiv64(n, 256);
    if (g > 4096) g = 4096;
    return (unsigned)(g < 1 ? 1 : g);
}

// development switch: RG_SPLITK_VEC=0 keeps the scalar finishing / reduction kernels
static bool splitk_vec() {
    static const int env = getenv("RG_SPLITK_VEC") ? atoi(getenv("RG_SPLITK_VEC")) : 1;
    return env != 0;
}

// ---- arrival counters for split-K without the finishing launch (rg_conv_splitk_arrivals) ----
// The library never allocates: the caller registers, per stream, a zero-initialised buffer of 32-bit counters that stays alive and is
// touched by nothing else.  Launches on one stream are ordered and every launch leaves its counters at zero (atomicInc wraps on the
// last arrival), so one buffer per stream serves every launch on it.  Not used inside a stream capture: a captured launch would carry
// the capture stream's counters into replays on whatever stream the graph is launched on, next to eager launches that use them.
// Measured (profiles/r04_splitk_inkernel.txt): a wash — the FD-GAN step 34.24-34.34 ms with the finishing kernels, 34.53-34.62 ms with
// the in-kernel finish on the same box (what a launch boundary costs, 1.5-1.9 us, is what the last arriver's acquire and its serial
// read of the tile's partials cost), so rg_hip registers counters only on request (RG_SPLITK_INKERNEL=1) and the default stays the
// finishing kernel.  The first form tried — __threadfence() in every workgroup — cost +27 us per split launch (36.9 vs 32.8 ms).
struct Arrivals { unsigned* ptr; int count; };
static std::map<hipStream_t, Arrivals> g_arrivals;
static std::mutex g_arrivals_mu;
static std::atomic<int> g_inkernel_launches{0};
// the counters a one-class split-K launch of `tiles` output tiles on `stream` may use, or nullptr (-> finishing kernel): the in-kernel
// finish moves float4s, so it takes the vector finisher's conditions
static unsigned* splitk_arrivals(hipStream_t stream, int tiles, const float* partial, const float* out, int M, int Ng, int PIX,
                                 const Epilogue& ep) {
    const bool al = ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ep.res) |
                      reinterpret_cast<uintptr_t>(ep.mask)) & 15) == 0;
    if (!al || (Ng & 3) || (PIX & 3) || (int64_t)M * Ng >= (1ll << 31) || ep.rowsum) return nullptr;
    Arrivals a = {nullptr, 0};
    {
        std::lock_guard<std::mutex> lock(g_arrivals_mu);
        auto it = g_arrivals.find(stream);
        if (it != g_arrivals.end()) a = it->second;
    }
    if (!a.ptr || tiles > a.count) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    ++g_inkernel_launches;
    return a.ptr;
}

static void launch_finish(hipStream_t stream, const float* partial, float* out, int M, int Ng, int PIX, const FastDiv& d_pix,
                          int splits, const Epilogue& ep) {
    const bool al = ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(ep.res) |
                      reinterpret_cast<uintptr_t>(ep.mask)) & 15) == 0;
    if (splitk_vec() && al && (Ng & 3) == 0 && (PIX & 3) == 0 && (int64_t)M * Ng < (1ll << 31)) {
        hipLaunchKernelGGL(conv_splitk_finish_vec_kernel, dim3(finish_grid((int64_t)M * (Ng >> 2))), dim3(256), 0, stream, partial, out,
                           M, Ng, PIX, d_pix, make_fastdiv(Ng >> 2), splits, ep);
        return;
    }
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(finish_grid((int64_t)M * Ng)), dim3(256), 0, stream, partial, out, M, Ng, PIX,
                       d_pix, splits, ep);
}

static void launch_reduce(hipStream_t stream, const float* ws, float* out, int64_t n, int splits, int rsc, int C, int RS) {
    const bool al = ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (splitk_vec() && al && (n & 3) == 0)
        hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)rg::cdiv64(n >> 2, 64)), dim3(256), 0, stream, ws, out, n, splits, rsc,
                           C, RS);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)rg::cdiv64(n, 64)), dim3(256), 0, stream, ws, out, n, splits, rsc, C, RS);
}

// Two implementations of the generic fwd / dgrad / wgrad kernels exist: the round-3 kernels (fp32 LDS tiles, every reading wave
// splits its fragments) and the bf16-plane path of conv_planes.h (operands split once on their way into LDS).  Neither wins
// everywhere: measured per layer they differ by up to +-20 % in both directions (profiles/r04_planes_vs_r03.txt: the plane path
// gains on the deep 1x1 layers and the 1x1 / shifted weight gradients, loses on the gather-loaded 4x4 layers).  So the choice is
// MEASURED, once per (family, geometry, plan): the first call of a geometry outside a stream capture runs both (one warm-up and
// one timed launch each, HIP events on the launch stream), keeps the faster (the plane path has to win by 3 %) and re-runs it if
// it was not the last one, so the output of every call — the first included — comes from the kernel that serves the geometry
// from then on: run-to-run bit-identity inside a process is untouched.  RG_CONV_TUNE=0: always the round-3 kernels.
// RG_CONV_PL / rg_conv_set_planes(mask): force the plane path per family (1 forward, 2 data gradient, 4 weight gradient).
static int g_planes_mask = -1;
static bool planes_enabled(int family_bit) {
    if (g_planes_mask < 0) g_planes_mask = getenv("RG_CONV_PL") ? atoi(getenv("RG_CONV_PL")) : 0;
    return (g_planes_mask & family_bit) != 0;
}
static bool path_tune_enabled() {
    static const int env = getenv("RG_CONV_TUNE_PATH") ? atoi(getenv("RG_CONV_TUNE_PATH")) : 1;
    return env != 0;
}
static bool tune_enabled() {
    static const int env = getenv("RG_CONV_TUNE") ? atoi(getenv("RG_CONV_TUNE")) : 1;
    return env != 0;
}
typedef std::array<int, 16> TuneKey;
static std::map<TuneKey, int> g_tune;
static std::mutex g_tune_mu;
// RG_CONV_TUNE_CACHE=<file>: measured choices are appended to the file (one line of 17 integers per geometry) and loaded from it by
// the next process, which then measures only what it has not seen — profiling runs (no measuring launches inside the trace) and
// runs that must repeat another process's kernels bit for bit use it
static const int kMaxCand = 16;          // candidates of one measured choice (3 kernel implementations; up to 16 tile / split plans)
static void tune_cache_load_locked() {
    static bool loaded = false;
    if (loaded) return;
    loaded = true;
    const char* path = getenv("RG_CONV_TUNE_CACHE");
    FILE* f = path ? fopen(path, "r") : nullptr;
    if (!f) return;
    TuneKey k;
    int choice;
    for (;;) {
        bool ok = true;
        for (int i = 0; i < 16 && ok; ++i) ok = fscanf(f, "%d", &k[i]) == 1;
        if (!ok || fscanf(f, "%d", &choice) != 1) break;
        g_tune[k] = choice < 0 ? 0 : (choice >= kMaxCand ? kMaxCand - 1 : choice);
    }
    fclose(f);
}
static void tune_cache_append_locked(const TuneKey& k, int choice) {
    const char* path = getenv("RG_CONV_TUNE_CACHE");
    FILE* f = path ? fopen(path, "a") : nullptr;
    if (!f) return;
    for (int i = 0; i < 16; ++i) fprintf(f, "%d ", k[i]);
    fprintf(f, "%d\n", choice);
    fclose(f);
}

// run(0): round-3 kernel, run(1): plane path, run(2) (ncand == 3: 128 x 128 tiles only): plane path with EIGHT waves per workgroup
// (4 x 2 waves of 32 x 64: the staging work per thread halves and four waves per SIMD hide each other's waits; +8 % on the
// micro-benchmark's 128 x 128 tile).  Each: the kernel launch only; split-K finishers follow.  Returns the implementation that ran
// LAST (= the chosen one).
template <typename Run>
static int choose_impl(int family_bit, const TuneKey& key, hipStream_t stream, int ncand, Run run) {
    if (planes_enabled(family_bit)) {
        const int c = (planes_enabled(8) && ncand > 2) ? 2 : 1;
        run(c);
        return c;
    }
    if (!tune_enabled()) { run(0); return 0; }
    static const int tune8 = getenv("RG_CONV_TUNE8") ? atoi(getenv("RG_CONV_TUNE8")) : 1;      // 0: the eight-wave form is not a candidate
    if (!tune8 && ncand == 3 && family_bit != 0) ncand = 2;
    {                                 // the lock covers the table only, never a launch: measurements may nest (path choice below)
        std::unique_lock<std::mutex> lock(g_tune_mu);
        tune_cache_load_locked();
        auto it = g_tune.find(key);
        if (it != g_tune.end() && it->second < ncand) {
            const int c = it->second;
            lock.unlock();
            run(c);
            return c;
        }
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        run(0);                       // no host synchronisation inside a capture: round-3 kernel, nothing recorded
        return 0;
    }
    if (ncand > kMaxCand) ncand = kMaxCand;
    hipEvent_t ev[2 * kMaxCand];
    bool ok = true;
    for (int i = 0; i < 2 * ncand; ++i) ok = hipEventCreate(&ev[i]) == hipSuccess && ok;
    float t[kMaxCand] = {0.f};
    if (ok) {
        // the candidates are timed on an otherwise idle GPU: work other streams were given earlier (weight gradients on the side
        // stream, D_pd on the auxiliary stream) would otherwise run beside some candidates and not others
        (void)hipDeviceSynchronize();
        for (int c = 0; c < ncand; ++c) run(c);       // warm-up (first launch of a code object loads it)
        for (int c = 0; c < ncand; ++c) {
            ok = hipEventRecord(ev[2 * c], stream) == hipSuccess && ok;
            run(c);
            ok = hipEventRecord(ev[2 * c + 1], stream) == hipSuccess && ok;
        }
        ok = hipEventSynchronize(ev[2 * ncand - 1]) == hipSuccess && ok;
        for (int c = 0; c < ncand && ok; ++c) ok = hipEventElapsedTime(&t[c], ev[2 * c], ev[2 * c + 1]) == hipSuccess;
    }
    for (int i = 0; i < 2 * ncand; ++i) (void)hipEventDestroy(ev[i]);
    if (!ok) {
        (void)hipGetLastError();
        run(0);
        return 0;
    }
    int choice = 0;                   // a plane-path candidate has to win by 3 %
    float best = t[0];
    for (int c = 1; c < ncand; ++c)
        if (t[c] < 0.97f * best) { best = t[c]; choice = c; }
    {
        std::lock_guard<std::mutex> lock(g_tune_mu);
        g_tune[key] = choice;
        tune_cache_append_locked(key, choice);
    }
    if (choice != ncand - 1) run(choice);             // the result must come from the chosen kernel
    return choice;
}

// ---- tap-reuse kernel (conv3x3_halo_kernel): geometry test, plan, launch ----
static bool halo_enabled() {
    static const int env = getenv("RG_CONV_HALO") ? atoi(getenv("RG_CONV_HALO")) : 1;
    return env != 0;
}
static bool halo_geom(int H, int W, int* HP, int* Wh, int* slab) {
    if (W < 4 || W > 64 || (W & (W - 1))) return false;
    const int R = 128 / W;                      // rows of a 128-pixel tile
    int rows, imgs;
    if (R <= H) {
        if (H % R) return false;
        rows = R;
        imgs = 1;
    } else {
        if (R % H) return false;                // whole images per tile
        rows = H;
        imgs = R / H;
    }
    *Wh = W + 2;
    *slab = (rows + 2) * (W + 2);
    *HP = imgs * *slab;
    return *HP <= 288;
}
struct HaloPlan {
    int bm, m_tiles, n_tiles, splits, per;
};
static HaloPlan halo_plan(int M, int64_t Ng, int Cred) {
    HaloPlan pl;
    pl.bm = M <= 64 ? 64 : 128;
    pl.m_tiles = rg::cdiv(M, pl.bm);
    pl.n_tiles = (int)rg::cdiv64(Ng, 128);
    const int cblocks = Cred / BK;
    const int64_t tiles = (int64_t)pl.m_tiles * pl.n_tiles;
    static const int target = getenv("RG_HALO_WG") ? atoi(getenv("RG_HALO_WG")) : 512;
    int64_t want = tiles >= (3 * target) / 4 ? 1 : rg::cdiv64(target, tiles);       // ~2 workgroups per CU
    if (want > cblocks / 2) want = cblocks / 2;                     // >= 2 channel blocks (18 k-tiles) per split
    if (want > 16) want = 16;
    if (want < 1) want = 1;
    while (want > 1 && want * (int64_t)M * Ng * 4 >= (1ll << 31)) --want;
    pl.per = (int)rg::cdiv64(cblocks, want);
    pl.splits = rg::cdiv(cblocks, pl.per);
    return pl;
}
static size_t halo_workspace(int M, int64_t Ng, int Cred) {
    const HaloPlan pl = halo_plan(M, Ng, Cred);
    return pl.splits > 1 ? (size_t)pl.splits * M * (size_t)Ng * sizeof(float) : 0;
}
// p: x / w (krsc) / y / ep / M / Ng / byte sizes filled by the caller; returns the launch status
template <bool DGRAD>
static int halo_launch(ConvP& p, int Cred, int H, int W, const HaloPlan& pl, void* workspace, hipStream_t stream, const char* op) {
    HaloP hp;
    int HPv, Wh, slab;
    halo_geom(H, W, &HPv, &Wh, &slab);
    p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles;
    p.splits = pl.splits;
    p.partial = pl.splits > 1 ? static_cast<float*>(workspace) : nullptr;
    p.partial_bytes = pl.splits > 1 ? (unsigned)((size_t)pl.splits * p.M * (size_t)p.Ng * sizeof(float)) : 0u;
    p.arrive = pl.splits > 1 ? splitk_arrivals(stream, pl.m_tiles * pl.n_tiles, p.partial, p.y, p.M, p.Ng, H * W, p.ep) : nullptr;
    hp.c = p;
    hp.Cred = Cred;
    hp.HP = HPv; hp.Wh = Wh; hp.slab = slab;
    hp.cblocks = Cred / BK;
    hp.cb_per_split = pl.per;
    hp.d_hp = make_fastdiv(HPv);
    hp.d_slab = make_fastdiv(slab);
    hp.d_wh = make_fastdiv(Wh);
    hp.d_hw = make_fastdiv(H * W);
    hp.d_w = make_fastdiv(W);
    const dim3 grid(pl.m_tiles * pl.n_tiles, pl.splits, 1);
    const bool small = 16 * HPv <= 13 * NT;
    if (pl.bm == 128) {
        if (small) hipLaunchKernelGGL((conv3x3_halo_kernel<128, DGRAD, 13>), grid, dim3(NT), 0, stream, hp);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<128, DGRAD, 18>), grid, dim3(NT), 0, stream, hp);
    } else {
        if (small) hipLaunchKernelGGL((conv3x3_halo_kernel<64, DGRAD, 13>), grid, dim3(NT), 0, stream, hp);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<64, DGRAD, 18>), grid, dim3(NT), 0, stream, hp);
    }
    if (pl.splits > 1 && !p.arrive) {
        if (int e = rg::check_launch(op)) return e;
        launch_finish(stream, p.partial, p.y, p.M, p.Ng, H * W, make_fastdiv(H * W), pl.splits, p.ep);
    }
    return rg::check_launch(op);
}

}  // namespace

#define RG_FWD_PL_LAUNCH(BM_, BN_, WM_, WN_)                                                                             \
    if (bmode == 2) hipLaunchKernelGGL((conv_fwd_pl_kernel<BM_, BN_, WM_, WN_, 2, true>), grid, dim3(64 * WM_ * WN_), 0, stream, p); \
    else if (bmode == 1) hipLaunchKernelGGL((conv_fwd_pl_kernel<BM_, BN_, WM_, WN_, 1, true>), grid, dim3(64 * WM_ * WN_), 0, stream, p); \
    else if (avec) hipLaunchKernelGGL((conv_fwd_pl_kernel<BM_, BN_, WM_, WN_, 0, true>), grid, dim3(64 * WM_ * WN_), 0, stream, p);  \
    else hipLaunchKernelGGL((conv_fwd_pl_kernel<BM_, BN_, WM_, WN_, 0, false>), grid, dim3(64 * WM_ * WN_), 0, stream, p)

#define RG_FWD_LAUNCH(BM_, BN_, WM_, WN_)                                                                             \
    if (bmode == 2) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 2, true>), grid, dim3(NT), 0, stream, p); \
    else if (bmode == 1) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 1, true>), grid, dim3(NT), 0, stream, p); \
    else if (avec) hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 0, true>), grid, dim3(NT), 0, stream, p);  \
    else hipLaunchKernelGGL((conv_fwd_kernel<BM_, BN_, WM_, WN_, 0, false>), grid, dim3(NT), 0, stream, p)

#define RG_DGRAD_LAUNCH(BM_, BN_, WM_, WN_)                                                                          \
    if (mode == 2) hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 2>), grid, dim3(NT), 0, stream, dp);    \
    else if (mode == 1) hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 1>), grid, dim3(NT), 0, stream, dp); \
    else hipLaunchKernelGGL((conv_dgrad_kernel<BM_, BN_, WM_, WN_, 0>), grid, dim3(NT), 0, stream, dp)

#define RG_DGRAD_PL_LAUNCH(BM_, BN_, WM_, WN_)                                                                          \
    if (mode == 2) hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, 2>), grid, dim3(64 * WM_ * WN_), 0, stream, dp);    \
    else if (mode == 1) hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, 1>), grid, dim3(64 * WM_ * WN_), 0, stream, dp); \
    else hipLaunchKernelGGL((conv_dgrad_pl_kernel<BM_, BN_, WM_, WN_, 0>), grid, dim3(64 * WM_ * WN_), 0, stream, dp)

#define RG_TILE_SWITCH(tile, LAUNCH)      \
    switch (tile) {                       \
        case 0: LAUNCH(128, 128, 2, 2); break; \
        case 1: LAUNCH(64, 128, 2, 2); break;  \
        case 2: LAUNCH(64, 64, 2, 2); break;   \
        default: LAUNCH(32, 256, 1, 4); break; \
    }

namespace {
// layers with <= 4 output channels (conv_fwd_k1_kernel / conv_wgrad_k1_kernel): filter sizes with an instantiation
static bool thin_filter(int K, int KH, int KW) {
    static const int env = getenv("RG_THIN_CONV") ? atoi(getenv("RG_THIN_CONV")) : 1;
    return env && K >= 1 && K <= 4 && ((KH == 4 && KW == 4 && K == 1) || (KH == 3 && KW == 3));
}
#define THIN_DISPATCH(KERNEL, grid, stream, t)                                                       \
    do {                                                                                              \
        if (KH == 4) hipLaunchKernelGGL((KERNEL<4, 4, 1>), grid, dim3(256), 0, stream, t);            \
        else if (K == 1) hipLaunchKernelGGL((KERNEL<3, 3, 1>), grid, dim3(256), 0, stream, t);        \
        else if (K == 2) hipLaunchKernelGGL((KERNEL<3, 3, 2>), grid, dim3(256), 0, stream, t);        \
        else if (K == 3) hipLaunchKernelGGL((KERNEL<3, 3, 3>), grid, dim3(256), 0, stream, t);        \
        else hipLaunchKernelGGL((KERNEL<3, 3, 4>), grid, dim3(256), 0, stream, t);                    \
    } while (0)
// four pixels per thread (conv_wgrad_k1_px4_kernel): 3x3 / stride 1 / pad <= 1 rows of float4 multiples
static bool thin_px4(int KH, int KW, int SH, int SW, int PH, int PW, int W, int Q, const void* a, const void* partial) {
    static const int env = getenv("RG_THIN_PX4") ? atoi(getenv("RG_THIN_PX4")) : 1;
    return env && KH == 3 && KW == 3 && SH == 1 && SW == 1 && PH <= 1 && PW <= 1 && (Q & 3) == 0 && Q + 2 - 2 * PW == W &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0;
}
#define THIN_PX4_DISPATCH(KERNEL, grid, stream, t)                                                    \
    do {                                                                                              \
        if (K == 1) hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, stream, t);                   \
        else if (K == 2) hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, stream, t);              \
        else if (K == 3) hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), 0, stream, t);              \
        else hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, stream, t);                          \
    } while (0)
// forward: channels per slice so that ~2048 workgroups exist (>= 4 channels each)
static int thin_fwd_per_slice(int C, int64_t Ng) {
    int64_t slices = rg::cdiv64(2048, rg::cdiv64(Ng, 256));
    if (slices > C / 4) slices = C / 4;
    if (slices < 1) slices = 1;
    return (int)rg::cdiv64(C, slices);
}
// weight gradient: pixels per slice so that ~2048 workgroups exist (>= 512 pixels each)
static int thin_wgrad_per_slice(int C, int64_t Ng) {
    int64_t slices = rg::cdiv64(2048, C);
    if (slices > Ng / 512) slices = Ng / 512;
    if (slices < 1) slices = 1;
    return (int)((rg::cdiv64(Ng, slices) + 3) / 4 * 4);      // a multiple of four pixels (conv_wgrad_k1_px4_kernel)
}
static void thin_fill(ThinP& t, const float* x, const float* a, float* partial, int N, int C, int H, int W, int SH, int SW,
                      int PH, int PW, int P, int Q, int per_slice) {
    t.x = x; t.a = a; t.partial = partial;
    t.N = N; t.C = C; t.H = H; t.W = W; t.P = P; t.Q = Q; t.SH = SH; t.SW = SW; t.PH = PH; t.PW = PW;
    t.per_slice = per_slice;
    t.x_bytes = (unsigned)((int64_t)N * C * H * W * 4);
    t.d_pq = make_fastdiv(P * Q);
    t.d_q = make_fastdiv(Q);
}
}  // namespace

extern "C" size_t rg_conv2d_fwd_workspace(int N, int C, int K, int KH, int KW, int P, int Q) {
    if (thin_filter(K, KH, KW)) {
        const int64_t Ng = (int64_t)N * P * Q;
        return (size_t)rg::cdiv(C, thin_fwd_per_slice(C, Ng)) * (size_t)K * (size_t)Ng * sizeof(float);
    }
    GemmPlan cands[kMaxCand];
    const int nc = plan_candidates(K, (int64_t)N * P * Q, (int64_t)C * KH * KW, cands, kMaxCand);
    size_t need = plans_workspace(cands, nc, K, (int64_t)N * P * Q);
    int hpv, wh, slab;
    if (KH == 3 && KW == 3 && C % BK == 0 && K >= 64 && halo_enabled() && halo_geom(P, Q, &hpv, &wh, &slab)) {   // stride 1 / pad 1: P x Q = H x W
        const size_t hn = halo_workspace(K, (int64_t)N * P * Q, C);
        if (hn > need) need = hn;
    }
    return need;
}

// w_krsc (optional): weights re-laid out as [K][KH*KW][C] (rg_weights_to_krsc); with C % 16 == 0 it selects the
// (r,s)-major reduction order whose pixel gather tests the padding bounds once per 16-deep k-tile.
extern "C" int rg_conv2d_fwd(const float* x, const float* w, const float* w_krsc, float* y, int N, int C, int H, int W,
                             int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q, const float* scale,
                             const float* shift, const float* residual, int act, float slope, void* workspace,
                             size_t workspace_bytes, hipStream_t stream) {
    if (int e = validate("rg_conv2d_fwd", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(x && w && y, "rg_conv2d_fwd: null tensor");
    ConvP p;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = x; p.w = w; p.y = y;
    p.ep = Epilogue{scale, shift, residual, act, slope, nullptr, nullptr, 0};
    p.M = K; p.Ng = N * P * Q; p.Kg = C * KH * KW;
    p.x_bytes = (unsigned)((int64_t)N * C * H * W * 4);
    p.w_bytes = (unsigned)((int64_t)K * C * KH * KW * 4);
    p.y_bytes = (unsigned)((int64_t)N * K * P * Q * 4);
    if (thin_filter(K, KH, KW)) {
        const int per = thin_fwd_per_slice(C, p.Ng);
        const int slices = rg::cdiv(C, per);
        const size_t need = (size_t)slices * (size_t)K * (size_t)p.Ng * sizeof(float);
        if (workspace && need <= workspace_bytes) {
            ThinP t;
            thin_fill(t, x, w, static_cast<float*>(workspace), N, C, H, W, SH, SW, PH, PW, P, Q, per);
            rg::ProfScope prof(rg::FAM_CONV_FWD, stream, 2.0 * (double)K * p.Ng * p.Kg, ALG_BYTES);
            const dim3 grid(rg::cdiv(p.Ng, 256), slices);
            THIN_DISPATCH(conv_fwd_k1_kernel, grid, stream, t);      // (a four-pixel forward was measured: 136 us against 120)
            if (int e = rg::check_launch("rg_conv2d_fwd(thin)")) return e;
            launch_finish(stream, t.partial, y, K, p.Ng, P * Q, p.d_pq, slices, p.ep);
            return rg::check_launch("rg_conv2d_fwd(thin finish)");
        }
    }
    // 3x3 / stride 1 / pad 1 layers: the tap-reuse kernel or the generic implicit GEMM on the (r,s)-major filters — whichever is
    // faster for the geometry, measured once like the kernel implementations (RG_CONV_TUNE_PATH=0: always the tap-reuse kernel)
    bool halo_ok = false;
    HaloPlan hpl = HaloPlan();
    {
        int hpv, wh, slab;
        // (fewer than 64 output rows: the 32 x 256 tile of the generic kernel wastes less than a half-empty 64-row tile)
        if (KH == 3 && KW == 3 && SH == 1 && SW == 1 && PH == 1 && PW == 1 && C % BK == 0 && K >= 64 && w_krsc &&
            ((reinterpret_cast<uintptr_t>(w_krsc) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && halo_enabled() &&
            halo_geom(H, W, &hpv, &wh, &slab)) {
            halo_ok = true;
            hpl = halo_plan(p.M, p.Ng, C);
            const size_t need = hpl.splits > 1 ? (size_t)hpl.splits * p.M * (size_t)p.Ng * sizeof(float) : 0;
            if (need > workspace_bytes || (need && !workspace)) {
                hpl.splits = 1;
                hpl.per = C / BK;
            }
        }
    }
    const ConvP p0 = p;
    auto launch_halo = [&]() -> int {
        ConvP ph = p0;
        ph.w = w_krsc;
        rg::ProfScope prof(rg::FAM_CONV_FWD, stream, 2.0 * ph.M * (double)ph.Ng * ph.Kg, ALG_BYTES);
        return halo_launch<false>(ph, C, H, W, hpl, workspace, stream, "rg_conv2d_fwd(3x3 tap reuse)");
    };
    auto launch_generic = [&]() -> int {
        p = p0;
        const bool is1x1 = KH == 1 && KW == 1;
        const bool aligned = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
        const bool avec = (p.Kg % 4 == 0) && aligned;
        int bmode = 0;
        if (avec && is1x1 && SH == 1 && SW == 1 && PH == 0 && PW == 0 && ((H * W) % 4 == 0)) bmode = 2;
        else if (avec && C % 16 == 0 && (is1x1 || (w_krsc && (reinterpret_cast<uintptr_t>(w_krsc) & 15) == 0))) {
            bmode = 1;
            if (!is1x1) p.w = w_krsc;
        }
        GemmPlan cands[kMaxCand];
        int nc = plan_candidates(p.M, p.Ng, p.Kg, cands, kMaxCand);
        if (nc > 1 && (plans_workspace(cands, nc, p.M, p.Ng) > workspace_bytes || !workspace)) nc = 1;     // not the queried scratch
        const ConvP pg = p;
        auto run_plan = [&](GemmPlan pl) -> int {
        p = pg;
        const size_t need = pl.splits > 1 ? (size_t)pl.splits * p.M * (size_t)p.Ng * sizeof(float) : 0;
        if (need > workspace_bytes || (need && !workspace)) {     // no scratch given: run unsplit
            pl.splits = 1;
            pl.ktiles_per_split = 1 << 30;
        }
        p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles;
        p.splits = pl.splits; p.ktiles_per_split = pl.ktiles_per_split;
        p.partial = pl.splits > 1 ? static_cast<float*>(workspace) : nullptr;
        p.partial_bytes = (unsigned)need;
        p.arrive = pl.splits > 1 ? splitk_arrivals(stream, p.m_tiles * p.n_tiles, p.partial, y, p.M, p.Ng, P * Q, p.ep) : nullptr;
        rg::ProfScope prof(rg::FAM_CONV_FWD, stream, 2.0 * p.M * (double)p.Ng * p.Kg, ALG_BYTES);
        const dim3 grid(p.m_tiles * p.n_tiles, pl.splits, 1);
        const TuneKey tk = {1, N, C, H, W, K, KH, KW, SH, SW, PH, PW, bmode * 2 + (avec ? 1 : 0), pl.tile, pl.splits,
                            (int)(p.ep.res != nullptr) * 4 + p.ep.act};
        choose_impl(1, tk, stream, pl.tile <= 1 ? 3 : 2, [&](int impl) {
            if (impl == 2) {                  // eight waves: 128 x 128 as 4 x 2 waves of 32 x 64, 64 x 128 as 2 x 4 waves of 32 x 32
                if (pl.tile == 0) { RG_FWD_PL_LAUNCH(128, 128, 4, 2); }
                else { RG_FWD_PL_LAUNCH(64, 128, 2, 4); }
            }
            else if (impl) { RG_TILE_SWITCH(pl.tile, RG_FWD_PL_LAUNCH); }
            else { RG_TILE_SWITCH(pl.tile, RG_FWD_LAUNCH); }
        });
        if (pl.splits > 1 && !p.arrive) {
            if (int e = rg::check_launch("rg_conv2d_fwd")) return e;
            launch_finish(stream, p.partial, y, p.M, p.Ng, P * Q, p.d_pq, pl.splits, p.ep);
        }
        return rg::check_launch("rg_conv2d_fwd");
        };
        if (nc > 1) {
            int st = RG_OK;
            const TuneKey pk = {64, N, C, H, W, K, KH, KW, SH, SW, PH, PW, bmode * 2 + (avec ? 1 : 0), nc, 0,
                                (int)(pg.ep.res != nullptr) * 4 + pg.ep.act};
            choose_impl(0, pk, stream, nc, [&](int c) {
                const int e = run_plan(cands[c]);
                if (e) st = e;
            });
            return st;
        }
        return run_plan(cands[0]);
    };
    if (halo_ok) {
        if (!tune_enabled() || !path_tune_enabled()) return launch_halo();
        int st = RG_OK;
        const TuneKey hk = {16, N, C, H, W, K, KH, KW, SH, SW, PH, PW, 0, 0, 0, (int)(p0.ep.res != nullptr) * 4 + p0.ep.act};
        choose_impl(0, hk, stream, 2, [&](int c) {
            const int e = c ? launch_generic() : launch_halo();
            if (e) st = e;
        });
        return st;
    }
    return launch_generic();
}

// development knob (tools/sweep_tiles.py): pin the planner's tile / split choice at run time; (-1, -1) releases it
extern "C" int rg_conv_set_force(int tile, int splits) {
    g_force_tile = tile;
    g_force_splits = splits;
    return RG_OK;
}

// development knob (tests, tools/bench_conv.py): kernel families on the bf16-plane operand path (bit 0 fwd, 1 dgrad, 2 wgrad);
// returns the previous mask
extern "C" int rg_conv_set_planes(int mask) {
    const int old = g_planes_mask < 0 ? (getenv("RG_CONV_PL") ? atoi(getenv("RG_CONV_PL")) : 0) : g_planes_mask;
    g_planes_mask = mask & 15;
    return old;
}

// number of choices the first-call chooser has measured so far (kernel implementation per (family, geometry, plan); tap-reuse vs
// generic path and tile / split plan per geometry); out[0] / out[1] (may be NULL): how many of the kernel-implementation choices
// went to the round-3 kernels / the plane path
extern "C" int rg_conv_tune_stats(int* out) {
    std::lock_guard<std::mutex> lock(g_tune_mu);
    int n[2] = {0, 0};
    for (const auto& kv : g_tune)
        if (kv.first[0] == 1 || kv.first[0] == 2 || kv.first[0] == 4) ++n[kv.second ? 1 : 0];      // out[1]: plane path, four or eight waves
    if (out) { out[0] = n[0]; out[1] = n[1]; }
    return (int)g_tune.size();
}

// Registers (count > 0) or removes (counters == NULL) the arrival counters of split-K launches on `stream`: `count` zero-initialised
// 32-bit words that the caller keeps alive and never writes.  With them the split forward / unit-stride data-gradient launches on
// that stream finish inside the convolution kernel (the last split of a tile to arrive sums the tile's partials in split order: the
// finishing kernel's values bit for bit); without them — or for launches with more tiles than `count`, or inside a stream capture —
// a finishing kernel follows as before.
extern "C" int rg_conv_splitk_arrivals(void* counters, int count, hipStream_t stream) {
    RG_REQUIRE(!counters || count > 0, "rg_conv_splitk_arrivals: count must be positive");
    RG_REQUIRE((reinterpret_cast<uintptr_t>(counters) & 3) == 0, "rg_conv_splitk_arrivals: counters must be 4-byte aligned");
    std::lock_guard<std::mutex> lock(g_arrivals_mu);
    if (counters) g_arrivals[stream] = Arrivals{static_cast<unsigned*>(counters), count};
    else g_arrivals.erase(stream);
    return RG_OK;
}

// test / development query: split-K launches that finished inside the convolution kernel so far (this process)
extern "C" int rg_conv_splitk_inkernel_count(void) { return g_inkernel_launches.load(); }

extern "C" size_t rg_conv2d_dgrad_workspace(int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW) {
    if (SH != 1 || SW != 1) {
        // strided: the classes' partial columns side by side (all N*H*W pixels); the plan depends on the mean class depth, which
        // does not depend on the padding (the classes' tap counts are a permutation) — planned as dgrad_impl does for pad 0
        if (SH > 2 || SW > 2) return 0;
        int64_t ng_eff = 0;
        double kw_sum = 0.0;
        for (int ah = 0; ah < SH; ++ah)
            for (int aw = 0; aw < SW; ++aw) {
                const int r0 = ah % SH, s0 = aw % SW;
                const int nrh = r0 < KH ? (KH - r0 + SH - 1) / SH : 0, nrw = s0 < KW ? (KW - s0 + SW - 1) / SW : 0;
                const int Hc = ah < H ? (H - ah + SH - 1) / SH : 0, Wc = aw < W ? (W - aw + SW - 1) / SW : 0;
                const int64_t ngc = (int64_t)N * Hc * Wc;
                if (nrh * nrw > 0) {
                    ng_eff += ngc;
                    kw_sum += (double)ngc * K * nrh * nrw;
                }
            }
        if (ng_eff <= 0 || C <= 4) return 0;
        static const int strided_split = getenv("RG_DGRAD_STRIDED_SPLIT") ? atoi(getenv("RG_DGRAD_STRIDED_SPLIT")) : 1;
        if (!strided_split || (int64_t)C * N * H * W >= (1ll << 31)) return 0;
        const int64_t kg_deep = (int64_t)K * ((KH + SH - 1) / SH) * ((KW + SW - 1) / SW);      // the deepest class (dgrad_impl plans on it)
        GemmPlan cands[kMaxCand];
        const int nc = plan_candidates(C, ng_eff, kg_deep, cands, kMaxCand);
        int smax = 1;
        for (int i = 0; i < nc; ++i) smax = cands[i].splits > smax ? cands[i].splits : smax;
        // one split more than planned: a padding whose classes order differently may plan one more
        const size_t need = smax > 1 ? (size_t)(smax + 1) * C * (size_t)N * H * W * sizeof(float) : 0;
        if (need < (1ull << 31)) return need;
        return cands[0].splits > 1 ? (size_t)(cands[0].splits + 1) * C * (size_t)N * H * W * sizeof(float) : 0;
    }
    GemmPlan cands[kMaxCand];
    const int nc = plan_candidates(C, (int64_t)N * H * W, (int64_t)K * KH * KW, cands, kMaxCand);
    size_t need = plans_workspace(cands, nc, C, (int64_t)N * H * W);
    if (need >= (1ull << 31)) need = cands[0].splits > 1 ? (size_t)cands[0].splits * C * (size_t)N * H * W * sizeof(float) : 0;
    int hpv, wh, slab;
    if (KH == 3 && KW == 3 && K % BK == 0 && C % 4 == 0 && C >= 64 && halo_enabled() && halo_geom(H, W, &hpv, &wh, &slab)) {
        const size_t hn = halo_workspace(C, (int64_t)N * H * W, K);
        if (hn > need) need = hn;
    }
    return need;
}

// w_krsc: the weights re-laid out as [K][KH*KW][C] by rg_weights_to_krsc (may be NULL; for 1x1 filters the
// original tensor already has that layout and is used directly).
namespace {
// `dry` != nullptr: plan only (as if the queried workspace were supplied) and report the number of row-sum column blocks the
// launch would write (0: split-K or the small-C kernel, which have no fused row sums); nothing is launched.
int dgrad_impl(const float* dy, const float* w, const float* w_krsc, float* dx, int N, int C, int H, int W, int K, int KH, int KW,
               int SH, int SW, int PH, int PW, int P, int Q, const float* scale, const float* shift, const float* residual,
               int act, float slope, const float* relu_mask, float* rowsum, int rowsum_cols, void* workspace,
               size_t workspace_bytes, hipStream_t stream, int* dry, bool allow_halo = true) {
    if (int e = validate("rg_conv2d_dgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(dry || (dy && w && dx), "rg_conv2d_dgrad: null tensor");
    RG_REQUIRE(SH <= 2 && SW <= 2, "rg_conv2d_dgrad: stride > 2 not supported (got %d,%d)", SH, SW);
    if (dry) *dry = 0;
    DgradP dp;
    ConvP& p = dp.c;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = dy; p.w = w; p.y = dx;
    p.ep = Epilogue{scale, shift, residual, act, slope, relu_mask, rowsum, rowsum_cols};
    p.M = C;
    p.x_bytes = (unsigned)((int64_t)N * K * P * Q * 4);
    p.w_bytes = (unsigned)((int64_t)K * C * KH * KW * 4);
    p.y_bytes = (unsigned)((int64_t)N * C * H * W * 4);
    int64_t ng_max = 0, kg_max = 0;
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
            if (cl.Kgc > kg_max) kg_max = cl.Kgc;
            flops += 2.0 * C * (double)cl.Ngc * cl.Kgc;
        }
    const bool one_class = SH == 1 && SW == 1;
    if (C <= 4 && (size_t)K * KH * KW * 4 * sizeof(float) <= 64 * 1024) {      // RGB-sized outputs: direct kernel
        if (dry) return RG_OK;
        RG_REQUIRE(!rowsum, "rg_conv2d_dgrad: row sums are not available on the small-C path (query rg_conv2d_dgrad_rowsum_cols)");
        int nmax = 0;
        for (int i = 0; i < SH * SW; ++i)
            if (dp.cls[i].Ngc > nmax) nmax = dp.cls[i].Ngc;
        p.Ng = nmax;
        p.Kg = K * KH * KW;
        rg::ProfScope prof(rg::FAM_CONV_DGRAD, stream, flops, ALG_BYTES);
        bool few_taps = C <= 3;
        int gmax = 0;
        for (int i = 0; i < SH * SW; ++i) {
            few_taps = few_taps && dp.cls[i].nrw <= 4 && dp.cls[i].nrh <= 4;
            const int g = N * rg::cdiv(dp.cls[i].Hc, 4) * dp.cls[i].Wc;
            if (g > gmax) gmax = g;
        }
        static const int px_env = getenv("RG_SMALLC_PX") ? atoi(getenv("RG_SMALLC_PX")) : 1;
        if (few_taps && px_env && gmax > 0)
            hipLaunchKernelGGL((conv_dgrad_smallc_px_kernel<4>), dim3(((rg::cdiv(gmax, 256) + 7) / 8) * 8 * SH * SW), dim3(256),
                               (size_t)K * KH * KW * 4 * sizeof(float), stream, dp);
        else
            hipLaunchKernelGGL((conv_dgrad_smallc_kernel<4>), dim3(rg::cdiv(nmax, 256), 1, SH * SW), dim3(256),
                               (size_t)K * KH * KW * 4 * sizeof(float), stream, dp);
        return rg::check_launch("rg_conv2d_dgrad(small-C)");
    }
    // weight operand layout / loader
    const bool is1x1 = KH == 1 && KW == 1;
    const bool w_al = (reinterpret_cast<uintptr_t>(w) & 15) == 0, dy_al = (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    const float* wk = is1x1 ? (w_al ? w : nullptr)
                            : ((w_krsc && (reinterpret_cast<uintptr_t>(w_krsc) & 15) == 0) ? w_krsc : nullptr);
    int mode = 0;
    if (wk && C % 4 == 0) {
        if (is1x1 && one_class && PH == 0 && PW == 0 && ((P * Q) % 4 == 0) && dy_al) mode = 2;
        else if (K % 16 == 0) mode = 1;
        if (mode) p.w = wk;
    }
    {
        int hpv, wh, slab;
        // (a planning call has no pointers: rg_hip.ops always passes the [K][9][C] copy for such layers)
        if (one_class && KH == 3 && KW == 3 && PH == 1 && PW == 1 && P == H && Q == W && K % BK == 0 && C % 4 == 0 && C >= 64 &&
            (dry || (wk && dy_al)) && halo_enabled() && allow_halo && halo_geom(H, W, &hpv, &wh, &slab)) {
            HaloPlan hpl = halo_plan(p.M, ng_max, K);
            const size_t need = hpl.splits > 1 ? (size_t)hpl.splits * p.M * (size_t)ng_max * sizeof(float) : 0;
            if (!dry && (need > workspace_bytes || (need && !workspace))) {
                hpl.splits = 1;
                hpl.per = K / BK;
            }
            const int cols = hpl.splits > 1 ? 0 : hpl.n_tiles * 2;
            if (dry) {
                *dry = cols;
                return RG_OK;
            }
            RG_REQUIRE(!rowsum || (cols > 0 && rowsum_cols == cols),
                       "rg_conv2d_dgrad: rowsum_cols %d does not match this launch (%d; query rg_conv2d_dgrad_rowsum_cols)",
                       rowsum_cols, cols);
            p.w = wk;
            p.Ng = (int)ng_max;
            p.Kg = K * KH * KW;
            const ConvP ph0 = p;
            auto launch_halo = [&]() -> int {
                ConvP ph = ph0;
                rg::ProfScope prof(rg::FAM_CONV_DGRAD, stream, flops, ALG_BYTES);
                return halo_launch<true>(ph, K, H, W, hpl, workspace, stream, "rg_conv2d_dgrad(3x3 tap reuse)");
            };
            // the tap-reuse kernel and the generic kernels compete per geometry, as in the forward pass — with fused row sums only
            // when the generic plan writes the same row-sum columns (their count is the caller's contract with the planning query:
            // both unsplit on 128-pixel tiles with two wave columns -> the same column per (n-tile, wave column))
            bool can_tune = tune_enabled() && path_tune_enabled();
            if (can_tune && rowsum) {
                int gcols = -1;
                can_tune = dgrad_impl(nullptr, nullptr, nullptr, nullptr, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, nullptr, nullptr, nullptr,
                                      act, slope, nullptr, nullptr, 0, nullptr, 0, nullptr, &gcols, false) == RG_OK && gcols == cols;
            }
            if (can_tune) {
                int st = RG_OK;
                const TuneKey hk = {32, N, C, H, W, K, KH, KW, SH, SW, PH, PW, 0, 0, 0,
                                    (int)(residual != nullptr) * 16 + (int)(relu_mask != nullptr) * 8 + (int)(rowsum != nullptr) * 4 + act};
                choose_impl(0, hk, stream, 2, [&](int c) {
                    const int e = c ? dgrad_impl(dy, w, w_krsc, dx, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, scale, shift, residual, act,
                                                 slope, relu_mask, rowsum, rowsum_cols, workspace, workspace_bytes, stream, nullptr, false)
                                    : launch_halo();
                    if (e) st = e;
                });
                return st;
            }
            return launch_halo();
        }
    }
    // strided classes: only classes that have filter taps do MFMA work (a 1x1 / stride-2 layer has ONE such class, the
    // others only zero-fill), and their reductions differ — plan on the columns that carry work and their mean depth
    int64_t ng_eff = 0;
    double kw_sum = 0.0;
    for (int i = 0; i < SH * SW; ++i)
        if (dp.cls[i].Kgc > 0) {
            ng_eff += dp.cls[i].Ngc;
            kw_sum += (double)dp.cls[i].Ngc * dp.cls[i].Kgc;
        }
    const int64_t kg_eff = ng_eff > 0 ? (int64_t)(kw_sum / (double)ng_eff) : kg_max;
    // Strided classes may split their reductions too (RG_DGRAD_STRIDED_SPLIT=0: never): the 3x3 / 1x1 stride-2 layers of layer3 /
    // layer4 have 32-128 tiles for 256 CUs and class depths that differ 4x (1, 2, 2 and 4 taps); every class is cut into the same
    // number of splits of ITS depth, the partial columns of the classes lie side by side and conv_splitk_finish_strided_kernel
    // scatters the sums to the classes' pixels with the fused epilogue.
    static const int strided_split = getenv("RG_DGRAD_STRIDED_SPLIT") ? atoi(getenv("RG_DGRAD_STRIDED_SPLIT")) : 1;
    // (planned on the DEEPEST class when splitting is allowed: the classes' workgroups start together, so an unsplit launch lasts
    // as long as its deepest class — 4 taps against a mean of 2.25 for 3x3 / 2 — and planning on the mean depth kept the layer4
    // gradients unsplit on 64 x 64 tiles: l4.0.conv2 95 -> 71 us, l4.0.down 83 -> 71 us with four splits, profiles/r04_strided_dgrad.txt)
    const bool ssplit = strided_split != 0 && ng_eff > 0 && (int64_t)p.M * N * H * W < (1ll << 31);
    GemmPlan pl = one_class ? plan_gemm(p.M, ng_max, kg_max, true)
                            : plan_gemm(p.M, ng_eff > 0 ? ng_eff : ng_max, ssplit ? kg_max : kg_eff, ssplit);
    int64_t ng_total = 0;
    for (int i = 0; i < SH * SW; ++i) {
        dp.cls[i].coff = (int)ng_total;
        ng_total += dp.cls[i].Ngc;
    }
    dp.ng_total = (int)ng_total;
    const int wn_of_tile[4] = {2, 2, 2, 4};               // wave columns of the tile shapes in RG_TILE_SWITCH
    auto cols_of = [&](const GemmPlan& q) -> int {
        if (q.splits > 1) return 0;
        int nts = 0;
        for (int i = 0; i < SH * SW; ++i) nts += rg::cdiv(dp.cls[i].Ngc, kTileBN[q.tile]);
        return nts * wn_of_tile[q.tile];
    };
    // measured plan choice (plan_candidates) for the one-class launches: with fused row sums only plans that write the column count
    // the caller sized (rg_conv2d_dgrad_rowsum_cols reports the model's plan) qualify
    GemmPlan cands[kMaxCand];
    int nc = 1;
    cands[0] = pl;
    if (!dry && (one_class || ng_eff > 0)) {
        GemmPlan all[kMaxCand];
        const int na = one_class ? plan_candidates(p.M, ng_max, kg_max, all, kMaxCand)
                                 : plan_candidates(p.M, ng_eff, ssplit ? kg_max : kg_eff, all, kMaxCand);
        // (strided classes: the partial columns of all classes lie side by side, ng_total columns per split)
        const size_t ws_all = plans_workspace(all, na, p.M, one_class ? ng_max : ng_total);
        if (na > 1 && all[0].tile == pl.tile && all[0].splits == pl.splits && workspace && ws_all <= workspace_bytes &&
            ws_all < (1ull << 31)) {
            const int want = cols_of(pl);
            for (int i = 1; i < na; ++i)
                if ((one_class || ssplit || all[i].splits == 1) && (!rowsum || cols_of(all[i]) == want)) cands[nc++] = all[i];
        }
    }
    const DgradP dp0 = dp;
    auto run_plan = [&](GemmPlan pl) -> int {
    dp = dp0;
    size_t need = pl.splits > 1 ? (size_t)pl.splits * p.M * (size_t)(one_class ? ng_max : ng_total) * sizeof(float) : 0;
    if (need >= (1ull << 31) || (!dry && (need > workspace_bytes || (need && !workspace)))) {
        pl.splits = 1;
        pl.ktiles_per_split = 1 << 30;
        need = 0;
    }
    for (int i = 0; i < SH * SW; ++i)          // k-tiles per split: the plan's for one class, each class' own depth / splits otherwise
        dp.cls[i].ktps = one_class ? pl.ktiles_per_split
                                   : (pl.splits > 1 ? (int)rg::cdiv64(rg::cdiv64(dp.cls[i].Kgc > 0 ? dp.cls[i].Kgc : 1, BK), pl.splits) : (1 << 30));
    p.m_tiles = pl.m_tiles;
    int nt_max = 0, nt_sum = 0;
    for (int i = 0; i < SH * SW; ++i) {
        dp.cls[i].ntiles = rg::cdiv(dp.cls[i].Ngc, kTileBN[pl.tile]);
        dp.cls[i].poff = nt_sum;
        nt_sum += dp.cls[i].ntiles;
        if (dp.cls[i].ntiles > nt_max) nt_max = dp.cls[i].ntiles;
    }
    const int wn_waves = wn_of_tile[pl.tile];
    const int cols = pl.splits > 1 ? 0 : nt_sum * wn_waves;
    if (dry) {
        *dry = cols;
        return RG_OK;
    }
    RG_REQUIRE(!rowsum || (cols > 0 && rowsum_cols == cols),
               "rg_conv2d_dgrad: rowsum_cols %d does not match this launch (%d; query rg_conv2d_dgrad_rowsum_cols)", rowsum_cols,
               cols);
    p.n_tiles = nt_max;
    p.Ng = (int)ng_max;
    p.Kg = K * KH * KW;
    p.splits = pl.splits; p.ktiles_per_split = pl.ktiles_per_split;
    p.partial = pl.splits > 1 ? static_cast<float*>(workspace) : nullptr;
    p.partial_bytes = (unsigned)need;
    p.arrive = (pl.splits > 1 && one_class) ? splitk_arrivals(stream, p.m_tiles * nt_max, p.partial, dx, p.M, (int)ng_max, H * W, p.ep)
                                            : nullptr;
    rg::ProfScope prof(rg::FAM_CONV_DGRAD, stream, flops, ALG_BYTES);
    const dim3 grid(p.m_tiles * nt_max, pl.splits, SH * SW);
    static const int dma_env = getenv("RG_CONV_DMA") ? atoi(getenv("RG_CONV_DMA")) : 1;
    const TuneKey tk = {2, N, C, H, W, K, KH, KW, SH, SW, PH, PW, mode, pl.tile, pl.splits,
                        (int)(p.ep.res != nullptr) * 16 + (int)(p.ep.mask != nullptr) * 8 + (int)(p.ep.rowsum != nullptr) * 4 + p.ep.act};
    // (the 64 x 128 eight-wave form has four wave columns: not with fused row sums, whose column count the caller sized for two)
    choose_impl(2, tk, stream, (pl.tile == 0 || (pl.tile == 1 && !p.ep.rowsum)) ? 3 : 2, [&](int impl) {
        if (impl == 2) {
            if (pl.tile == 0) { RG_DGRAD_PL_LAUNCH(128, 128, 4, 2); }
            else { RG_DGRAD_PL_LAUNCH(64, 128, 2, 4); }
        } else if (impl) {
            RG_TILE_SWITCH(pl.tile, RG_DGRAD_PL_LAUNCH);
        } else if (mode == 2 && dma_env && (pl.tile == 0 || pl.tile == 1) && C % 4 == 0) {
            // 1x1 / stride 1: both operands are lane-linear in memory -> LDS-DMA ring (conv1x1_dma_kernel)
            if (pl.tile == 0) hipLaunchKernelGGL((conv1x1_dma_kernel<128>), grid, dim3(NT), 0, stream, dp);
            else hipLaunchKernelGGL((conv1x1_dma_kernel<64>), grid, dim3(NT), 0, stream, dp);
        } else {
            RG_TILE_SWITCH(pl.tile, RG_DGRAD_LAUNCH);
        }
    });
    if (pl.splits > 1 && !p.arrive) {
        if (int e = rg::check_launch("rg_conv2d_dgrad")) return e;
        if (one_class) launch_finish(stream, p.partial, dx, p.M, (int)ng_max, H * W, make_fastdiv(H * W), pl.splits, p.ep);
        else hipLaunchKernelGGL(conv_splitk_finish_strided_kernel, dim3(finish_grid((int64_t)p.M * N * H * W)), dim3(256), 0, stream,
                                p.partial, dx, dp, pl.splits, make_fastdiv(H * W), make_fastdiv(W), make_fastdiv(N * H * W));
    }
    return rg::check_launch("rg_conv2d_dgrad");
    };
    if (nc > 1) {
        int st = RG_OK;
        const TuneKey pk = {128, N, C, H, W, K, KH, KW, SH, SW, PH, PW, mode, nc, rowsum ? rowsum_cols : 0,
                            (int)(residual != nullptr) * 16 + (int)(relu_mask != nullptr) * 8 + (int)(rowsum != nullptr) * 4 + act};
        choose_impl(0, pk, stream, nc, [&](int c) {
            const int e = run_plan(cands[c]);
            if (e) st = e;
        });
        return st;
    }
    return run_plan(cands[0]);
}
}  // namespace

extern "C" int rg_conv2d_dgrad(const float* dy, const float* w, const float* w_krsc, float* dx, int N, int C, int H,
                               int W, int K, int KH, int KW, int SH, int SW, int PH, int PW, int P, int Q,
                               const float* scale, const float* shift, const float* residual, int act, float slope,
                               const float* relu_mask, float* rowsum, int rowsum_cols, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    return dgrad_impl(dy, w, w_krsc, dx, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, scale, shift, residual, act, slope,
                      relu_mask, rowsum, rowsum_cols, workspace, workspace_bytes, stream, nullptr);
}

// Number of row-sum column blocks rg_conv2d_dgrad writes for this geometry when given the workspace of
// rg_conv2d_dgrad_workspace (0: the launch uses split-K or the small-C kernel, which do not produce row sums).
extern "C" int rg_conv2d_dgrad_rowsum_cols(int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH, int PW,
                                           int P, int Q) {
    int cols = 0;
    if (dgrad_impl(nullptr, nullptr, nullptr, nullptr, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, nullptr, nullptr, nullptr,
                   0, 0.f, nullptr, nullptr, 0, nullptr, 0, nullptr, &cols) != RG_OK)
        return 0;
    return cols;
}

extern "C" int rg_weights_to_krsc(const float* w, float* w_krsc, int K, int C, int KH, int KW, hipStream_t stream) {
    RG_REQUIRE(w && w_krsc && K > 0 && C > 0 && KH > 0 && KW > 0, "rg_weights_to_krsc: bad arguments");
    const int64_t total = (int64_t)K * C * KH * KW;
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(weights_to_krsc_kernel, dim3(finish_grid(total)), dim3(256), 0, stream, w, w_krsc, total, C,
                       KH * KW);
    return rg::check_launch("rg_weights_to_krsc");
}

extern "C" int rg_krsc_chunk(void) { return KRSC_CHUNK; }

// table: device memory, `count` entries of 6 int64 words {w, w_krsc, K, C, KH*KW, first block}; blocks of rg_krsc_chunk() elements
extern "C" int rg_weights_to_krsc_multi(const void* table, int count, int total_blocks, hipStream_t stream) {
    RG_REQUIRE(table && count > 0 && total_blocks > 0, "rg_weights_to_krsc_multi: bad arguments");
    rg::ProfScope prof(rg::FAM_CONV_FWD, stream, 0.0, 8.0 * (double)total_blocks * KRSC_CHUNK);
    hipLaunchKernelGGL(weights_to_krsc_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, static_cast<const long long*>(table),
                       count);
    return rg::check_launch("rg_weights_to_krsc_multi");
}

namespace {
struct WgradPlan {
    int tile, m_tiles, n_tiles, splits, ktiles_per_split;
};
static WgradPlan plan_wgrad(int M, int Ng, int64_t Kg) {
    WgradPlan pl;
    // the reduction (N*P*Q) is long, so parallelism comes from split-K: always take the largest tile that fits
    // (a 64x64 tile issues 2x the LDS reads and 4x the loader instructions per MFMA of the 128x128 one)
    pl.tile = (M <= 32) ? 3 : ((M <= 64 || Ng <= 64) ? 2 : 0);     // (64 x 128 for M = 64 measured 8-10 % slower than 64 x 64 here)
    pl.m_tiles = rg::cdiv(M, kTileBM[pl.tile]);
    pl.n_tiles = rg::cdiv(Ng, kTileBN[pl.tile]);
    const int64_t nk = rg::cdiv64(Kg, BK);
    const int64_t mn = (int64_t)pl.m_tiles * pl.n_tiles;
    // 1024 workgroups = one full round at the kernel's 4 workgroups per CU (measured: 512 / 768 / 896 / 1024 / 1152 / 1280 /
    // 1536 -> 71.9 / 75.2 / 75.7 / 78.5 / 75.3 / 77.7 / 76.9 TFLOP/s over the FD-GAN step's wgrad launches; the fwd cost
    // model over-splits here)
    static const int target_wg = getenv("RG_WGRAD_WG") ? atoi(getenv("RG_WGRAD_WG")) : 1024;
    int64_t want = rg::cdiv64(target_wg, mn);
    if (want > nk / 16) want = nk / 16;   // >= 16 k-tiles per split keeps partial traffic small
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    while (want > 1 && want * (int64_t)M * Ng * 4 >= (1ll << 31)) --want;
    static const int xcd_splits = getenv("RG_WGRAD_XCD") ? atoi(getenv("RG_WGRAD_XCD")) : 1;
    if (xcd_splits && want >= 8) {
        // a multiple of 8 splits (see the kernel's split -> XCD mapping); trailing splits may be empty (they store zeros)
        int64_t w8 = (want + 4) / 8 * 8;
        while (w8 > 8 && w8 * (int64_t)M * Ng * 4 >= (1ll << 31)) w8 -= 8;
        if (w8 * (int64_t)M * Ng * 4 < (1ll << 31) && w8 <= nk) {
            pl.ktiles_per_split = (int)rg::cdiv64(nk, w8);
            pl.splits = (int)w8;
            return pl;
        }
    }
    pl.ktiles_per_split = (int)rg::cdiv64(nk, want);
    pl.splits = (int)rg::cdiv64(nk, pl.ktiles_per_split);
    return pl;
}
}  // namespace

namespace {
// Measured split depth (choose_impl, like plan_candidates for the forward / data-gradient GEMMs): the model's split count first,
// then 1/2, 3/4, 3/2 and 2x of it (multiples of 8 from 8 up: the kernel's split -> XCD mapping), >= 8 k-tiles per split.  What is
// timed is the whole call: kernel + split-K reduction (+ folded-BatchNorm finish).
static int wgrad_plan_candidates(int M, int Ng, int64_t Kg, WgradPlan* out, int max_out) {
    out[0] = plan_wgrad(M, Ng, Kg);
    int n = 1;
    if (!plan_tune_enabled() || !tune_enabled() || getenv("RG_WGRAD_WG")) return n;
    if (2.0 * M * (double)Ng * (double)Kg < kPlanTuneMinFlop) return n;
    const int64_t nk = rg::cdiv64(Kg, BK);
    const int base = out[0].splits;
    const int alts[4] = {base / 2, base * 3 / 4, base * 3 / 2, base * 2};
    for (int i = 0; i < 4 && n < max_out; ++i) {
        int64_t sp = alts[i];
        if (sp >= 8) sp = (sp + 4) / 8 * 8;
        if (sp < 1) sp = 1;
        if (sp > 512) sp = 512;
        if (sp > 1 && nk / sp < 8) continue;
        if (sp * (int64_t)M * Ng * 4 >= (1ll << 31)) continue;
        WgradPlan pl = out[0];
        pl.ktiles_per_split = (int)rg::cdiv64(nk, sp);
        pl.splits = sp >= 8 ? (int)sp : (int)rg::cdiv64(nk, pl.ktiles_per_split);
        if (pl.splits > nk) continue;
        bool dup = false;
        for (int j = 0; j < n; ++j) dup = dup || out[j].splits == pl.splits;
        if (!dup) out[n++] = pl;
    }
    return n;
}
}  // namespace

extern "C" size_t rg_conv2d_wgrad_workspace(int N, int C, int K, int KH, int KW, int P, int Q) {
    if (thin_filter(K, KH, KW)) {
        const int64_t Ng = (int64_t)N * P * Q;
        return (size_t)rg::cdiv64(Ng, thin_wgrad_per_slice(C, Ng)) * (size_t)K * (size_t)C * KH * KW * sizeof(float);
    }
    WgradPlan cands[8];
    const int nc = wgrad_plan_candidates(K, C * KH * KW, (int64_t)N * P * Q, cands, 8);
    int smax = 0;
    for (int i = 0; i < nc; ++i) smax = cands[i].splits > smax ? cands[i].splits : smax;
    return (size_t)smax * (size_t)K * (size_t)C * KH * KW * sizeof(float);
}

extern "C" int rg_bn_fold_wgrad(const float* w, float* g, const float* scale, const float* invstd, const float* running_mean,
                                const float* sum_g, const float* partials, int n_slices, float* dbeta, float* dgamma, int K,
                                int M, hipStream_t stream);        // norm.hip

namespace {
struct FoldArgs {                 // folded frozen-statistics BatchNorm behind the convolution (rg_conv2d_wgrad_fold)
    const float *w, *scale, *invstd, *mean, *sum_g, *partials;
    int n_slices;
    float *dbeta, *dgamma;
};
static int fold_after(const FoldArgs* f, float* dw, int K, int M, hipStream_t stream) {
    return rg_bn_fold_wgrad(f->w, dw, f->scale, f->invstd, f->mean, f->sum_g, f->partials, f->n_slices, f->dbeta, f->dgamma, K, M,
                            stream);
}
static bool fold_fused_enabled() {
    static const int env = getenv("RG_WGRAD_FOLD_FUSED") ? atoi(getenv("RG_WGRAD_FOLD_FUSED")) : 1;
    return env != 0;
}

int wgrad_impl(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH, int KW, int SH, int SW, int PH,
               int PW, int P, int Q, const FoldArgs* fold, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    if (int e = validate("rg_conv2d_wgrad", N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q)) return e;
    RG_REQUIRE(x && dy && dw, "rg_conv2d_wgrad: null tensor");
    RG_REQUIRE(KH < 65536 && KW < 65536, "rg_conv2d_wgrad: filter too large");
    ConvP p;
    fill_common(p, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q);
    p.x = x; p.w = dy;
    p.ep = Epilogue{nullptr, nullptr, nullptr, 0, 0.f, nullptr, nullptr, 0};
    p.M = K; p.Ng = C * KH * KW; p.Kg = N * P * Q;
    if (thin_filter(K, KH, KW)) {
        const int per = thin_wgrad_per_slice(C, p.Kg);
        const int slices = rg::cdiv(p.Kg, per);
        const size_t need_thin = (size_t)slices * (size_t)K * (size_t)p.Ng * sizeof(float);
        if (workspace && need_thin <= workspace_bytes) {
            ThinP t;
            thin_fill(t, x, dy, static_cast<float*>(workspace), N, C, H, W, SH, SW, PH, PW, P, Q, per);
            rg::ProfScope prof(rg::FAM_CONV_WGRAD, stream, 2.0 * (double)K * p.Ng * p.Kg, ALG_BYTES);
            const dim3 grid(C, slices);
            if (thin_px4(KH, KW, SH, SW, PH, PW, W, Q, dy, dy) && ((P * Q) & 3) == 0)
                THIN_PX4_DISPATCH(conv_wgrad_k1_px4_kernel, grid, stream, t);
            else
                THIN_DISPATCH(conv_wgrad_k1_kernel, grid, stream, t);
            if (int e = rg::check_launch("rg_conv2d_wgrad(thin)")) return e;
            const int64_t n = (int64_t)K * p.Ng;
            launch_reduce(stream, static_cast<const float*>(workspace), dw, n, slices, 0, C, KH * KW);
            if (int e = rg::check_launch("rg_conv2d_wgrad(thin reduce)")) return e;
            return fold ? fold_after(fold, dw, K, p.Ng, stream) : RG_OK;
        }
    }
    WgradPlan cands[8];
    int nc = wgrad_plan_candidates(p.M, p.Ng, p.Kg, cands, 8);
    for (int i = 1; i < nc; ++i)                 // a caller with the model plan's scratch only: the model's plan only
        if (!workspace || (size_t)cands[i].splits * p.M * (size_t)p.Ng * sizeof(float) > workspace_bytes) nc = 1;
    const ConvP pw0 = p;
    auto run_plan = [&](const WgradPlan& pl) -> int {
    p = pw0;
    p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles;
    p.splits = pl.splits; p.ktiles_per_split = pl.ktiles_per_split;
    const size_t need = (size_t)pl.splits * p.M * (size_t)p.Ng * sizeof(float);      // see rg_conv2d_wgrad_workspace
    if (need > workspace_bytes || !workspace) {
        rg::set_error("rg_conv2d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
        return RG_ERR_WORKSPACE;
    }
    RG_REQUIRE(need < (1ull << 31), "rg_conv2d_wgrad: partial buffer exceeds 2 GiB");
    // (r,s)-major columns: always through the workspace (the finishing kernel restores the checkpoint order)
    static const int rsc_env = getenv("RG_WGRAD_RSC") ? atoi(getenv("RG_WGRAD_RSC")) : 0;
    const bool rsc = rsc_env && (KH * KW > 1) && (C % 16 == 0);
    const bool via_ws = pl.splits > 1 || rsc;
    p.a_vec4 = rsc ? 1 : 0;
    p.y = via_ws ? static_cast<float*>(workspace) : dw;
    p.x_bytes = (unsigned)((int64_t)N * C * H * W * 4);
    p.w_bytes = (unsigned)((int64_t)N * K * P * Q * 4);
    p.y_bytes = via_ws ? (unsigned)need : (unsigned)((int64_t)K * C * KH * KW * 4);
    const dim3 grid(p.m_tiles * p.n_tiles, 1, pl.splits);
    {
        rg::ProfScope prof(rg::FAM_CONV_WGRAD, stream, 2.0 * p.M * (double)p.Ng * p.Kg, ALG_BYTES);
        const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
        const bool veca = al && ((P * Q) % 4 == 0);
        bool vec = veca && KH == 1 && KW == 1 && SH == 1 && SW == 1 && PH == 0 && PW == 0;
        static const int shift_env = getenv("RG_WGRAD_SHIFT") ? atoi(getenv("RG_WGRAD_SHIFT")) : 1;
        if (!vec && !rsc && shift_env && veca && SH == 1 && SW == 1 && PW <= 1 && KW <= PW + 2 && Q % 4 == 0 && W >= 4 && KW * KH > 1) {
            vec = true;                  // im2col operand = shifted float4 loads of x (conv_wgrad_kernel, p.wshift)
            p.wshift = 1;
        }
#define RG_WGRAD_LAUNCH(BM_, BN_, WM_, WN_)                                                                        \
    if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, WM_, WN_, true, true>), grid, dim3(NT), 0, stream, p); \
    else if (veca) hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, WM_, WN_, false, true>), grid, dim3(NT), 0, stream, p); \
    else hipLaunchKernelGGL((conv_wgrad_kernel<BM_, BN_, WM_, WN_, false, false>), grid, dim3(NT), 0, stream, p)
#define RG_WGRAD_PL_LAUNCH(BM_, BN_, WM_, WN_)                                                                        \
    if (vec) hipLaunchKernelGGL((conv_wgrad_pl_kernel<BM_, BN_, WM_, WN_, true, true>), grid, dim3(64 * WM_ * WN_), 0, stream, p); \
    else if (veca) hipLaunchKernelGGL((conv_wgrad_pl_kernel<BM_, BN_, WM_, WN_, false, true>), grid, dim3(64 * WM_ * WN_), 0, stream, p); \
    else hipLaunchKernelGGL((conv_wgrad_pl_kernel<BM_, BN_, WM_, WN_, false, false>), grid, dim3(64 * WM_ * WN_), 0, stream, p)
        const TuneKey tk = {4, N, C, H, W, K, KH, KW, SH, SW, PH, PW, (vec ? 2 : 0) + (veca ? 1 : 0) + p.wshift * 4, pl.tile, pl.splits, 0};
        choose_impl(4, tk, stream, pl.tile == 0 ? 3 : 2, [&](int impl) {
            if (impl == 2) {
                RG_WGRAD_PL_LAUNCH(128, 128, 4, 2);
            } else if (impl) {
                switch (pl.tile) {
                    case 0: RG_WGRAD_PL_LAUNCH(128, 128, 2, 2); break;
                    case 2: RG_WGRAD_PL_LAUNCH(64, 64, 2, 2); break;
                    default: RG_WGRAD_PL_LAUNCH(32, 256, 1, 4); break;
                }
            } else {
                switch (pl.tile) {
                    case 0: RG_WGRAD_LAUNCH(128, 128, 2, 2); break;
                    case 2: RG_WGRAD_LAUNCH(64, 64, 2, 2); break;
                    default: RG_WGRAD_LAUNCH(32, 256, 1, 4); break;
                }
            }
        });
#undef RG_WGRAD_PL_LAUNCH
#undef RG_WGRAD_LAUNCH
        if (int e = rg::check_launch("rg_conv2d_wgrad")) return e;
        if (via_ws) {
            const int64_t n = (int64_t)p.M * p.Ng;
            const bool al16 = ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(dw) |
                                reinterpret_cast<uintptr_t>(fold ? fold->w : nullptr)) & 15) == 0;
            if (fold && !rsc && (p.Ng & 3) == 0 && al16 && fold_fused_enabled()) {
                // reduction + BatchNorm-fold finish in one launch (one workgroup per filter)
                hipLaunchKernelGGL(splitk_reduce_fold_kernel, dim3(p.M), dim3(256), 0, stream, static_cast<const float*>(workspace), dw,
                                   fold->w, p.Ng, n, pl.splits, fold->scale, fold->invstd, fold->mean, fold->sum_g, fold->partials,
                                   fold->n_slices, fold->dbeta, fold->dgamma);
                return rg::check_launch("rg_conv2d_wgrad(reduce + fold)");
            }
            launch_reduce(stream, static_cast<const float*>(workspace), dw, n, pl.splits, rsc ? 1 : 0, C, KH * KW);
        }
    }
    if (int e = rg::check_launch("rg_conv2d_wgrad(reduce)")) return e;
    return fold ? fold_after(fold, dw, K, C * KH * KW, stream) : RG_OK;
    };
    if (nc > 1) {
        int st = RG_OK;
        const TuneKey pk = {256, N, C, H, W, K, KH, KW, SH, SW, PH, PW, fold ? 1 : 0, nc, cands[0].splits, 0};
        choose_impl(0, pk, stream, nc, [&](int c) {
            const int e = run_plan(cands[c]);
            if (e) st = e;
        });
        return st;
    }
    return run_plan(cands[0]);
}
}  // namespace

extern "C" int rg_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH,
                               int KW, int SH, int SW, int PH, int PW, int P, int Q, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    return wgrad_impl(x, dy, dw, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, nullptr, workspace, workspace_bytes, stream);
}

// Weight gradient of a convolution with a folded frozen-statistics BatchNorm behind it, finished in the same call:
// G = wgrad(x, dy); dgamma = invstd (sum_m w G - mean sum_g); dbeta = sum_g (when it comes as slice partials); dw = scale G
// (arguments as rg_bn_fold_wgrad; dgamma / dbeta may be NULL).  With split-K the finish runs inside the reduction launch.
extern "C" int rg_conv2d_wgrad_fold(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int K, int KH, int KW,
                                    int SH, int SW, int PH, int PW, int P, int Q, const float* w, const float* scale,
                                    const float* invstd, const float* running_mean, const float* sum_g, const float* partials,
                                    int n_slices, float* dbeta, float* dgamma, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    RG_REQUIRE(w && scale, "rg_conv2d_wgrad_fold: null filters / scale");
    RG_REQUIRE(!dgamma || (invstd && running_mean && (sum_g || partials)), "rg_conv2d_wgrad_fold: dgamma needs invstd, mean and the sums");
    RG_REQUIRE(!partials || n_slices > 0, "rg_conv2d_wgrad_fold: partials need their slice count");
    const FoldArgs f{w, scale, invstd, running_mean, sum_g, partials, n_slices, dbeta, dgamma};
    return wgrad_impl(x, dy, dw, N, C, H, W, K, KH, KW, SH, SW, PH, PW, P, Q, &f, workspace, workspace_bytes, stream);
}
