// bf16-plane operand tiles for the split-bf16 convolution kernels (included by conv_igemm.hip inside its anonymous namespace).
//
// Round-3 kernels kept the LDS operand tiles in fp32 and every wave split every fragment it read into its three bf16 pieces
// (9 VALU instructions per element pair, per reading wave): 230 VALU instructions beside 24 MFMAs per 16-deep k-tile, VALU-bound.
// Here an operand element is split ONCE, by the thread that staged it, on its way from the staging registers into LDS; the LDS
// tile holds three bf16 images (hi / mid / lo) and the waves read finished MFMA fragments:
//   PL_R  [piece][k half][row][8 k]    16 B per (row, k half): one ds_read_b128 per piece and 32 x 16 block.  For operands whose
//                                      staging thread holds consecutive k of one row: float4 along k (filters [K][RS][C] in the
//                                      forward pass, both weight-gradient operands) or a scalar gather with a blocked k
//                                      assignment (thread = one column, 8 consecutive k: im2col pixels, strided gradients);
//   PL_T  [piece][k][row]  (k-major)   for operands whose float4 runs along the GEMM row (NCHW pixels of a 1x1 layer, [K][RS][C]
//                                      filters in the data gradient): the thread writes 4 rows of one k (8 B per piece) and the
//                                      fragment is read with two ds_read_b64_tr_b16 (gfx950 transposing read: the 16 lanes of a
//                                      group receive 4 consecutive k of their own row from a 4 k x 16 row block).
// Bank behaviour (MI355X_MICROARCH.md LDS table): PL_R reads are 512 contiguous bytes per 32-lane half; PL_T pitches are
// = 64 or 192 (mod 256) bytes so the four k rows of a transposed read fall on disjoint bank quarters; the 8-byte stores of a
// 16-lane group are 128 contiguous bytes (PL_T) or two 64-byte runs 16 banks apart (PL_R, through the +64 B pad of a k half).
#pragma once
#include "conv_planes_core.h"

// ---------------------------------------------------------------------------------------------
// forward.  Same loaders, split-K and epilogues as conv_fwd_kernel; BMODE 0 generic gather (order (c, r, s)), 1: (r, s)-major
// over the [K][RS][C] filter copy, 2: 1x1 / stride 1 / pad 0 with float4 pixel loads.
// Filters -> PL_R (float4 along k, or a scalar chunk of consecutive k per thread); pixels -> PL_T (BMODE 2: float4 along the
// pixels of one channel) or PL_R (gathers: thread = one output pixel, EB consecutive reduction indices; every load instruction
// still has one wave-uniform k and lanes along the pixels).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BMODE, bool AVEC>
__global__ __launch_bounds__(64 * WM * WN) void conv_fwd_pl_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int NTH = 64 * WM * WN;                       // 4 waves (the tiles of RG_TILE_SWITCH) or 8 (128 x 128 as 4 x 2 waves)
    static_assert(BN >= 64, "the gather loader needs a wave-uniform k");
    using LA = PlTile<PL_R, BM>;
    using LB = PlTile<BMODE == 2 ? PL_T : PL_R, BN>;
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILEB];
    const unsigned lds0 = (unsigned)(size_t)((lds_u8_t*)lds);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = tile % p.m_tiles, nt = tile / p.m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;
    const int HW = p.H * p.W;
    const int RS = p.KH * p.KW;
    const rsrc_t rw = make_rsrc(p.w, p.w_bytes), rx = make_rsrc(p.x, p.x_bytes);

    // ---- A operand (filters) ----
    constexpr int NAV = (BM * 4 + NTH - 1) / NTH;            // AVEC: float4 per thread and tile
    constexpr bool AFULL = BM * 4 >= NTH;                   // every thread stages a float4
    constexpr int EA = BM * BK / NTH;                       // scalar: consecutive k per thread (thread = row tid % BM, chunk tid / BM)
    static_assert(AVEC || EA >= 2, "scalar filter chunk of at least one pair");
    constexpr int NUA = AVEC ? NAV : 1, SPA = AVEC ? 2 : EA / 2;      // staged units and element pairs per unit
    unsigned aoff[NUA];                                    // global byte offset at k-tile 0, or OOB
    unsigned awr[NUA];                                     // LDS byte address (piece 0, buffer 0) of the unit's store
    int akq[NUA];
    bool aact = true;
    if (AVEC) {
#pragma unroll
        for (int i = 0; i < NAV; ++i) {
            const int v = tid + NTH * i;
            const int row = v >> 2;
            akq[i] = (v & 3) * 4;
            aact = AFULL || v < BM * 4;
            aoff[i] = (aact && m0 + row < p.M) ? (unsigned)(((int64_t)(m0 + row) * p.Kg + akq[i]) * 4) : OOB;
            awr[i] = lds0 + LA::off_rk(aact ? row : 0, akq[i]);
        }
    } else {
        const int row = tid % BM, kc = tid / BM;
        akq[0] = kc * EA;
        aoff[0] = (m0 + row < p.M) ? (unsigned)(((int64_t)(m0 + row) * p.Kg + akq[0]) * 4) : OOB;
        awr[0] = lds0 + LA::off_rk(row, akq[0]);
    }

    // ---- B operand (pixels) ----
    constexpr int BV = BN / 4;
    constexpr int BVSTEP = NTH / BV;
    constexpr int BVCNT = BV * BK / NTH;
    static_assert(BV * BK % NTH == 0, "whole float4 passes");
    constexpr int EB = BN * BK / NTH;                       // gather: consecutive k per thread (thread = column tid % BN, chunk tid / BN)
    constexpr int EBU = EB > 8 ? 8 : EB;                   // .. per staged unit (one 16-byte store per piece at most)
    constexpr int NUB = BMODE == 2 ? BVCNT : EB / EBU, SPB = BMODE == 2 ? 2 : EBU / 2;
    const int vcol = tid % BV, vrow0 = tid / BV;
    const int bcol = tid % BN;
    const int bkc = __builtin_amdgcn_readfirstlane(tid / BN);
    bool bvalid;
    int h0 = 0, w0 = 0, pixb = 0;
    unsigned bvoff = OOB;
    unsigned bwr[NUB];
    if (BMODE == 2) {
        const int n = n0 + 4 * vcol;
        bvalid = n < p.Ng;
        if (bvalid) {
            const int img = fdiv(n, p.d_pq);
            bvoff = (unsigned)((((int64_t)img * p.C + vrow0) * HW + (n - img * HW)) * 4);
        }
#pragma unroll
        for (int u = 0; u < NUB; ++u) bwr[u] = lds0 + LA::BYTES + LB::off_rk(4 * vcol, vrow0 + u * BVSTEP);
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
#pragma unroll
        for (int u = 0; u < NUB; ++u) bwr[u] = lds0 + LA::BYTES + LB::off_rk(bcol, bkc * EB + u * EBU);
    }

    float ra[AVEC ? 4 * NAV : EA];
    float rb[BMODE == 2 ? 4 * BVCNT : EB];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        const unsigned kb4 = (unsigned)kbase * 4u;
        const bool ktail = kbase + BK > p.Kg;                    // uniform; only the last tile of ragged Kg
        if (AVEC) {
#pragma unroll
            for (int i = 0; i < NAV; ++i) {
                unsigned o = aoff[i] + kb4;
                if (ktail && kbase + akq[i] >= p.Kg) o = OOB;
                const float4 t = bload4(rw, o);
                ra[4 * i + 0] = t.x; ra[4 * i + 1] = t.y; ra[4 * i + 2] = t.z; ra[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                unsigned o = aoff[0] + kb4 + 4u * e;
                if (ktail && kbase + akq[0] + e >= p.Kg) o = OOB;
                ra[e] = bload(rw, o);
            }
        }
        if (BMODE == 2) {
            const unsigned kstride = (unsigned)HW * 4u;
#pragma unroll
            for (int i = 0; i < BVCNT; ++i) {
                unsigned o = bvoff + (unsigned)(kbase + i * BVSTEP) * kstride;
                if (ktail && kbase + vrow0 + i * BVSTEP >= p.Kg) o = OOB;
                const float4 t = bload4(rx, o);
                rb[4 * i + 0] = t.x; rb[4 * i + 1] = t.y; rb[4 * i + 2] = t.z; rb[4 * i + 3] = t.w;
            }
        } else if (BMODE == 1) {
            const int rs = fdiv(kbase, p.d_c);                   // scalar: the whole tile shares (r, s)
            const int c0 = kbase - rs * p.C;
            const int r = fdiv(rs, p.d_kw);
            const int s = rs - r * p.KW;
            const int h = h0 + r, w = w0 + s;
            const bool ok = bvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            const unsigned o0 = ok ? (unsigned)(pixb + r * p.W + s + (c0 + bkc * EB) * HW) * 4u : OOB;
            const unsigned cstride = (unsigned)HW * 4u;
#pragma unroll
            for (int e = 0; e < EB; ++e) rb[e] = bload(rx, o0 + (unsigned)e * cstride);
        } else {
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = kbase + bkc * EB + e;              // wave-uniform -> scalar unit
                const int c = fdiv(k, p.d_rs);
                const int rs = k - c * RS;
                const int r = fdiv(rs, p.d_kw);
                const int s = rs - r * p.KW;
                const int h = h0 + r, w = w0 + s;
                const bool ok = bvalid && k < p.Kg && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                rb[e] = bload(rx, ok ? (unsigned)(pixb + c * HW + r * p.W + s) * 4u : OOB);
            }
        }
    };

    // ---- staging micro-steps (two per element pair); a unit's pieces are written behind its last pair ----
    constexpr int SA = 2 * NUA * SPA, S = SA + 2 * NUB * SPB;
    PlChunk<SPA> ca[NUA];
    PlChunk<SPB> cb[NUB];
    auto stage_step = [&](int wbuf, int s) {                  // wbuf: byte offset of the buffer being filled
        if (s < SA) {
            const int pr = s >> 1, u = pr / SPA, j = pr % SPA;
            if (!(s & 1)) {
                ca[u].step_a(j, ra[u * 2 * SPA + 2 * j], ra[u * 2 * SPA + 2 * j + 1]);
            } else {
                ca[u].step_b(j);
                if (j == SPA - 1) {
                    if (AFULL || !AVEC) ca[u].write(awr[u] + wbuf, LA::PIECE);
                    else if (aact) ca[u].write(awr[u] + wbuf, LA::PIECE);
                }
            }
        } else {
            const int pr = (s - SA) >> 1, u = pr / SPB, j = pr % SPB;
            if (!(s & 1)) {
                cb[u].step_a(j, rb[u * 2 * SPB + 2 * j], rb[u * 2 * SPB + 2 * j + 1]);
            } else {
                cb[u].step_b(j);
                if (j == SPB - 1) cb[u].write(bwr[u] + wbuf, LB::PIECE);
            }
        }
    };

    const unsigned fa = lds0 + LA::frag_base(lane, wm * T::WTM), fb = lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN);
    constexpr int NM = 6 * T::TM * T::TN;

    const int nk = (p.Kg + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk) kt_end = nk;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
#pragma unroll
        for (int s = 0; s < S; ++s) stage_step(0, s);
    }
    __syncthreads();
    pl_kloop(kt_begin, kt_end, load_tile, [&](auto cur_tag, auto stage_tag) {
        constexpr int RD = decltype(cur_tag)::value ? TILEB : 0, WR = TILEB - RD;
        constexpr bool STAGE = decltype(stage_tag)::value;
        mma_pl<T::TM, T::TN>([&](int i, int pc) { return LA::frag(fa + RD + pc * LA::PIECE + i * LA::BLK); },
                             [&](int j, int pc) { return LB::frag(fb + RD + pc * LB::PIECE + j * LB::BLK); }, acc,
                             [&](int slot) {
                                 if (STAGE) {
#pragma unroll
                                     for (int s = pl_first_step<S, NM>(slot); s < pl_end_step<S, NM>(slot); ++s)
                                         if (s >= 0) stage_step(WR, s);
                                 }
                             });
    });
    store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, p.Ng, p.P * p.Q, p.d_pq, split);
}

// ---------------------------------------------------------------------------------------------
// data gradient (and the forward of ConvTranspose2d): conv_dgrad_kernel's loaders, classes, split-K and epilogues.
// MODE 0: filters [K][C][KH][KW] by scalar loads (thread = input channel, EA consecutive reduction indices) -> PL_R;
// MODE 1 / 2: filters [K][RS][C], float4 along the input channels (GEMM rows) -> PL_T.
// Gradients: MODE 2 float4 along the pixels -> PL_T; MODE 0 / 1 gather (thread = one class pixel, EB consecutive reduction
// indices, each load instruction one wave-uniform k with lanes along the pixels) -> PL_R.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void conv_dgrad_pl_kernel(const DgradP dp) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int NTH = 64 * WM * WN;                       // 4 waves (the tiles of RG_TILE_SWITCH) or 8 (128 x 128 as 4 x 2 waves)
    static_assert(BN >= 64, "the gather loader needs a wave-uniform k");
    using LA = PlTile<MODE == 0 ? PL_R : PL_T, BM>;
    using LB = PlTile<MODE == 2 ? PL_T : PL_R, BN>;
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILEB];
    const unsigned lds0 = (unsigned)(size_t)((lds_u8_t*)lds);
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
    constexpr int BV = BN / 4, BVSTEP = NTH / BV, BVCNT = BV * BK / NTH;
    static_assert(BV * BK % NTH == 0, "whole float4 passes");
    constexpr int EB = BN * BK / NTH, EBU = EB > 8 ? 8 : EB;
    constexpr int NUB = MODE == 2 ? BVCNT : EB / EBU;
    PlStager<NUB, MODE == 2 ? 2 : EBU / 2> sb;
    const int vcol = tid % BV, vrow0 = tid / BV;
    const int bcol = tid % BN;
    const int bkc = __builtin_amdgcn_readfirstlane(tid / BN);
    bool bvalid;
    int hb = 0, wb = 0, imgb = 0;
    unsigned bvoff = OOB;
    if constexpr (MODE == 2) {
        const int n = n0 + 4 * vcol;
        bvalid = n < cl.Ngc;
        if (bvalid) {
            const int img = fdiv(n, cl.d_hw);
            bvoff = (unsigned)((((int64_t)img * p.K + vrow0) * PQ + (n - img * PQ)) * 4);
        }
#pragma unroll
        for (int u = 0; u < NUB; ++u) sb.wr[u] = lds0 + LA::BYTES + LB::off_rk(4 * vcol, vrow0 + u * BVSTEP);
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
#pragma unroll
        for (int u = 0; u < NUB; ++u) sb.wr[u] = lds0 + LA::BYTES + LB::off_rk(bcol, bkc * EB + u * EBU);
    }

    // ---- A operand (filters), GEMM row m = input channel c ----
    constexpr int EA = BM * BK / NTH;                       // MODE 0: consecutive k per thread (thread = row tid % BM, chunk tid / BM)
    static_assert(MODE != 0 || EA >= 2, "scalar filter chunk of at least one pair");
    constexpr int AV = BM / 4, AVSTEP = NTH / AV, AVCNT = (AV * BK + NTH - 1) / NTH;
    constexpr bool AFULL = AV * BK >= NTH;                  // every thread stages a float4 (BM >= 64)
    constexpr int NUA = MODE == 0 ? 1 : AVCNT;
    PlStager<NUA, MODE == 0 ? EA / 2 : 2> sa;
    const int arow = tid % BM, akc = tid / BM;
    const int avcol = tid % AV, avrow0 = tid / AV;
    const bool aact = AFULL || avrow0 < BK;
    // MODE 1/2: byte offset of (row k' = avrow0, m) inside one tap block of the [K][RS][C] tensor, or OOB
    const unsigned avoff = (MODE != 0 && aact && m0 + 4 * avcol < p.M) ? (unsigned)(((int64_t)avrow0 * RS * p.C + m0 + 4 * avcol) * 4) : OOB;
    if constexpr (MODE == 0) {
        sa.wr[0] = lds0 + LA::off_rk(arow, akc * EA);
    } else {
#pragma unroll
        for (int u = 0; u < NUA; ++u) sa.wr[u] = lds0 + LA::off_rk(4 * avcol, aact ? avrow0 + u * AVSTEP : 0);
    }

    float ra[MODE == 0 ? EA : 4 * AVCNT];
    float rb[MODE == 2 ? 4 * BVCNT : EB];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        const bool ktail = kbase + BK > cl.Kgc;
        if constexpr (MODE == 0) {
            const int am = m0 + arow;
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int k = kbase + akc * EA + e;
                const int ko = fdiv(k, cl.d_taps);
                const int t = k - ko * taps;
                const int j = fdiv(t, cl.d_nrw);
                const int jj = t - j * cl.nrw;
                const int r = cl.r0 + p.SH * j, s = cl.s0 + p.SW * jj;
                const bool ok = am < p.M && k < cl.Kgc;
                ra[e] = bload(rw, ok ? (unsigned)((((int64_t)ko * p.C + am) * RS + r * p.KW + s) * 4) : OOB);
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int k = kbase + bkc * EB + e;              // wave-uniform -> scalar unit
                const int ko = fdiv(k, cl.d_taps);
                const int t = k - ko * taps;
                const int j = fdiv(t, cl.d_nrw);
                const int jj = t - j * cl.nrw;
                const int pp = hb - j, qq = wb - jj;
                const bool ok = bvalid && k < cl.Kgc && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
                rb[e] = bload(rdy, ok ? (unsigned)(imgb + ko * PQ + pp * p.Q + qq) * 4u : OOB);
            }
        } else {
            // tap-major order: the whole tile shares one filter tap (scalar decode)
            const int tap = fdiv(kbase, p.d_k);
            const int ko0 = kbase - tap * p.K;
            const int j = fdiv(tap, cl.d_nrw);
            const int jj = tap - j * cl.nrw;
            const int rs = (cl.r0 + p.SH * j) * p.KW + cl.s0 + p.SW * jj;
            const unsigned tbase = (unsigned)(((int64_t)ko0 * RS + rs) * p.C * 4);
            const unsigned akstride = (unsigned)(AVSTEP * RS * p.C) * 4u;
#pragma unroll
            for (int i = 0; i < AVCNT; ++i) {
                unsigned o = avoff + tbase + (unsigned)i * akstride;          // (an out-of-range offset stays out of range)
                if (ktail && kbase + avrow0 + i * AVSTEP >= cl.Kgc) o = OOB;
                pl_unpack4(&ra[4 * i], bload4(rw, o));
            }
            if constexpr (MODE == 2) {
                const unsigned kstride = (unsigned)PQ * 4u;
#pragma unroll
                for (int i = 0; i < BVCNT; ++i) {
                    unsigned o = bvoff + (unsigned)(kbase + i * BVSTEP) * kstride;
                    if (ktail && kbase + vrow0 + i * BVSTEP >= cl.Kgc) o = OOB;
                    pl_unpack4(&rb[4 * i], bload4(rdy, o));
                }
            } else {
                const int pp = hb - j, qq = wb - jj;
                const bool ok = bvalid && (unsigned)pp < (unsigned)p.P && (unsigned)qq < (unsigned)p.Q;
                const unsigned o0 = ok ? (unsigned)(imgb + (ko0 + bkc * EB) * PQ + pp * p.Q + qq) * 4u : OOB;
                const unsigned kstride = (unsigned)PQ * 4u;
#pragma unroll
                for (int e = 0; e < EB; ++e) rb[e] = bload(rdy, o0 + (unsigned)e * kstride);
            }
        }
    };

    constexpr int SA = decltype(sa)::STEPS, S = SA + decltype(sb)::STEPS;
    const unsigned fa = lds0 + LA::frag_base(lane, wm * T::WTM), fb = lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN);
    const int nk = (cl.Kgc + BK - 1) / BK;
    const int kt_begin = split * cl.ktps;
    int kt_end = kt_begin + cl.ktps;
    if (kt_end > nk) kt_end = nk;
    pl_mainloop<T, LA, LB, S>(fa, fb, kt_begin, kt_end, acc, load_tile, [&](unsigned wbuf, int s) {
        if (s < SA) {
            if constexpr (AFULL || MODE == 0) sa.step(s, ra, wbuf, LA::PIECE);
            else sa.step(s, ra, wbuf, LA::PIECE, aact);
        } else {
            sb.step(s - SA, rb, wbuf, LB::PIECE);
        }
    });

    if (p.SH == 1 && p.SW == 1) {       // one class: output pixels are contiguous, shared epilogue (+ split-K)
        store_tile_nchw<T>(p, acc, m0, n0, wm, wn, lane, cl.Ngc, p.H * p.W, cl.d_hw, split, (cl.poff + nt) * WN + wn);
        return;
    }
    if (p.partial) {                    // strided split-K (conv_splitk_finish_strided_kernel)
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
// weight gradient: conv_wgrad_kernel's split mapping, loaders and partial-tile store.  The reduction runs over output pixels, so
// both operands are "k-contiguous" where they vectorise (dy rows; x rows of a 1x1 layer or shifted float4s of a stride-1 filter
// tap): float4 along k -> PL_R.  The im2col gather (strided / wide filters) and ragged dy rows use the blocked scalar form:
// thread = one GEMM row / column, E consecutive output pixels (the pixel decode is wave-uniform: scalar unit).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool VEC, bool VECA>
__global__ __launch_bounds__(64 * WM * WN) void conv_wgrad_pl_kernel(const ConvP p) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int NTH = 64 * WM * WN;                       // 4 waves (the tiles of RG_TILE_SWITCH) or 8 (128 x 128 as 4 x 2 waves)
    static_assert(BN >= 64, "the gather loader needs a wave-uniform pixel chunk");
    using LA = PlTile<PL_R, BM>;
    using LB = PlTile<PL_R, BN>;
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILEB];
    const unsigned lds0 = (unsigned)(size_t)((lds_u8_t*)lds);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // split -> XCD mapping of conv_wgrad_kernel
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
    const int PQ = p.P * p.Q, HW = p.H * p.W, RS = p.KH * p.KW;
    const bool rsc = p.a_vec4 != 0;
    const bool nopad = p.PH == 0 && p.PW == 0 && (p.P - 1) * p.SH + p.KH <= p.H && (p.Q - 1) * p.SW + p.KW <= p.W;

    // ---- A operand (dy): rows = output channels ----
    constexpr int AVN = (BM * 4 + NTH - 1) / NTH;            // VECA: float4 (4 consecutive pixels of one row) per thread and tile
    constexpr bool AFULL = BM * 4 >= NTH;
    constexpr int EA = BM * BK / NTH;                       // scalar: consecutive pixels per thread (thread = row tid % BM, chunk tid / BM)
    static_assert(VECA || EA >= 2, "scalar chunk of at least one pair");
    PlStager<VECA ? AVN : 1, VECA ? 2 : EA / 2> sa;
    const int vrow = tid >> 2, vkq = (tid & 3) * 4;
    const bool aact = AFULL || vrow < BM;
    unsigned avoff[VECA ? AVN : 1];
    const int arow = tid % BM, akc = tid / BM;
    if constexpr (VECA) {
#pragma unroll
        for (int i = 0; i < AVN; ++i) {
            const int m = m0 + vrow + (NTH / 4) * i;
            avoff[i] = (aact && m < p.M) ? (unsigned)m * (unsigned)PQ * 4u : OOB;
            sa.wr[i] = lds0 + LA::off_rk(aact ? vrow + (NTH / 4) * i : 0, vkq);
        }
    } else {
        avoff[0] = (m0 + arow < p.M) ? (unsigned)(m0 + arow) * (unsigned)PQ * 4u : OOB;
        sa.wr[0] = lds0 + LA::off_rk(arow, akc * EA);
    }

    // ---- B operand (im2col of x, transposed): columns = (c, r, s) ----
    constexpr int BVN = (BN * 4 + NTH - 1) / NTH;            // VEC: float4 per thread and tile
    static_assert(BN * 4 >= NTH, "every thread stages a column float4");
    constexpr int EB = BN * BK / NTH, EBU = EB > 8 ? 8 : EB;
    constexpr int NUB = VEC ? BVN : EB / EBU;
    PlStager<NUB, VEC ? 2 : EBU / 2> sb;
    unsigned bvoff[VEC ? BVN : 1];
    int vrr[VEC ? BVN : 1], vss[VEC ? BVN : 1];           // p.wshift: tap offset (r - PH, s - PW) of the row's column n = (c, r, s)
    const int bcol = tid % BN;
    const int bkc = __builtin_amdgcn_readfirstlane(tid / BN);
    int coff = 0, cr = 0, cs = 0;                          // gather: element offset c*HW + r*W + s and tap of this thread's column
    bool cvalid = false;
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < BVN; ++i) {
            const int n = n0 + vrow + (NTH / 4) * i;
            int c = n;
            vrr[i] = vss[i] = 0;
            if (p.wshift) {
                c = fdiv(n, p.d_rs);
                const int rs = n - c * RS;
                const int r = fdiv(rs, p.d_kw);
                vrr[i] = r - p.PH;
                vss[i] = rs - r * p.KW - p.PW;
            }
            bvoff[i] = n < p.Ng ? (unsigned)c * (unsigned)HW * 4u : OOB;
            sb.wr[i] = lds0 + LA::BYTES + LB::off_rk(vrow + (NTH / 4) * i, vkq);
        }
    } else {
        const int n = n0 + bcol;
        cvalid = n < p.Ng;
        if (cvalid) {
            int c, rs;
            if (rsc) {
                rs = fdiv(n, p.d_c);
                c = n - rs * p.C;
            } else {
                c = fdiv(n, p.d_rs);
                rs = n - c * RS;
            }
            cr = fdiv(rs, p.d_kw);
            cs = rs - cr * p.KW;
            coff = c * HW + cr * p.W + cs;
        }
#pragma unroll
        for (int u = 0; u < NUB; ++u) sb.wr[u] = lds0 + LA::BYTES + LB::off_rk(bcol, bkc * EB + u * EBU);
    }

    float ra[VECA ? 4 * AVN : EA];
    float rb[VEC ? 4 * BVN : EB];
    floatx16 acc[T::TM][T::TN];
    zero_acc<T>(acc);

    auto load_tile = [&](int kt) {
        // pixel quad of the float4 loaders
        int img4 = 0, pq4 = 0;
        bool g4valid = false;
        if constexpr (VEC || VECA) {
            const int g = kt * BK + vkq;
            g4valid = g < p.Kg;
            img4 = g4valid ? fdiv(g, p.d_pq) : 0;
            pq4 = g - img4 * PQ;
        }
        if constexpr (VECA) {
            const unsigned ab = g4valid ? (unsigned)(img4 * p.K * PQ + pq4) * 4u : OOB;
#pragma unroll
            for (int i = 0; i < AVN; ++i) pl_unpack4(&ra[4 * i], bload4(rdy, ((ab | avoff[i]) & OOB) ? OOB : ab + avoff[i]));
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int g = kt * BK + akc * EA + e;
                const bool ok = g < p.Kg && avoff[0] != OOB;
                const int img = fdiv(ok ? g : 0, p.d_pq);
                ra[e] = bload(rdy, ok ? (unsigned)(img * p.K * PQ + (g - img * PQ)) * 4u + avoff[0] : OOB);
            }
        }
        if constexpr (VEC) {
            if (p.wshift) {
                // stride-1 filter tap (r, s): the four output pixels (pp, q0 .. q0 + 3) read x at (pp + r - PH, q0 + s - PW ..), four
                // CONSECUTIVE floats; a column that falls off a row end: load moved one element inwards, vector shifted
                const int pp = fdiv(pq4, p.d_q);
                const int q0 = pq4 - pp * p.Q;
#pragma unroll
                for (int i = 0; i < BVN; ++i) {
                    const int hh = pp + vrr[i], wb = q0 + vss[i];
                    const bool ok = g4valid && bvoff[i] != OOB && (unsigned)hh < (unsigned)p.H;
                    const bool neg = wb < 0, over = wb + 3 >= p.W;
                    const int e = img4 * p.C * HW + hh * p.W + wb + (neg ? 1 : 0) - (over ? 1 : 0);
                    const float4 t = bload4(rx, ok ? (unsigned)e * 4u + bvoff[i] : OOB);
                    rb[4 * i + 0] = neg ? 0.f : (over ? t.y : t.x);
                    rb[4 * i + 1] = neg ? t.x : (over ? t.z : t.y);
                    rb[4 * i + 2] = neg ? t.y : (over ? t.w : t.z);
                    rb[4 * i + 3] = neg ? t.z : (over ? 0.f : t.w);
                }
            } else {
                const unsigned bb = g4valid ? (unsigned)(img4 * p.C * HW + pq4) * 4u : OOB;
#pragma unroll
                for (int i = 0; i < BVN; ++i) pl_unpack4(&rb[4 * i], bload4(rx, ((bb | bvoff[i]) & OOB) ? OOB : bb + bvoff[i]));
            }
        } else {
            // wave-uniform pixel decode on the scalar unit: two divisions for the first pixel of the chunk, increments with carry for
            // the others (the chunk's pixels are consecutive output pixels)
            const int g0 = kt * BK + bkc * EB;
            int img = fdiv(g0 < p.Kg ? g0 : 0, p.d_pq);
            int pp = fdiv(g0 < p.Kg ? g0 - img * PQ : 0, p.d_q);
            int qq = (g0 < p.Kg ? g0 - img * PQ : 0) - pp * p.Q;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const bool gvalid = g0 + e < p.Kg;
                const int h0 = pp * p.SH - p.PH, w0 = qq * p.SW - p.PW;
                const int xb = img * p.C * HW + h0 * p.W + w0;   // element index of (img, 0, h0, w0); may sit in the padding
                const bool ok = gvalid && cvalid && (nopad || ((unsigned)(h0 + cr) < (unsigned)p.H && (unsigned)(w0 + cs) < (unsigned)p.W));
                rb[e] = bload(rx, ok ? (unsigned)(xb + coff) * 4u : OOB);
                if (++qq == p.Q) {
                    qq = 0;
                    if (++pp == p.P) { pp = 0; ++img; }
                }
            }
        }
    };

    constexpr int SA = decltype(sa)::STEPS, S = SA + decltype(sb)::STEPS;
    const unsigned fa = lds0 + LA::frag_base(lane, wm * T::WTM), fb = lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN);
    const int nk_total = (p.Kg + BK - 1) / BK;
    const int kt_begin = split * p.ktiles_per_split;
    int kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk_total) kt_end = nk_total;
    pl_mainloop<T, LA, LB, S>(fa, fb, kt_begin, kt_end, acc, load_tile, [&](unsigned wbuf, int s) {
        if (s < SA) {
            if constexpr (AFULL || !VECA) sa.step(s, ra, wbuf, LA::PIECE);
            else sa.step(s, ra, wbuf, LA::PIECE, aact);
        } else {
            sb.step(s - SA, rb, wbuf, LB::PIECE);
        }
    });

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
