// Core of the bf16-plane operand path (layouts, split-and-store chunks, the MFMA k-step, the k-loop): see conv_planes.h for the
// design notes.  Depends only on the arithmetic typedefs of conv_igemm.hip (floatx16, int4r, Split3, split3_pair, mfma_bf16,
// RG_PIN), so tools/micro/gemm_pl_bench.hip can include it on its own.
#pragma once

enum { PL_R = 0, PL_T = 1 };

typedef int int2r __attribute__((ext_vector_type(2)));
typedef short short4r __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4r lds_short4_t;
typedef __attribute__((address_space(3))) int4r lds_int4_t;
typedef __attribute__((address_space(3))) int2r lds_int2_t;
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
template <bool B> struct BoolTag { static constexpr bool value = B; };

constexpr int pl_tpitch(int rows) {
    const int p = rows * 2, b = p / 256 * 256;
    return b + 64 >= p ? b + 64 : (b + 192 >= p ? b + 192 : b + 320);
}

template <int KIND, int ROWS>
struct PlTile {
    static constexpr int HALF = ROWS * 16 + 64;                 // PL_R: bytes between the two k halves of a piece
    static constexpr int PITCH = pl_tpitch(ROWS);               // PL_T: bytes between k rows
    static constexpr int PIECE = KIND == PL_R ? 2 * HALF : 16 * PITCH;
    static constexpr int BYTES = 3 * PIECE;
    static constexpr int BLK = KIND == PL_R ? 512 : 64;         // bytes between consecutive 32-row blocks of a fragment read
    // byte offset, inside a piece, of this lane's fragment read of the 32-row block that starts at row0
    __device__ __forceinline__ static unsigned frag_base(int lane, int row0) {
        if (KIND == PL_R) return (unsigned)((lane >> 5) * HALF + (row0 + (lane & 31)) * 16);
        const int g = lane >> 4, i = lane & 15;                 // group g: rows row0 + 16 (g & 1) .., k 8 (g >> 1) ..
        return (unsigned)((8 * (g >> 1) + (i >> 2)) * PITCH + (row0 + 16 * (g & 1) + 4 * (i & 3)) * 2);
    }
    // the 8 bf16 (k = 8 (lane >> 5) .. + 7) of this lane's row; `a`: LDS byte address (lane base + compile-time piece / block / buffer)
    __device__ __forceinline__ static int4r frag(unsigned a) {
        if (KIND == PL_R) return *(const lds_int4_t*)(size_t)(a);
        const short4r lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(size_t)(a));
        const short4r hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(size_t)(a + 4 * PITCH));
        const int2r l = __builtin_bit_cast(int2r, lo), h = __builtin_bit_cast(int2r, hi);
        return int4r{l[0], l[1], h[0], h[1]};
    }
    // store offsets (inside a piece)
    __device__ __forceinline__ static unsigned off_rk(int row, int k) {       // element (row, k); PL_R: k % chunk == 0 keeps a chunk whole
        if (KIND == PL_R) return (unsigned)((k >> 3) * HALF + row * 16 + (k & 7) * 2);
        return (unsigned)(k * PITCH + row * 2);
    }
};

// x - y as ONE v_sub_f32: left to itself the compiler pairs the residual subtractions of an element pair into v_pk_add_f32, which
// costs ~4x a plain VALU instruction beside MFMAs (MI355X_MICROARCH.md, cycle constants: packed f32 is an anti-lever there)
__device__ __forceinline__ float pl_sub(float x, float y) {
    float d;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
    return d;
}

// A staged chunk of NP element pairs (PL_R: consecutive k of one row; PL_T: consecutive rows of one k), split in two micro-steps
// per pair (5 + 6 VALU instructions) so that the kernels can spread the work behind the MFMAs of a k-tile, and written (one store
// of NP dwords per piece) once the last pair is done.  Same values as split3_pair.
template <int NP>
struct PlChunk {
    int h[NP], m[NP], l[NP];
    float r0, r1;                                               // residual of the pair between its two micro-steps
    __device__ __forceinline__ void step_a(int j, float x0, float x1) {
        const float2r x = {x0, x1};
        h[j] = __builtin_bit_cast(int, __builtin_convertvector(x, bf16x2));
        r0 = pl_sub(x0, __builtin_bit_cast(float, (unsigned)h[j] << 16));
        r1 = pl_sub(x1, __builtin_bit_cast(float, (unsigned)h[j] & 0xffff0000u));
    }
    __device__ __forceinline__ void step_b(int j) {
        const float2r r = {r0, r1};
        m[j] = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
        const float2r t = {pl_sub(r0, __builtin_bit_cast(float, (unsigned)m[j] << 16)),
                           pl_sub(r1, __builtin_bit_cast(float, (unsigned)m[j] & 0xffff0000u))};
        l[j] = __builtin_bit_cast(int, __builtin_convertvector(t, bf16x2));
    }
    __device__ __forceinline__ void write(unsigned a, int piece_bytes) const {       // a: LDS byte address in piece 0
        if (NP == 1) {
            *(lds_int_t*)(size_t)(a) = h[0];
            *(lds_int_t*)(size_t)(a + piece_bytes) = m[0];
            *(lds_int_t*)(size_t)(a + 2 * piece_bytes) = l[0];
        } else if (NP == 2) {
            *(lds_int2_t*)(size_t)(a) = int2r{h[0], h[1 % NP]};
            *(lds_int2_t*)(size_t)(a + piece_bytes) = int2r{m[0], m[1 % NP]};
            *(lds_int2_t*)(size_t)(a + 2 * piece_bytes) = int2r{l[0], l[1 % NP]};
        } else {
            *(lds_int4_t*)(size_t)(a) = int4r{h[0], h[1 % NP], h[2 % NP], h[3 % NP]};
            *(lds_int4_t*)(size_t)(a + piece_bytes) = int4r{m[0], m[1 % NP], m[2 % NP], m[3 % NP]};
            *(lds_int4_t*)(size_t)(a + 2 * piece_bytes) = int4r{l[0], l[1 % NP], l[2 % NP], l[3 % NP]};
        }
    }
};

// One 16-deep k-step of a (TM x 32) x (TN x 32) wave tile: ra(i, pc) / rb(j, pc) return piece pc (0 hi, 1 mid, 2 lo) of A block i /
// B block j.  hook(slot) runs behind MFMA number `slot` (0 .. 6 TM TN - 1): the kernels split and store the next tile's staged
// registers there, a few VALU / LDS instructions per MFMA, which the matrix pipe (32 cycles per MFMA, 8 of them issue) covers.
template <int TM, int TN, typename RA, typename RB, typename Hook>
__device__ __forceinline__ void mma_pl(RA ra, RB rb, floatx16 (&acc)[TM][TN], Hook hook) {
    Split3 a[TM], b[TN];
    // reads issued in the order of first use (small terms first: lo x hi, hi x lo, mid x mid, ...)
    a[0].lo = ra(0, 2); b[0].hi = rb(0, 0);
    a[0].hi = ra(0, 0); b[0].lo = rb(0, 2);
    a[0].mid = ra(0, 1); b[0].mid = rb(0, 1);
#pragma unroll
    for (int i = 1; i < TM; ++i) { a[i].lo = ra(i, 2); a[i].hi = ra(i, 0); a[i].mid = ra(i, 1); }
#pragma unroll
    for (int j = 1; j < TN; ++j) { b[j].hi = rb(j, 0); b[j].lo = rb(j, 2); b[j].mid = rb(j, 1); }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int s0 = 6 * (j * TM + i);
            floatx16& c = acc[i][j];
            c = mfma_bf16(a[i].lo, b[j].hi, c);   hook(s0 + 0);  RG_PIN();
            c = mfma_bf16(a[i].hi, b[j].lo, c);   hook(s0 + 1);  RG_PIN();
            c = mfma_bf16(a[i].mid, b[j].mid, c); hook(s0 + 2);  RG_PIN();
            c = mfma_bf16(a[i].mid, b[j].hi, c);  hook(s0 + 3);  RG_PIN();
            c = mfma_bf16(a[i].hi, b[j].mid, c);  hook(s0 + 4);  RG_PIN();
            c = mfma_bf16(a[i].hi, b[j].hi, c);   hook(s0 + 5);  RG_PIN();
        }
}

// staging micro-steps [first, end) that run behind MFMA slot `slot` of NM: the S micro-steps sit at the END of the tile's MFMAs
// (the global loads issued at the top of the tile get the longest possible cover), one per slot; more steps than MFMAs: several
template <int S, int NM>
__device__ __forceinline__ constexpr int pl_first_step(int slot) {
    return S <= NM ? slot - (NM - S) : (slot * S) / NM;
}
template <int S, int NM>
__device__ __forceinline__ constexpr int pl_end_step(int slot) {
    return S <= NM ? slot - (NM - S) + 1 : ((slot + 1) * S) / NM;
}

// The k-loop shared by the kernels: LDS double buffer with compile-time buffer offsets (the loop body handles two k-tiles, so
// every LDS address is a per-thread base + an immediate), one barrier per k-tile, global loads of tile t+1 issued before the
// MFMAs of tile t and split / stored behind them.  load(kt): global -> staging registers; tile(cur_tag, stage_tag): the MFMAs
// of the tile in buffer cur (+ staging into the other buffer).
#ifdef PL_NO_BARRIER                                             // tools/micro ablation only
#define PL_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define PL_SYNC() __syncthreads()
#endif
template <typename Load, typename TileFn>
__device__ __forceinline__ void pl_kloop(int kt_begin, int kt_end, Load load, TileFn tile) {
    int kt = kt_begin;
    for (; kt + 2 < kt_end; kt += 2) {
        load(kt + 1);
        tile(BoolTag<false>{}, BoolTag<true>{});
        PL_SYNC();
        load(kt + 2);
        tile(BoolTag<true>{}, BoolTag<true>{});
        PL_SYNC();
    }
    if (kt + 1 < kt_end) {
        load(kt + 1);
        tile(BoolTag<false>{}, BoolTag<true>{});
        PL_SYNC();
        tile(BoolTag<true>{}, BoolTag<false>{});
    } else if (kt < kt_end) {
        tile(BoolTag<false>{}, BoolTag<false>{});
    }
}

// ---- shared pieces of the data-gradient / weight-gradient kernels ---------------------------------------------------------
// NU staged units of SP element pairs each (values r[u * 2 SP ..]), micro-step s of 2 NU SP; wr[u]: LDS byte address of the
// unit's store in piece 0 of buffer 0
template <int NU, int SP>
struct PlStager {
    static constexpr int STEPS = 2 * NU * SP;
    unsigned wr[NU];
    PlChunk<SP> c[NU];
    template <int NR>
    __device__ __forceinline__ void step(int s, const float (&r)[NR], unsigned wbuf, int piece_bytes, bool active = true) {
        const int pr = s >> 1, u = pr / SP, j = pr % SP;
        if (!(s & 1)) {
            c[u].step_a(j, r[u * 2 * SP + 2 * j], r[u * 2 * SP + 2 * j + 1]);
        } else {
            c[u].step_b(j);
            if (j == SP - 1 && active) c[u].write(wr[u] + wbuf, piece_bytes);
        }
    }
};

__device__ __forceinline__ void pl_unpack4(float* r, const float4& t) { r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }

// prologue + k-loop of a workgroup tile: fa / fb are the lane's fragment-read addresses in buffer 0, stage(wbuf, s) runs
// micro-step s (of S) of the staging into the buffer at byte offset wbuf
template <typename T, typename LA, typename LB, int S, typename Load, typename Stage>
__device__ __forceinline__ void pl_mainloop(unsigned fa, unsigned fb, int kt_begin, int kt_end, floatx16 (&acc)[T::TM][T::TN],
                                            Load load_tile, Stage stage) {
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    constexpr int NM = 6 * T::TM * T::TN;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
#pragma unroll
        for (int s = 0; s < S; ++s) stage(0, s);
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
                                         if (s >= 0) stage(WR, s);
                                 }
                             });
    });
}


// ---- wave-specialised k-loop ----------------------------------------------------------------------------------------------
// The workgroup holds twice the waves of its MFMA tile: waves 0 .. NW - 1 are CONSUMERS (fragment reads + MFMAs, nothing else in
// their instruction stream), waves NW .. 2 NW - 1 PRODUCERS (global loads, split, LDS stores of the next tile).  With the usual
// round-robin placement every SIMD holds one of each, so the split's VALU work and the vector-memory waits sit in a different
// wave from the one that feeds the matrix pipe.  Lockstep through ONE workgroup barrier per k-tile: in interval t the consumers
// read buffer t & 1 while the producers fill buffer (t + 1) & 1 with tile t + 1 (loads issued one interval earlier) and issue the
// loads of tile t + 2.  The barrier waits for LDS only (lgkmcnt), never for the producers' outstanding global loads.
#define PL_WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int I> struct IntTag { static constexpr int value = I; };

// NPF: register sets of the producers = k-tiles whose global loads are in flight (tile j lives in set j % NPF);
// load_tile(kt, IntTag<set>), stage(wbuf, s, IntTag<set>)
template <typename T, typename LA, typename LB, int S, int NPF, typename Load, typename Stage>
__device__ __forceinline__ void pl_mainloop_ws(bool producer, unsigned fa, unsigned fb, int kt_begin, int kt_end,
                                               floatx16 (&acc)[T::TM][T::TN], Load load_tile, Stage stage) {
    static_assert(NPF >= 1 && NPF <= 4, "1 .. 4 register sets");
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    const int n = kt_end > kt_begin ? kt_end - kt_begin : 0;
    if (producer) {
        if (n > 0) load_tile(kt_begin, IntTag<0>{});
        if (NPF > 1 && n > 1) load_tile(kt_begin + 1, IntTag<1 % NPF>{});
        if (NPF > 2 && n > 2) load_tile(kt_begin + 2, IntTag<2 % NPF>{});
        if (NPF > 3 && n > 3) load_tile(kt_begin + 3, IntTag<3 % NPF>{});
        if (n > 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) stage(0u, s, IntTag<0>{});
            if (n > NPF) load_tile(kt_begin + NPF, IntTag<0>{});
        }
        // step t (t % NPF == P): barrier, then tile t + 1 from set (t + 1) % NPF into buffer (t + 1) & 1, then the loads of tile t + 1 + NPF
        auto step = [&](auto ptag, int t) {
            constexpr int SET = (decltype(ptag)::value + 1) % NPF;
            PL_WS_BARRIER();
            if (t + 1 < n) {
                const unsigned wbuf = ((t + 1) & 1) ? (unsigned)TILEB : 0u;
#pragma unroll
                for (int s = 0; s < S; ++s) stage(wbuf, s, IntTag<SET>{});
                if (t + 1 + NPF < n) load_tile(kt_begin + t + 1 + NPF, IntTag<SET>{});
            }
        };
        for (int t = 0; t < n; t += NPF) {
            step(IntTag<0>{}, t);
            if (NPF > 1 && t + 1 < n) step(IntTag<1 % NPF>{}, t + 1);
            if (NPF > 2 && t + 2 < n) step(IntTag<2 % NPF>{}, t + 2);
            if (NPF > 3 && t + 3 < n) step(IntTag<3 % NPF>{}, t + 3);
        }
        return;
    }
    auto tile = [&](auto cur_tag) {
        constexpr int RD = decltype(cur_tag)::value ? TILEB : 0;
        mma_pl<T::TM, T::TN>([&](int i, int pc) { return LA::frag(fa + RD + pc * LA::PIECE + i * LA::BLK); },
                             [&](int j, int pc) { return LB::frag(fb + RD + pc * LB::PIECE + j * LB::BLK); }, acc, [](int) {});
    };
    int t = 0;
    for (; t + 1 < n; t += 2) {
        PL_WS_BARRIER();
        tile(BoolTag<false>{});
        PL_WS_BARRIER();
        tile(BoolTag<true>{});
    }
    if (t < n) {
        PL_WS_BARRIER();
        tile(BoolTag<false>{});
    }
}

// ---- 16x16x32 variant ------------------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 issues the same FLOPs per cycle as the 32x32x16 form, but the chip holds a higher clock on it under
// load (MI355X_MICROARCH.md, DVFS item 7; tools/micro/mfma_shape_bench.hip).  Operand tiles are 32 deep: PL_R = [piece][k group of
// 8][row][8 k] (rows XOR-ed with 2 x group inside their aligned 8-row block: the four 8-byte store runs of a 16-lane group land on
// four different bank quarters, the 16-byte fragment reads stay 256 contiguous bytes per 16 lanes), PL_T = [piece][32 k][row] with
// the 16-column halves swapped on odd k groups (the two k groups a 32-lane half reads transposed fall on disjoint banks).
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int BK32 = 32;

__device__ __forceinline__ floatx4 mfma16_bf16(const int4r& a, const int4r& b, const floatx4& c) {
#ifdef NO_MFMA
    floatx4 r = c;
    r[0] += __builtin_bit_cast(float, a[0] ^ b[0]); r[1] += __builtin_bit_cast(float, a[1] ^ b[1]);
    r[2] += __builtin_bit_cast(float, a[2] ^ b[2]); r[3] += __builtin_bit_cast(float, a[3] ^ b[3]);
    return r;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

template <int KIND, int ROWS>
struct PlTile32 {
    static constexpr int GROUP = ROWS * 16;                     // PL_R: bytes between k groups
    static constexpr int PITCH = pl_tpitch(ROWS);               // PL_T: bytes between k rows
    static constexpr int PIECE = KIND == PL_R ? 4 * GROUP : 32 * PITCH;
    static constexpr int BYTES = 3 * PIECE;
    static constexpr int BLK = KIND == PL_R ? 256 : 32;         // bytes between consecutive 16-row blocks of a fragment read
    // fragment read of the 16-row block b (0 ..) of the wave's rows: base(lane, row0, b & 1) + (b >> 1) * 2 * BLK for PL_T (the
    // swapped halves make odd and even blocks use different bases), base + b * BLK for PL_R; row0 % 32 == 0
    __device__ __forceinline__ static unsigned frag_base(int lane, int row0, int odd) {
        const int g = lane >> 4, i = lane & 15;
        if (KIND == PL_R) return (unsigned)(g * GROUP + ((row0 + 16 * odd + i) ^ (2 * g)) * 16);
        const int k = 8 * g + (i >> 2), col = (row0 + 16 * odd + 4 * (i & 3)) ^ (16 * (g & 1));
        return (unsigned)(k * PITCH + col * 2);
    }
    __device__ __forceinline__ static int4r frag(unsigned a) {
        if (KIND == PL_R) return *(const lds_int4_t*)(size_t)(a);
        const short4r lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(size_t)(a));
        const short4r hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(size_t)(a + 4 * PITCH));
        const int2r l = __builtin_bit_cast(int2r, lo), h = __builtin_bit_cast(int2r, hi);
        return int4r{l[0], l[1], h[0], h[1]};
    }
    __device__ __forceinline__ static unsigned off_rk(int row, int k) {
        if (KIND == PL_R) return (unsigned)((k >> 3) * GROUP + ((row ^ (2 * (k >> 3))) * 16) + (k & 7) * 2);
        return (unsigned)(k * PITCH + (row ^ (16 * ((k >> 3) & 1))) * 2);
    }
};

// one 32-deep k-step of a (TM x 32) x (TN x 32) wave tile as (2 TM) x (2 TN) blocks of 16 x 16; ra(ib, pc) / rb(jb, pc): piece pc of
// 16-row block ib / jb; hook(slot) behind MFMA number slot (0 .. 24 TM TN - 1)
template <int TM, int TN, typename RA, typename RB, typename Hook>
__device__ __forceinline__ void mma_pl16(RA ra, RB rb, floatx4 (&acc)[2 * TM][2 * TN], Hook hook) {
    Split3 a[2 * TM], b[2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) { a[i].lo = ra(i, 2); a[i].hi = ra(i, 0); a[i].mid = ra(i, 1); }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) { b[j].hi = rb(j, 0); b[j].lo = rb(j, 2); b[j].mid = rb(j, 1); }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
            const int s0 = 6 * (j * 2 * TM + i);
            floatx4& c = acc[i][j];
            c = mfma16_bf16(a[i].lo, b[j].hi, c);   hook(s0 + 0);  RG_PIN();
            c = mfma16_bf16(a[i].hi, b[j].lo, c);   hook(s0 + 1);  RG_PIN();
            c = mfma16_bf16(a[i].mid, b[j].mid, c); hook(s0 + 2);  RG_PIN();
            c = mfma16_bf16(a[i].mid, b[j].hi, c);  hook(s0 + 3);  RG_PIN();
            c = mfma16_bf16(a[i].hi, b[j].mid, c);  hook(s0 + 4);  RG_PIN();
            c = mfma16_bf16(a[i].hi, b[j].hi, c);   hook(s0 + 5);  RG_PIN();
        }
}

template <typename T, typename LA, typename LB, int S, typename Load, typename Stage>
__device__ __forceinline__ void pl_mainloop16(const unsigned (&fa)[2], const unsigned (&fb)[2], int kt_begin, int kt_end,
                                              floatx4 (&acc)[2 * T::TM][2 * T::TN], Load load_tile, Stage stage) {
    constexpr int TILEB = LA::BYTES + LB::BYTES;
    constexpr int NM = 24 * T::TM * T::TN;
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
#pragma unroll
        for (int s = 0; s < S; ++s) stage(0, s);
    }
    __syncthreads();
    pl_kloop(kt_begin, kt_end, load_tile, [&](auto cur_tag, auto stage_tag) {
        constexpr int RD = decltype(cur_tag)::value ? TILEB : 0, WR = TILEB - RD;
        constexpr bool STAGE = decltype(stage_tag)::value;
        mma_pl16<T::TM, T::TN>([&](int i, int pc) { return LA::frag(fa[i & 1] + RD + pc * LA::PIECE + (i >> 1) * 2 * LA::BLK); },
                               [&](int j, int pc) { return LB::frag(fb[j & 1] + RD + pc * LB::PIECE + (j >> 1) * 2 * LB::BLK); }, acc,
                               [&](int slot) {
                                   if (STAGE) {
#pragma unroll
                                       for (int s = pl_first_step<S, NM>(slot); s < pl_end_step<S, NM>(slot); ++s)
                                           if (s >= 0) stage(WR, s);
                                   }
                               });
    });
}
