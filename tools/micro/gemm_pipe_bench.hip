// Pipeline study for the conv kernels' main loop on a plain GEMM C[M][N] = A[M][K] * B[K][N] (A k-contiguous = filters,
// B n-contiguous = pixels of a 1x1 convolution), 128x128 tile, 256 threads, fp32 MFMA 32x32x2.
//   v0: the scheme of conv_igemm.hip today — BK 16, global -> VGPR -> LDS, two LDS buffers, stores of tile t+1 interleaved
//       into the last k-steps of tile t, one barrier per tile (prefetch distance 1)
//   v1: three LDS buffers, prefetch distance 2 (loads of tile t+2 issued at the top of tile t, stored during tile t+1)
//   v2: BK 32, two buffers (dynamic LDS, 67.6 KB)
// usage: gemm_pipe_bench <variant> <tiles_per_cu 1|2|3> [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256, BM = 128, BN = 128;

template <int BK>
struct Stage {
    float4 a[BK / 8];      // A: 128 rows x BK k = 128*BK/4 float4 / 256 threads
    float4 b[BK / 8];      // B: BK k x 128 n
};

template <int BK>
__device__ __forceinline__ void gload(Stage<BK>& s, const float* __restrict__ A, const float* __restrict__ B, int K, int N,
                                      int m0, int n0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int v = tid + NT * i;                  // A: v -> (row = v / (BK/4), kq = (v % (BK/4)) * 4)
        const int row = v / (BK / 4), kq = (v % (BK / 4)) * 4;
        s.a[i] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * K + k0 + kq);
        const int kk = v / 32, nq = (v % 32) * 4;    // B: v -> (k = v / 32, n = (v % 32) * 4)
        s.b[i] = *reinterpret_cast<const float4*>(B + (int64_t)(k0 + kk) * N + n0 + nq);
    }
}

template <int BK, int LD>
__device__ __forceinline__ void lstore(const Stage<BK>& s, float (*As)[LD], float (*Bs)[LD], int tid, int part, int parts) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        if ((i * parts) / (BK / 8) != part && part >= 0) continue;
        const int v = tid + NT * i;
        const int row = v / (BK / 4), kq = (v % (BK / 4)) * 4;
        As[kq + 0][row] = s.a[i].x; As[kq + 1][row] = s.a[i].y; As[kq + 2][row] = s.a[i].z; As[kq + 3][row] = s.a[i].w;
        const int kk = v / 32, nq = (v % 32) * 4;
#ifdef B_SCALAR_STORE
        Bs[kk][nq + 0] = s.b[i].x; Bs[kk][nq + 1] = s.b[i].y; Bs[kk][nq + 2] = s.b[i].z; Bs[kk][nq + 3] = s.b[i].w;
#elif defined(B_B64_STORE)
        *reinterpret_cast<float2*>(&Bs[kk][nq]) = make_float2(s.b[i].x, s.b[i].y);
        *reinterpret_cast<float2*>(&Bs[kk][nq + 2]) = make_float2(s.b[i].z, s.b[i].w);
#else
        *reinterpret_cast<float4*>(&Bs[kk][nq]) = s.b[i];
#endif
    }
}

template <int BK, int LD, typename Hook>
__device__ __forceinline__ void mma(const float (*As)[LD], const float (*Bs)[LD], floatx16 (&acc)[2][2], int wm, int wn, int lane,
                                    Hook hook) {
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
        const int k = 2 * ks + kh;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[k][wm * 64 + i * 32 + l32];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[k][wn * 64 + j * 32 + l32];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        hook(ks);
    }
}

__device__ __forceinline__ void store_c(float* C, int N, int m0, int n0, int wm, int wn, int lane, const floatx16 (&acc)[2][2]) {
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                C[(int64_t)m * N + n0 + wn * 64 + j * 32 + l32] = acc[i][j][r];
            }
}

// ---- v0: two buffers, distance 1 ------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemm_v0(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Stage<BK> st;
    gload<BK>(st, A, B, K, N, m0, n0, 0, tid);
    lstore<BK, LD>(st, As[0], Bs[0], tid, -1, 1);
    __syncthreads();
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) gload<BK>(st, A, B, K, N, m0, n0, (kt + 1) * BK, tid);
        mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
            if (has_next && ks >= BK / 4) lstore<BK, LD>(st, As[cur ^ 1], Bs[cur ^ 1], tid, ks - BK / 4, BK / 4);
        });
        __syncthreads();
        cur ^= 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v1: three buffers, distance 2 ----------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemm_v1(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float As[3][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[3][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;            // even, >= 4 in this benchmark
    Stage<BK> s0, s1;                 // s0 holds tile kt+1 (to be stored during tile kt), s1 receives tile kt+2
    gload<BK>(s0, A, B, K, N, m0, n0, 0, tid);
    lstore<BK, LD>(s0, As[0], Bs[0], tid, -1, 1);
    gload<BK>(s0, A, B, K, N, m0, n0, BK, tid);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; kt += 2) {
        // even half: store s0 (tile kt+1) while computing kt, load kt+2 into s1
        if (kt + 2 < nk) gload<BK>(s1, A, B, K, N, m0, n0, (kt + 2) * BK, tid);
        {
            const int nb = (cur + 1) % 3;
            mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
                if (kt + 1 < nk && ks >= BK / 4) lstore<BK, LD>(s0, As[nb], Bs[nb], tid, ks - BK / 4, BK / 4);
            });
            __syncthreads();
            cur = nb;
        }
        if (kt + 1 >= nk) break;
        // odd half: store s1 (tile kt+2) while computing kt+1, load kt+3 into s0
        if (kt + 3 < nk) gload<BK>(s0, A, B, K, N, m0, n0, (kt + 3) * BK, tid);
        {
            const int nb = (cur + 1) % 3;
            mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
                if (kt + 2 < nk && ks >= BK / 4) lstore<BK, LD>(s1, As[nb], Bs[nb], tid, ks - BK / 4, BK / 4);
            });
            __syncthreads();
            cur = nb;
        }
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v2: BK 32, two buffers in dynamic LDS ------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemm_v2(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 32, LD = 132;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float(*As)[BK][LD] = reinterpret_cast<float(*)[BK][LD]>(smem);
    float(*Bs)[BK][LD] = reinterpret_cast<float(*)[BK][LD]>(smem + 2 * BK * LD);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Stage<BK> st;
    gload<BK>(st, A, B, K, N, m0, n0, 0, tid);
    lstore<BK, LD>(st, As[0], Bs[0], tid, -1, 1);
    __syncthreads();
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) gload<BK>(st, A, B, K, N, m0, n0, (kt + 1) * BK, tid);
        mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
            if (has_next && ks >= BK / 4) lstore<BK, LD>(st, As[cur ^ 1], Bs[cur ^ 1], tid, ks - BK / 4, BK / 4);
        });
        __syncthreads();
        cur ^= 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v3: three LDS buffers, ONE register stage: the tile loaded during tile t-1/t is stored into LDS in the FIRST k-steps
// of tile t (buffer (t+2)%3, free since the barrier that ended tile t-1), the next loads are issued right behind the stores:
// every global load has ~0.9 tile of matrix work between issue and first use --------------------------------------------
__global__ __launch_bounds__(NT) void gemm_v3(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float As[3][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[3][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
    Stage<BK> st;
    gload<BK>(st, A, B, K, N, m0, n0, 0, tid);
    lstore<BK, LD>(st, As[0], Bs[0], tid, -1, 1);
    if (nk > 1) {
        gload<BK>(st, A, B, K, N, m0, n0, BK, tid);
        lstore<BK, LD>(st, As[1], Bs[1], tid, -1, 1);
    }
    if (nk > 2) gload<BK>(st, A, B, K, N, m0, n0, 2 * BK, tid);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int wb = cur == 0 ? 2 : cur - 1;          // (cur + 2) % 3
        mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
            if (ks == 0 && kt + 2 < nk) lstore<BK, LD>(st, As[wb], Bs[wb], tid, 0, 2);
            if (ks == 1 && kt + 2 < nk) lstore<BK, LD>(st, As[wb], Bs[wb], tid, 1, 2);
            if (ks == 1 && kt + 3 < nk) gload<BK>(st, A, B, K, N, m0, n0, (kt + 3) * BK, tid);
        });
        __syncthreads();
        cur = cur == 2 ? 0 : cur + 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v4: v0 with parts switched off at compile time (timing only, results wrong): -DNO_GLOAD / -DNO_LSTORE / -DNO_BARRIER -----
__global__ __launch_bounds__(NT) void gemm_v4(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Stage<BK> st;
    gload<BK>(st, A, B, K, N, m0, n0, 0, tid);
    lstore<BK, LD>(st, As[0], Bs[0], tid, -1, 1);
    lstore<BK, LD>(st, As[1], Bs[1], tid, -1, 1);
    __syncthreads();
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
#ifndef NO_GLOAD
        if (has_next) gload<BK>(st, A, B, K, N, m0, n0, (kt + 1) * BK, tid);
#endif
        mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
#ifndef NO_LSTORE
            if (has_next && ks >= BK / 4) lstore<BK, LD>(st, As[cur ^ 1], Bs[cur ^ 1], tid, ks - BK / 4, BK / 4);
#endif
        });
#ifndef NO_BARRIER
        __syncthreads();
#else
        asm volatile("" ::: "memory");
#endif
        cur ^= 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v5: v0 with the A operand m-major in LDS ([128][20] floats): the k-contiguous global float4 goes to LDS with ONE
// ds_write_b128 (instead of 4 transposed ds_write_b32) and one ds_read_b128 feeds 4 MFMA steps; MFMA step (h, s) of half kh
// uses tile position p = 8h + 4kh + s for both operands (the order inside a k-tile is free) ------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(NT) void gemm_v5(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132, LA = 20;
    __shared__ __attribute__((aligned(16))) float As[2][BM][LA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, kh = lane >> 5;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Stage<BK> st;
    auto lst = [&](int buf, int part) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (part >= 0 && i != part) continue;
            const int v = tid + NT * i;
            const int row = v / 4, kq = (v % 4) * 4;
            *reinterpret_cast<float4*>(&As[buf][row][kq]) = st.a[i];
            const int kk = v / 32, nq = (v % 32) * 4;
            *reinterpret_cast<float4*>(&Bs[buf][kk][nq]) = st.b[i];
        }
    };
    gload<BK>(st, A, B, K, N, m0, n0, 0, tid);
    lst(0, -1);
    __syncthreads();
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) gload<BK>(st, A, B, K, N, m0, n0, (kt + 1) * BK, tid);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f4 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f4*>(&As[cur][wm * 64 + i * 32 + l32][8 * h + 4 * kh]);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const int pos = 8 * h + 4 * kh + s_;
                float b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Bs[cur][pos][wn * 64 + j * 32 + l32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s_], b[j], acc[i][j], 0, 0, 0);
                if (has_next && h == 1 && s_ >= 2) lst(cur ^ 1, s_ - 2);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v6: v0 with the A operand pre-transposed in GLOBAL memory (At[K][M], m contiguous — e.g. filters re-laid out once per
// weight version): both operands are staged exactly like B (float4 along the tile row, one ds_write_b128 each, no transposed
// scalar LDS writes) --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemm_v6(const float* At, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 132;
    __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[2], rb[2];
    auto ld = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + NT * i, kk = v / 32, q = (v % 32) * 4;
            ra[i] = *reinterpret_cast<const float4*>(At + (int64_t)(k0 + kk) * M + m0 + q);
            rb[i] = *reinterpret_cast<const float4*>(B + (int64_t)(k0 + kk) * N + n0 + q);
        }
    };
    auto stq = [&](int buf, int part) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (part >= 0 && i != (part >> 1)) continue;
            const int v = tid + NT * i, kk = v / 32, q = (v % 32) * 4;
            if (part < 0 || (part & 1) == 0) *reinterpret_cast<float4*>(&As[buf][kk][q]) = ra[i];
            if (part < 0 || (part & 1) == 1) *reinterpret_cast<float4*>(&Bs[buf][kk][q]) = rb[i];
        }
    };
    ld(0);
    stq(0, -1);
    __syncthreads();
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) ld((kt + 1) * BK);
        mma<BK, LD>(As[cur], Bs[cur], acc, wm, wn, lane, [&](int ks) {
            if (has_next && ks >= BK / 4) stq(cur ^ 1, ks - BK / 4);
        });
        __syncthreads();
        cur ^= 1;
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v7: both operands by LDS-DMA (global_load_lds_dwordx4 from At[K][M] / B[K][N]): no VGPR staging and no ds_write at all.
// NB LDS buffers of [16][128] floats per operand (UNPADDED: the DMA image is lane-linear, reads along the row are conflict-free
// anyway); the DMAs of tile t + NB-1 are issued at the top of tile t into the buffer tile t-1 just left, a counted vmcnt retires
// tile t+1 before the (raw) barrier that ends tile t.  Per thread and tile: 4 DMA instructions instead of 4 loads + 10 ds_writes.
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
template <int NB>
__global__ __launch_bounds__(NT) void gemm_v7(const float* At, const float* B, float* C, int M, int N, int K) {
    constexpr int BK = 16, LD = 128;
    __shared__ __attribute__((aligned(16))) float S[NB][2][BK][LD];      // one object: [buffer][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // a tile is 16 rows x 512 B = 8 pieces of 1 KiB (two rows); wave w moves pieces 2w and 2w+1 of A and of B
    auto dma = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wid * 2 + i;
            const int kk = 2 * piece + (lane >> 5), q = (lane & 31) * 4;
            __builtin_amdgcn_global_load_lds((glb_void*)(At + (int64_t)(k0 + kk) * M + m0 + q), (lds_void*)&S[buf][0][2 * piece][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(B + (int64_t)(k0 + kk) * N + n0 + q), (lds_void*)&S[buf][1][2 * piece][0], 16, 0, 0);
        }
    };
    const int nk = K / BK;
#pragma unroll
    for (int t = 0; t < NB - 1; ++t)
        if (t < nk) dma(t, t * BK);
    // tile 0 must have landed: all but the (NB-2) younger tiles' 4 DMAs each
    if (NB == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NB == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt % NB;
        const int pre = kt + NB - 1;
        if (pre < nk) dma(pre % NB, pre * BK);                            // into the buffer tile kt-1 used (all waves passed the barrier)
        mma<BK, LD>(S[cur][0], S[cur][1], acc, wm, wn, lane, [&](int) {});
        // retire tile kt+1 (leave the younger NB-2 tiles in flight), then let every wave finish reading tile kt
        if (NB == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NB == 3) { if (pre < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { if (pre < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    store_c(C, N, m0, n0, wm, wn, lane, acc);
}

// ---- v8: EIGHT waves per workgroup on the same 128x128xBK tile: wave group g = wid >> 2 runs the MFMA k-steps of half g of every
// LDS tile (same LDS traffic and global loads per tile as v0, half the staging work per thread, TWO waves per SIMD at one
// workgroup per CU), the two partial accumulators are added through LDS at the end.  BK8 = 16 or 32. --------------------------
template <int BK8>
__global__ __launch_bounds__(512) void gemm_v8(const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int LD = 132, T = 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];            // max(staging 2*2*BK8*LD, reduction 128*LD) floats
    float (*As)[BK8][LD] = reinterpret_cast<float (*)[BK8][LD]>(smem);
    float (*Bs)[BK8][LD] = reinterpret_cast<float (*)[BK8][LD]>(smem + 2 * BK8 * LD);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, grp = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1;
    const int m0 = (blockIdx.x % (M / BM)) * BM, n0 = (blockIdx.x / (M / BM)) * BN;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int NV = BK8 / 16;                   // float4 per thread, operand and tile (128*BK8/4 / 512)
    float4 ra[NV], rb[NV];
    auto ld = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + T * i;
            const int row = v / (BK8 / 4), kq = (v % (BK8 / 4)) * 4;
            ra[i] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * K + k0 + kq);
            const int kk = v / 32, nq = (v % 32) * 4;
            rb[i] = *reinterpret_cast<const float4*>(B + (int64_t)(k0 + kk) * N + n0 + nq);
        }
    };
    auto st = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + T * i;
            const int row = v / (BK8 / 4), kq = (v % (BK8 / 4)) * 4;
            As[buf][kq + 0][row] = ra[i].x; As[buf][kq + 1][row] = ra[i].y; As[buf][kq + 2][row] = ra[i].z; As[buf][kq + 3][row] = ra[i].w;
            const int kk = v / 32, nq = (v % 32) * 4;
            *reinterpret_cast<float4*>(&Bs[buf][kk][nq]) = rb[i];
        }
    };
    ld(0);
    st(0);
    __syncthreads();
    const int nk = K / BK8;
    const int l32 = lane & 31, kh = lane >> 5;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool has_next = kt + 1 < nk;
        if (has_next) ld((kt + 1) * BK8);
#pragma unroll
        for (int ks = 0; ks < BK8 / 4; ++ks) {      // this group's half of the tile's k-steps
            const int k = grp * (BK8 / 2) + 2 * ks + kh;
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[cur][k][wm * 64 + i * 32 + l32];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[cur][k][wn * 64 + j * 32 + l32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (has_next && ks == BK8 / 4 - 1) st(cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }
    // group 1 hands its partial tile over through LDS (aliases the staging buffers: every wave is past the last barrier)
    float (*R)[LD] = reinterpret_cast<float (*)[LD]>(smem);
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    R[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh][wn * 64 + j * 32 + l32] = acc[i][j][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += R[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh][wn * 64 + j * 32 + l32];
        store_c(C, N, m0, n0, wm, wn, lane, acc);
    }
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 1;
    const int K = argc > 3 ? atoi(argv[3]) : 512;
    const int M = 512, N = 128 * (256 * per_cu / (M / 128));       // tiles = (M/128) * (N/128) = 256 * per_cu
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> ha((size_t)M * K), hb((size_t)K * N);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = 0.01f * (float)((int)((i * 7) % 13) - 6);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = 0.02f * (float)((int)((i * 5) % 11) - 5);
    hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    float* At;
    hipMalloc(&At, (size_t)M * K * 4);
    {
        std::vector<float> hat((size_t)M * K);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hat[(size_t)k * M + m] = ha[(size_t)m * K + k];
        hipMemcpy(At, hat.data(), hat.size() * 4, hipMemcpyHostToDevice);
    }
    const int blocks = (M / 128) * (N / 128);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v2), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 32 * 132 * 4);
    const int v8_lds16 = 128 * 132 * 4, v8_lds32 = 128 * 132 * 4;      // reduction tile (67.6 KB) >= staging (33.8 / 67.6 KB)
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v8<16>), hipFuncAttributeMaxDynamicSharedMemorySize, v8_lds16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v8<32>), hipFuncAttributeMaxDynamicSharedMemorySize, v8_lds32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(gemm_v0, dim3(blocks), dim3(NT), 0, 0, A, B, C, M, N, K);
        else if (variant == 1) hipLaunchKernelGGL(gemm_v1, dim3(blocks), dim3(NT), 0, 0, A, B, C, M, N, K);
        else if (variant == 6) hipLaunchKernelGGL(gemm_v6, dim3(blocks), dim3(NT), 0, 0, At, B, C, M, N, K);
        else if (variant == 72) hipLaunchKernelGGL(gemm_v7<2>, dim3(blocks), dim3(NT), 0, 0, At, B, C, M, N, K);
        else if (variant == 73) hipLaunchKernelGGL(gemm_v7<3>, dim3(blocks), dim3(NT), 0, 0, At, B, C, M, N, K);
        else if (variant == 74) hipLaunchKernelGGL(gemm_v7<4>, dim3(blocks), dim3(NT), 0, 0, At, B, C, M, N, K);
        else if (variant == 816) hipLaunchKernelGGL(gemm_v8<16>, dim3(blocks), dim3(512), v8_lds16, 0, A, B, C, M, N, K);
        else if (variant == 832) hipLaunchKernelGGL(gemm_v8<32>, dim3(blocks), dim3(512), v8_lds32, 0, A, B, C, M, N, K);
        else if (variant == 5) hipLaunchKernelGGL(gemm_v5, dim3(blocks), dim3(NT), 0, 0, A, B, C, M, N, K);
        else if (variant == 4) hipLaunchKernelGGL(gemm_v4, dim3(blocks), dim3(NT), 0, 0, A, B, C, M, N, K);
        else if (variant == 3) hipLaunchKernelGGL(gemm_v3, dim3(blocks), dim3(NT), 0, 0, A, B, C, M, N, K);
        else hipLaunchKernelGGL(gemm_v2, dim3(blocks), dim3(NT), 2 * 2 * 32 * 132 * 4, 0, A, B, C, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    // checksum against a host reference of a few entries
    std::vector<float> hc((size_t)M * N);
    hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 64; ++t) {
        const int m = (t * 37) % M, n = (t * 101) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)ha[(size_t)m * K + k] * hb[(size_t)k * N + n];
        const double e = ref - hc[(size_t)m * N + n];
        if ((e < 0 ? -e : e) > maxerr) maxerr = e < 0 ? -e : e;
    }
    printf("v%d  %d tile(s)/CU  M %d N %d K %d: %.1f us  %.1f TFLOP/s  (max err %.2e)\n", variant, per_cu, M, N, K, best * 1e3,
           2.0 * M * N * K / best / 1e9, maxerr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
