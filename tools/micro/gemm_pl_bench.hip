// Pipeline study for the bf16-plane operand path (csrc/conv_planes_core.h) on a plain GEMM C[M][N] = A[M][K] * B[K][N], fp32 in
// memory (A k-contiguous = filters -> PL_R image, B n-contiguous = pixels of a 1x1 convolution -> PL_T image), split-bf16
// arithmetic (6 MFMA products of exactly split operands) — the forward 1x1 kernel without the convolution indexing.
//   usage: gemm_pl_bench <tile 0: 128x128/4 waves, 1: 256x128/8, 2: 128x256/8, 3: 256x256/8, 4-6: wave-specialised> <tiles_per_cu> [K] [M]
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I reid-gan_amd/csrc tools/micro/gemm_pl_bench.hip -o gpb [-DNO_GLOAD] ...
//   ablations (timing only, wrong results): -DNO_GLOAD (loads only for the first tile), -DNO_STAGE (no loads, split or LDS
//   stores), -DPL_NO_BARRIER, -DNO_MFMA, -DNO_EPI (no C stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int int4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2r __attribute__((ext_vector_type(2)));
struct Split3 { int4r hi, mid, lo; };
__device__ __forceinline__ void split3_pair(float x0, float x1, int& hi, int& mid, int& lo) {
    const float2r x = {x0, x1};
    hi = __builtin_bit_cast(int, __builtin_convertvector(x, bf16x2));
    const float2r r = {x0 - __builtin_bit_cast(float, (unsigned)hi << 16), x1 - __builtin_bit_cast(float, (unsigned)hi & 0xffff0000u)};
    mid = __builtin_bit_cast(int, __builtin_convertvector(r, bf16x2));
    const float2r l = {r[0] - __builtin_bit_cast(float, (unsigned)mid << 16), r[1] - __builtin_bit_cast(float, (unsigned)mid & 0xffff0000u)};
    lo = __builtin_bit_cast(int, __builtin_convertvector(l, bf16x2));
}
__device__ __forceinline__ floatx16 mfma_bf16(const int4r& a, const int4r& b, const floatx16& c) {
#ifdef NO_MFMA
    floatx16 r = c;
    r[0] += __builtin_bit_cast(float, a[0] ^ b[0]); r[1] += __builtin_bit_cast(float, a[1] ^ b[1]);
    r[2] += __builtin_bit_cast(float, a[2] ^ b[2]); r[3] += __builtin_bit_cast(float, a[3] ^ b[3]);
    return r;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
#define RG_PIN() __builtin_amdgcn_sched_barrier(0)
constexpr int BK = 16;
#include "conv_planes_core.h"

template <int BM_, int BN_, int WM_, int WN_, bool WS_ = false>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr bool WS = WS_;                        // wave-specialised: as many producer waves again
    static constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static constexpr int NL = 64 * WM * WN;                // loader threads (= consumer threads)
    static constexpr int NT = WS ? 2 * NL : NL;
};

template <typename T>
__global__ __launch_bounds__(T::NT) void gemm_pl(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                 int M, int N, int K, unsigned long long* __restrict__ clk) {
    // in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): shader cycles / 100 MHz real-time ticks around the whole workgroup
    unsigned long long c0 = 0, r0 = 0;
    if (clk && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int BM = T::BM, BN = T::BN, NT = T::NL;          // NT: threads of the loader mapping
    using LA = PlTile<PL_R, BM>;
    using LB = PlTile<PL_T, BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(size_t)((lds_u8_t*)lds);
    const bool producer = T::WS && __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) >= T::NL / 64;
    const int tid = producer ? threadIdx.x - T::NL : threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / T::WN, wn = wid % T::WN;
    const int m_tiles = M / BM;
#ifdef XCD_REMAP                                         // blocks b, b + 8, .. share an XCD: give each XCD a contiguous range of tiles
    const int bid = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
#else
    const int bid = blockIdx.x;
#endif
    const int mt = bid % m_tiles, nt = bid / m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
#ifdef LDA_PAD
    const int lda = K + LDA_PAD;
#else
    const int lda = K;
#endif
#ifdef CONSUMER_PRIO
    if (T::WS && !producer) __builtin_amdgcn_s_setprio(CONSUMER_PRIO);
#endif

    constexpr int NUA = BM * 4 / NT, NUB = (BN / 4) * BK / NT;
    constexpr int BROWS = NT / (BN / 4);                   // k rows of B per pass
    static_assert(NUA >= 1 && NUB >= 1, "loader shapes");
    PlStager<NUA, 2> sa;
    PlStager<NUB, 2> sb;
    const float* ap[NUA];
    const float* bp[NUB];
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
        const int v = tid + NT * i, row = v >> 2, kq = (v & 3) * 4;
        ap[i] = A + (int64_t)(m0 + row) * lda + kq;
        sa.wr[i] = lds0 + LA::off_rk(row, kq);
    }
    const int vcol = tid % (BN / 4), vrow0 = tid / (BN / 4);
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
        bp[i] = B + (int64_t)(vrow0 + i * BROWS) * N + n0 + 4 * vcol;
        sb.wr[i] = lds0 + LA::BYTES + LB::off_rk(4 * vcol, vrow0 + i * BROWS);
    }
#ifndef BENCH_NPF
#define BENCH_NPF 1
#endif
    constexpr int NSET = T::WS ? BENCH_NPF : 1;
    float ra[NSET][4 * NUA], rb[NSET][4 * NUB];
    floatx16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_set = [&](int kt, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#if defined(NO_GLOAD) || defined(NO_STAGE)
        if (kt >= NSET) return;
#endif
#ifdef SAME_TILE_LOADS                                   // every tile re-reads k-tile 0: same instructions, L1 / L2 hits only
        kt = 0;
#endif
#pragma unroll
        for (int i = 0; i < NUA; ++i) pl_unpack4(&ra[SET][4 * i], *reinterpret_cast<const float4*>(ap[i] + kt * BK));
#pragma unroll
        for (int i = 0; i < NUB; ++i) pl_unpack4(&rb[SET][4 * i], *reinterpret_cast<const float4*>(bp[i] + (int64_t)kt * BK * N));
    };
    auto load_tile = [&](int kt) { load_set(kt, IntTag<0>{}); };
    constexpr int SA = decltype(sa)::STEPS, S = SA + decltype(sb)::STEPS;
    const unsigned fa = lds0 + LA::frag_base(lane, wm * T::WTM), fb = lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN);
    auto stage_set = [&](unsigned wbuf, int s, auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#ifdef NO_STAGE
        if (wbuf != 0) return;
#endif
        if (s < SA) sa.step(s, ra[SET], wbuf, LA::PIECE);
        else sb.step(s - SA, rb[SET], wbuf, LB::PIECE);
    };
    auto stage = [&](unsigned wbuf, int s) { stage_set(wbuf, s, IntTag<0>{}); };
    if constexpr (T::WS) {
        pl_mainloop_ws<T, LA, LB, S, NSET>(producer, fa, fb, 0, K / BK, acc, load_set, stage_set);
        if (producer) return;
    } else {
        pl_mainloop<T, LA, LB, S>(fa, fb, 0, K / BK, acc, load_tile, stage);
    }
#ifndef NO_EPI
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * T::WTM + i * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
                C[(int64_t)m * N + n0 + wn * T::WTN + j * 32 + l32] = acc[i][j][r];
            }
#else
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot += acc[i][j][r];
    if (tot == 123.456f) C[0] = tot;
#endif
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}


// ---- the same pipeline on v_mfma_f32_16x16x32_bf16 with 32-deep k-tiles ----
template <typename T>
__global__ __launch_bounds__(T::NT) void gemm_pl16(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, unsigned long long* __restrict__ clk) {
    unsigned long long c0 = 0, r0 = 0;
    if (clk && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int BM = T::BM, BN = T::BN, NT = T::NL;
    using LA = PlTile32<PL_R, BM>;
    using LB = PlTile32<PL_T, BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lds0 = (unsigned)(size_t)((lds_u8_t*)lds);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / T::WN, wn = wid % T::WN;
    const int m_tiles = M / BM;
    const int mt = blockIdx.x % m_tiles, nt = blockIdx.x / m_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    constexpr int NUA = BM * 8 / NT, NUB = (BN / 4) * BK32 / NT;       // float4 units per thread and 32-deep tile
    constexpr int BROWS = NT / (BN / 4);
    static_assert(NUA >= 1 && NUB >= 1, "loader shapes");
    PlStager<NUA, 2> sa;
    PlStager<NUB, 2> sb;
    const float* ap[NUA];
    const float* bp[NUB];
#pragma unroll
    for (int i = 0; i < NUA; ++i) {
        const int v = tid + NT * i, row = v >> 3, kq = (v & 7) * 4;
        ap[i] = A + (int64_t)(m0 + row) * K + kq;
        sa.wr[i] = lds0 + LA::off_rk(row, kq);
    }
    const int vcol = tid % (BN / 4), vrow0 = tid / (BN / 4);
#pragma unroll
    for (int i = 0; i < NUB; ++i) {
        bp[i] = B + (int64_t)(vrow0 + i * BROWS) * N + n0 + 4 * vcol;
        sb.wr[i] = lds0 + LA::BYTES + LB::off_rk(4 * vcol, vrow0 + i * BROWS);
    }
    float ra[4 * NUA], rb[4 * NUB];
    floatx4 acc[2 * T::TM][2 * T::TN];
#pragma unroll
    for (int i = 0; i < 2 * T::TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * T::TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto load_tile = [&](int kt) {
#if defined(NO_GLOAD) || defined(NO_STAGE)
        if (kt > 0) return;
#endif
#pragma unroll
        for (int i = 0; i < NUA; ++i) pl_unpack4(&ra[4 * i], *reinterpret_cast<const float4*>(ap[i] + kt * BK32));
#pragma unroll
        for (int i = 0; i < NUB; ++i) pl_unpack4(&rb[4 * i], *reinterpret_cast<const float4*>(bp[i] + (int64_t)kt * BK32 * N));
    };
    constexpr int SA = decltype(sa)::STEPS, S = SA + decltype(sb)::STEPS;
    const unsigned fa[2] = {lds0 + LA::frag_base(lane, wm * T::WTM, 0), lds0 + LA::frag_base(lane, wm * T::WTM, 1)};
    const unsigned fb[2] = {lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN, 0), lds0 + LA::BYTES + LB::frag_base(lane, wn * T::WTN, 1)};
    pl_mainloop16<T, LA, LB, S>(fa, fb, 0, K / BK32, acc, load_tile, [&](unsigned wbuf, int s) {
#ifdef NO_STAGE
        if (wbuf != 0) return;
#endif
        if (s < SA) sa.step(s, ra, wbuf, LA::PIECE);
        else sb.step(s - SA, rb, wbuf, LB::PIECE);
    });
    const int l16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2 * T::TN; ++j)
#pragma unroll
        for (int i = 0; i < 2 * T::TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * T::WTM + i * 16 + 4 * g + r;
                C[(int64_t)m * N + n0 + wn * T::WTN + j * 16 + l16] = acc[i][j][r];
            }
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T, bool K16 = false>
static void run(int tpc, int K, int M) {
    const size_t shmem = K16 ? 2 * (size_t)(PlTile32<PL_R, T::BM>::BYTES + PlTile32<PL_T, T::BN>::BYTES)
                             : 2 * (size_t)(PlTile<PL_R, T::BM>::BYTES + PlTile<PL_T, T::BN>::BYTES);
    auto kern = K16 ? gemm_pl16<T> : gemm_pl<T>;
    const int m_tiles = M / T::BM;
    const int n_tiles = 256 * tpc / m_tiles;
    const int N = n_tiles * T::BN;
    #ifdef LDA_PAD
    const int lda = K + LDA_PAD;
#else
    const int lda = K;
#endif
    std::vector<float> hA((size_t)M * lda), hB((size_t)K * N);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const dim3 grid(m_tiles * n_tiles), block(T::NT);
    unsigned long long* dclk;
    CK(hipMalloc(&dclk, (size_t)grid.x * 16));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flop = 2.0 * M * (double)N * K;
    // clock: after ~reps launches of load, one more launch with stamps (the producer waves return early: thread 0 is a consumer)
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, K, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(kern, grid, block, shmem, 0, dA, dB, dC, M, N, K, dclk);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hclk((size_t)grid.x * 2);
    CK(hipMemcpy(hclk.data(), dclk, hclk.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz, wgcyc;
    for (unsigned b = 0; b < grid.x; ++b)
        if (hclk[2 * b + 1] > 0) { ghz.push_back((double)hclk[2 * b] / (double)hclk[2 * b + 1] * 0.1); wgcyc.push_back((double)hclk[2 * b]); }
    std::sort(ghz.begin(), ghz.end()); std::sort(wgcyc.begin(), wgcyc.end());
    const double clk_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2], wg_cycles = wgcyc.empty() ? 0.0 : wgcyc[wgcyc.size() / 2];
    // spot check against fp64
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0.0;
    for (int t = 0; t < 64; ++t) {
        const int m = (t * 37) % M, n = (int)(((int64_t)t * 7919) % N);
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * lda + k] * hB[(size_t)k * N + n];
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * N + n]));
    }
    printf("%stile %dx%d waves %d  M %d N %d K %d  WGs %d (%d/CU)  lds %zu B: %.3f ms  %.1f TFLOP/s  clock %.2f GHz  WG %.0f cyc = %.0f per k-tile  (MFMA-bound: %d per k-tile)  maxerr %.2e\n",
           K16 ? "16x16x32 " : "", T::BM, T::BN, T::NT / 64, M, N, K, m_tiles * n_tiles, tpc, shmem, ms, flop / ms * 1e-9, clk_ghz, wg_cycles, wg_cycles / (K / BK),
           6 * T::TM * T::TN * 32 * (K16 ? 2 : 1), maxerr);
    CK(hipFree(dclk));
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main(int argc, char** argv) {
    const int tile = argc > 1 ? atoi(argv[1]) : 0;
    const int tpc = argc > 2 ? atoi(argv[2]) : 2;
    const int K = argc > 3 ? atoi(argv[3]) : 2048;
    const int M = argc > 4 ? atoi(argv[4]) : 512;
    if (tile == 0) run<Tile<128, 128, 2, 2>>(tpc, K, M);
    else if (tile == 1) run<Tile<256, 128, 4, 2>>(tpc, K, M);
    else if (tile == 2) run<Tile<128, 256, 2, 4>>(tpc, K, M);
    else if (tile == 3) run<Tile<256, 256, 4, 2>>(tpc, K, M);
    else if (tile == 4) run<Tile<128, 128, 2, 2, true>>(tpc, K, M);      // 4 consumer + 4 producer waves
    else if (tile == 5) run<Tile<256, 128, 4, 2, true>>(tpc, K, M);      // 8 + 8
    else if (tile == 6) run<Tile<128, 256, 2, 2, true>>(tpc, K, M);      // 4 + 4, 64 x 128 wave tiles
    else if (tile == 7) run<Tile<128, 128, 2, 2>, true>(tpc, K, M);      // v_mfma_f32_16x16x32_bf16, 32-deep k-tiles
    else if (tile == 8) run<Tile<256, 128, 4, 2>, true>(tpc, K, M);
    else if (tile == 9) run<Tile<128, 128, 4, 2>, true>(tpc, K, M);      // 8 waves on the 128 x 128 tile (32 x 64 wave tiles)
    else if (tile == 10) run<Tile<128, 128, 2, 4>, true>(tpc, K, M);     // 64 x 32 wave tiles
    else if (tile == 11) run<Tile<128, 128, 4, 2>>(tpc, K, M);           // 8 waves, 32x32x16
    else if (tile == 12) run<Tile<128, 256, 2, 4>, true>(tpc, K, M);
    return 0;
}
