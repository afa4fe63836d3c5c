// Batched strided fp32 GEMM on the MFMA pipe (v_mfma_f32_32x32x2_f32) for the attention blocks of the dual_gan
// generators: nn.MultiheadAttention inside CAB / TTB, CC/dual_gan/models/PTM.py:162-247.
//
//   C[b0][b1][m][n] = alpha * sum_k A[b0][b1][m][k] * B[b0][b1][k][n]  (+ beta * C)
//
// Every operand is addressed by element strides (row / column / two batch levels), so the token maps stay in the
// [B][C][L] layout of the surrounding convolutions: head h of a projection is just a channel offset, Q^T K, P V and
// the four backward products are stride choices — none of the permutes of the reference's [L,B,C] layout exist here.
//
// 64x64x16 tile, 256 threads = 4 wave64 (2x2, one 32x32 MFMA block each), k-major LDS with +4 padding so the MFMA
// operand reads (64 consecutive floats) are conflict-free, register-staged global loads overlapped with the MFMAs.
// The load mapping follows the unit-stride axis of each operand (uniform branch) so global reads stay coalesced.
#include "rg_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int TB = 64, TK = 16, NT = 256, LD = TB + 4;

struct BgemmP {
    const float* A;
    const float* B;
    float* C;
    int M, N, K, batch1;
    int64_t a_ms, a_ks, b_ks, b_ns, c_ms, c_ns;
    int64_t a_b0, a_b1, b_b0, b_b1, c_b0, c_b1;
    float alpha, beta;
};

// stage a 64(rows r) x 16(k) block of an operand into regs; element (r, k) at base[r * rs + k * ks]
__device__ __forceinline__ void stage(const float* __restrict__ base, int64_t rs, int64_t ks, int r0, int k0, int R, int K,
                                      bool r_contig, int tid, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r, k;
        if (r_contig) {
            r = tid & 63;
            k = (tid >> 6) + 4 * i;
        } else {
            k = tid & 15;
            r = (tid >> 4) + 16 * i;
        }
        const int gr = r0 + r, gk = k0 + k;
        v[i] = (gr < R && gk < K) ? base[gr * rs + gk * ks] : 0.f;
    }
}

__device__ __forceinline__ void commit(float (*S)[LD], bool r_contig, int tid, const float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (r_contig)
            S[(tid >> 6) + 4 * i][tid & 63] = v[i];
        else
            S[tid & 15][(tid >> 4) + 16 * i] = v[i];
    }
}

__global__ __launch_bounds__(NT) void bgemm_kernel(const BgemmP p) {
    __shared__ float As[TK][LD];
    __shared__ float Bs[TK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l32 = lane & 31, kh = lane >> 5;
    const int b0 = blockIdx.z / p.batch1, b1 = blockIdx.z % p.batch1;
    const float* A = p.A + b0 * p.a_b0 + b1 * p.a_b1;
    const float* B = p.B + b0 * p.b_b0 + b1 * p.b_b1;
    float* C = p.C + b0 * p.c_b0 + b1 * p.c_b1;
    const int m0 = blockIdx.x * TB, n0 = blockIdx.y * TB;
    const bool a_rc = p.a_ms == 1, b_rc = p.b_ns == 1;

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float ra[4], rb[4];
    stage(A, p.a_ms, p.a_ks, m0, 0, p.M, p.K, a_rc, tid, ra);
    stage(B, p.b_ns, p.b_ks, n0, 0, p.N, p.K, b_rc, tid, rb);
    for (int k0 = 0; k0 < p.K; k0 += TK) {
        __syncthreads();
        commit(As, a_rc, tid, ra);
        commit(Bs, b_rc, tid, rb);
        __syncthreads();
        if (k0 + TK < p.K) {
            stage(A, p.a_ms, p.a_ks, m0, k0 + TK, p.M, p.K, a_rc, tid, ra);
            stage(B, p.b_ns, p.b_ks, n0, k0 + TK, p.N, p.K, b_rc, tid, rb);
        }
#pragma unroll
        for (int ks = 0; ks < TK / 2; ++ks) {
            const int k = 2 * ks + kh;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][wm * 32 + l32], Bs[k][wn * 32 + l32], acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + l32;
    if (n >= p.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < p.M) {
            float* c = C + m * p.c_ms + n * p.c_ns;
            const float v = p.alpha * acc[r];
            *c = p.beta != 0.f ? v + p.beta * *c : v;
        }
    }
}

// y[r][:] = softmax(scale * x[r][:]); one wave per row, in place allowed
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int rows,
                                                               int cols, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * cols;
    float* yr = y + (int64_t)row * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c] * scale);
    mx = rg_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(xr[c] * scale - mx);
    s = rg_wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < cols; c += 64) yr[c] = expf(xr[c] * scale - mx) * inv;
}

// ds = scale * p * (dp - sum_c dp * p); in place on dp allowed
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               float* __restrict__ ds, int rows, int cols, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + (int64_t)row * cols;
    const float* gr = dp + (int64_t)row * cols;
    float* dr = ds + (int64_t)row * cols;
    float dot = 0.f;
    for (int c = lane; c < cols; c += 64) dot += pr[c] * gr[c];
    dot = rg_wave_sum(dot);
    for (int c = lane; c < cols; c += 64) dr[c] = scale * pr[c] * (gr[c] - dot);
}

}  // namespace

extern "C" int rg_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t a_ms, int64_t a_ks,
                        int64_t b_ks, int64_t b_ns, int64_t c_ms, int64_t c_ns, int batch0, int batch1, int64_t a_b0,
                        int64_t a_b1, int64_t b_b0, int64_t b_b1, int64_t c_b0, int64_t c_b1, float alpha, float beta,
                        hipStream_t stream) {
    RG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch0 > 0 && batch1 > 0, "rg_bgemm: bad arguments");
    RG_REQUIRE((int64_t)batch0 * batch1 <= 65535, "rg_bgemm: more than 65535 batches");
    BgemmP p{A, B, C, M, N, K, batch1, a_ms, a_ks, b_ks, b_ns, c_ms, c_ns, a_b0, a_b1, b_b0, b_b1, c_b0, c_b1, alpha, beta};
    rg::ProfScope prof(rg::FAM_MISC, stream, 2.0 * M * N * K * batch0 * batch1, 0.0);
    hipLaunchKernelGGL(bgemm_kernel, dim3(rg::cdiv(M, TB), rg::cdiv(N, TB), batch0 * batch1), dim3(NT), 0, stream, p);
    return rg::check_launch("rg_bgemm");
}

extern "C" int rg_softmax_rows_fwd(const float* x, float* y, int rows, int cols, float scale, hipStream_t stream) {
    RG_REQUIRE(x && y && rows > 0 && cols > 0, "rg_softmax_rows_fwd: bad arguments");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 8.0 * rows * cols);
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(rg::cdiv(rows, 4)), dim3(256), 0, stream, x, y, rows, cols, scale);
    return rg::check_launch("rg_softmax_rows_fwd");
}

extern "C" int rg_softmax_rows_bwd(const float* p, const float* dp, float* ds, int rows, int cols, float scale,
                                   hipStream_t stream) {
    RG_REQUIRE(p && dp && ds && rows > 0 && cols > 0, "rg_softmax_rows_bwd: bad arguments");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 12.0 * rows * cols);
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(rg::cdiv(rows, 4)), dim3(256), 0, stream, p, dp, ds, rows, cols, scale);
    return rg::check_launch("rg_softmax_rows_bwd");
}
