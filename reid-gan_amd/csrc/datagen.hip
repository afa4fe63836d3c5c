// On-device input synthesis: what the reference's CPU data pipeline does per sample with PIL / scipy,
// as batch kernels on tensors that already live in HBM.  All HBM-bound (one write of the output, no re-reads):
//   pose heat maps, FD-GAN form     FD/reid/utils/data/preprocessor.py:114-131  (_generate_pose_map: delta at the landmark
//                                   -> scipy.ndimage gaussian_filter(sigma, truncate 4, mode 'reflect') -> / max)
//   pose heat maps, dual_gan form   CC/clustercontrast/utils/data/pose_utils.py:51-70  (cords_to_map: exp(-d^2 / 2 sigma^2))
//   horizontal flip + Pad + RandomCrop   FD preprocessor.py:88-91 (np.flip(maps, 2)), CC/examples/...infomap.py:115-116
//   RandomErasing rectangle fill    CC/clustercontrast/utils/data/transforms.py:52-96
// The random draws (which joint to erase, sigma, rectangles, offsets, flips) stay on the host in the reference's own
// order (`random` module), so a seeded run selects the same augmentations; the kernels apply them to the whole batch.
#include "rg_common.h"

namespace {

constexpr int MAXDIM = 1024;      // H, W <= MAXDIM (profiles live in LDS)

// scipy's 1-D Gaussian correlate of a unit impulse at `c` on [0, n) with 'reflect' extension (d c b a | a b c d | d c b a):
// out[i] = sum_{k=-r..r} w[k] * [reflect(i + k) == c],  w[k] = exp(-k^2 / (2 sigma^2)) / sum_k(...)   (float64, as numpy)
__device__ double impulse_response(int i, int c, int n, int r, double sigma, double wsum) {
    double v = 0.0;
    const double inv = -0.5 / (sigma * sigma);
    int k = c - i;                                  // inside the array
    if (k >= -r && k <= r) v += exp(inv * (double)k * (double)k) / wsum;
    k = -1 - c - i;                                 // mirror image below 0: i + k = -1 - c
    if (k >= -r && k <= r && i + k < 0) v += exp(inv * (double)k * (double)k) / wsum;
    k = 2 * n - 1 - c - i;                          // mirror image above n-1: i + k = 2n - 1 - c
    if (k >= -r && k <= r && i + k >= n) v += exp(inv * (double)k * (double)k) / wsum;
    return v;
}

// grid (J, N): one workgroup per (sample, joint).  mode 0: FD-GAN form, mode 1: plain Gaussian.
__global__ __launch_bounds__(256) void pose_maps_kernel(const int* __restrict__ centers, const float* __restrict__ sigma,
                                                        float* __restrict__ out, int J, int H, int W, int mode) {
    __shared__ double fy[MAXDIM], fx[MAXDIM];
    __shared__ double red[8];
    const int j = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cy = centers[((int64_t)n * J + j) * 2 + 0], cx = centers[((int64_t)n * J + j) * 2 + 1];
    float* o = out + ((int64_t)n * J + j) * H * W;
    // missing / erased joint -> zeros (uniform per block).  mode 0 indexes the map at the landmark, so anything outside is
    // missing; mode 1 is a closed form that the reference also evaluates for centres outside the image (after the
    // affine map), so only the sentinel INT_MIN marks a missing joint there
    const bool missing = mode == 0 ? (cy < 0 || cx < 0 || cy >= H || cx >= W) : (cy == INT32_MIN || cx == INT32_MIN);
    if (missing) {
        for (int i = tid; i < H * W; i += 256) o[i] = 0.f;
        return;
    }
    const double sg = (double)sigma[n];
    if (mode == 0) {
        const int r = (int)(4.0 * sg + 0.5);                 // scipy: int(truncate * sd + 0.5)
        // kernel normalisation sum_k exp(-k^2 / 2 sigma^2), summed in scipy's order (k ascending) by one thread
        if (tid == 0) {
            double acc = 0.0;
            for (int k = -r; k <= r; ++k) acc += exp(-0.5 / (sg * sg) * (double)k * (double)k);
            red[0] = acc;
        }
        __syncthreads();
        const double wsum = red[0];
        __syncthreads();
        for (int i = tid; i < H; i += 256) fy[i] = impulse_response(i, cy, H, r, sg, wsum);
        for (int i = tid; i < W; i += 256) fx[i] = impulse_response(i, cx, W, r, sg, wsum);
        __syncthreads();
        // map.max() = max(fy) * max(fx) (all entries >= 0)
        double my = 0.0, mx = 0.0;
        for (int i = tid; i < H; i += 256) my = fmax(my, fy[i]);
        for (int i = tid; i < W; i += 256) mx = fmax(mx, fx[i]);
        for (int off = 32; off > 0; off >>= 1) {
            my = fmax(my, __shfl_xor(my, off, 64));
            mx = fmax(mx, __shfl_xor(mx, off, 64));
        }
        if ((tid & 63) == 0) {
            red[tid >> 6] = my;
            red[4 + (tid >> 6)] = mx;
        }
        __syncthreads();
        my = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        mx = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
        const double inv = 1.0 / (my * mx);
        for (int i = tid; i < H * W; i += 256) {
            const int y = i / W, x = i - y * W;
            o[i] = (float)(fy[y] * fx[x] * inv);
        }
    } else {
        // exp(-(dy^2 + dx^2) / 2 sigma^2) as the product of two float64 profiles (one exp per row / column instead of
        // one per pixel; differs from the single exp by ~1e-16 relative, invisible after the float32 rounding)
        const double inv = -1.0 / (2.0 * sg * sg);
        for (int i = tid; i < H; i += 256) fy[i] = exp((double)(i - cy) * (double)(i - cy) * inv);
        for (int i = tid; i < W; i += 256) fx[i] = exp((double)(i - cx) * (double)(i - cx) * inv);
        __syncthreads();
        for (int i = tid; i < H * W; i += 256) {
            const int y = i / W, x = i - y * W;
            o[i] = (float)(fy[y] * fx[x]);
        }
    }
}

// out[n][c][y][x] = P[n][c][top + y][left + (flip ? W - 1 - x : x)],  P = x padded by `pad` pixels of pad_value[c]
__global__ __launch_bounds__(256) void flip_pad_crop_kernel(const float* __restrict__ x, const int* __restrict__ params,
                                                            const float* __restrict__ pad_value, float* __restrict__ out,
                                                            int C, int Hs, int Ws, int H, int W, int pad, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int xo = (int)(i % W);
        int64_t t = i / W;
        const int yo = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const int flip = params[n * 3 + 0], top = params[n * 3 + 1], left = params[n * 3 + 2];
        const int ys = top + yo - pad;
        const int xs = left + (flip ? W - 1 - xo : xo) - pad;
        float v = pad_value ? pad_value[c] : 0.f;
        if ((unsigned)ys < (unsigned)Hs && (unsigned)xs < (unsigned)Ws) v = x[(((int64_t)n * C + c) * Hs + ys) * Ws + xs];
        out[i] = v;
    }
}

// in place: x[n][c][y1 : y1 + h][x1 : x1 + w] = fill[c]  for rects[n] = (y1, x1, h, w), h = 0: untouched
__global__ __launch_bounds__(256) void erase_rects_kernel(float* __restrict__ x, const int* __restrict__ rects,
                                                          const float* __restrict__ fill, int C, int H, int W) {
    const int n = blockIdx.y, c = blockIdx.z;
    const int y1 = rects[n * 4 + 0], x1 = rects[n * 4 + 1], h = rects[n * 4 + 2], w = rects[n * 4 + 3];
    if (h <= 0 || w <= 0) return;
    float* p = x + ((int64_t)n * C + c) * H * W;
    const float v = fill[c];
    if (v != v) return;                               // NaN: this channel is left as it is
    for (int i = blockIdx.x * 256 + threadIdx.x; i < h * w; i += gridDim.x * 256) {
        const int dy = i / w, dx = i - dy * w;
        const int yy = y1 + dy, xx = x1 + dx;
        if (yy < H && xx < W) p[yy * W + xx] = v;
    }
}

}  // namespace

extern "C" int rg_pose_maps(const int* centers, const float* sigma, float* out, int N, int J, int H, int W, int mode,
                            hipStream_t stream) {
    RG_REQUIRE(centers && sigma && out && N > 0 && J > 0 && H > 0 && W > 0, "rg_pose_maps: bad arguments");
    RG_REQUIRE(H <= MAXDIM && W <= MAXDIM, "rg_pose_maps: H, W must be <= %d", MAXDIM);
    RG_REQUIRE(mode == 0 || mode == 1, "rg_pose_maps: mode must be 0 (filtered impulse, FD-GAN) or 1 (plain Gaussian)");
    RG_REQUIRE(N <= 65535, "rg_pose_maps: N > 65535");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 4.0 * N * (double)J * H * W);
    hipLaunchKernelGGL(pose_maps_kernel, dim3(J, N), dim3(256), 0, stream, centers, sigma, out, J, H, W, mode);
    return rg::check_launch("rg_pose_maps");
}

extern "C" int rg_flip_pad_crop(const float* x, const int* params, const float* pad_value, float* out, int N, int C, int Hs,
                                int Ws, int H, int W, int pad, hipStream_t stream) {
    RG_REQUIRE(x && params && out && N > 0 && C > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && pad >= 0,
               "rg_flip_pad_crop: bad arguments");
    RG_REQUIRE(H <= Hs + 2 * pad && W <= Ws + 2 * pad, "rg_flip_pad_crop: crop %dx%d larger than the padded image", H, W);
    const int64_t total = (int64_t)N * C * H * W;
    int64_t g = rg::cdiv64(total, 256);
    if (g > 16384) g = 16384;
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 8.0 * total);
    hipLaunchKernelGGL(flip_pad_crop_kernel, dim3((unsigned)g), dim3(256), 0, stream, x, params, pad_value, out, C, Hs, Ws, H, W,
                       pad, total);
    return rg::check_launch("rg_flip_pad_crop");
}

extern "C" int rg_erase_rects(float* x, const int* rects, const float* fill, int N, int C, int H, int W, hipStream_t stream) {
    RG_REQUIRE(x && rects && fill && N > 0 && C > 0 && H > 0 && W > 0, "rg_erase_rects: bad arguments");
    RG_REQUIRE(N <= 65535 && C <= 65535, "rg_erase_rects: N, C must be <= 65535");
    rg::ProfScope prof(rg::FAM_MISC, stream, 0.0, 4.0 * N * (double)C * H * W * 0.2);
    hipLaunchKernelGGL(erase_rects_kernel, dim3(rg::cdiv(H * W, 256 * 8), N, C), dim3(256), 0, stream, x, rects, fill, C, H, W);
    return rg::check_launch("rg_erase_rects");
}
