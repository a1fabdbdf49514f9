// ImageProcessor stages on one 8-bit plane: 256-bin histogram, equalisation
// LUT, and a fused LUT -> separable fixed-point Gaussian -> centre-band mask
// tile kernel (reference src/image_processor.cpp:41-79).
#include "vo_internal.h"

namespace vo {

namespace {

constexpr int IP_THREADS = 256;
constexpr int IP_TW = 128, IP_TH = 16;
constexpr int IP_RMAX = 15;  // kernel size up to 31

__device__ __forceinline__ int ip_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// One block per band of rows; bins are accumulated in LDS and flushed once.
__global__ __launch_bounds__(IP_THREADS) void k_hist256(const uint8_t *__restrict__ img, int w, int h, int pitch,
                                                        int rows_per_block, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t bins[256];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const int y0 = blockIdx.x * rows_per_block;
    const int y1 = min(h, y0 + rows_per_block);
    const int w4 = w >> 2;
    for (int y = y0; y < y1; y++) {
        const uint8_t *row = img + (size_t)y * pitch;
        const uint32_t *row4 = reinterpret_cast<const uint32_t *>(row);  // plane rows are 128-byte aligned
        for (int x = threadIdx.x; x < w4; x += IP_THREADS) {
            const uint32_t v = row4[x];
            atomicAdd(&bins[v & 255u], 1u);
            atomicAdd(&bins[(v >> 8) & 255u], 1u);
            atomicAdd(&bins[(v >> 16) & 255u], 1u);
            atomicAdd(&bins[v >> 24], 1u);
        }
        for (int x = (w4 << 2) + threadIdx.x; x < w; x += IP_THREADS) atomicAdd(&bins[row[x]], 1u);
    }
    __syncthreads();
    if (bins[threadIdx.x]) atomicAdd(&hist[threadIdx.x], bins[threadIdx.x]);
}

// cv::equalizeHist's table: the first occupied bin maps to 0 and the rest to
// round(cumulative count * 255.f / (total - count of that bin)), in float.
__global__ __launch_bounds__(256) void k_equalize_lut(const uint32_t *__restrict__ hist, int total,
                                                      uint8_t *__restrict__ lut)
{
    __shared__ uint32_t cum[256];
    __shared__ int first;
    const int t = threadIdx.x;
    cum[t] = hist[t];
    __syncthreads();
    if (t == 0) {
        int i0 = 0;
        while (i0 < 255 && cum[i0] == 0) ++i0;
        first = i0;
        uint32_t run = 0;
        for (int i = 0; i < 256; i++) {
            const uint32_t c = cum[i];
            run = i > i0 ? run + c : 0;
            cum[i] = i == i0 ? c : run;  // bin i0 keeps its own count for the scale
        }
    }
    __syncthreads();
    const int i0 = first;
    const uint32_t n0 = cum[i0];
    uint8_t v;
    if ((int)n0 == total) {
        v = (uint8_t)i0;
    } else if (t <= i0) {
        v = 0;
    } else {
        const float scale = 255.f / (float)(total - (int)n0);
        const int r = __float2int_rn((float)cum[t] * scale);
        v = (uint8_t)min(max(r, 0), 255);
    }
    lut[t] = v;
}

struct GaussTaps {
    int n;
    uint16_t w[2 * IP_RMAX + 1];
};

// One 128x16 tile per block.  src is read through the optional LUT into LDS
// with BORDER_REFLECT_101 halos; rows are filtered into 8.8 fixed point, then
// columns into 16.16 and rounded -- cv::GaussianBlur's arithmetic for 8-bit
// images.  `enh` receives the LUT-mapped tile, `den` the blurred tile and
// `msk` the blurred tile with everything outside rows [h/4, 3h/4] zeroed.
__global__ __launch_bounds__(IP_THREADS) void k_enhance_blur_mask(const uint8_t *__restrict__ src, int w, int h,
                                                                 int pitch, const uint8_t *__restrict__ lut,
                                                                 GaussTaps taps, uint8_t *__restrict__ enh,
                                                                 uint8_t *__restrict__ den,
                                                                 uint8_t *__restrict__ msk)
{
    __shared__ uint8_t s_lut[256];
    __shared__ uint16_t s_tap[2 * IP_RMAX + 1];
    __shared__ uint8_t s_src[(IP_TH + 2 * IP_RMAX) * (IP_TW + 2 * IP_RMAX)];
    __shared__ uint16_t s_row[(IP_TH + 2 * IP_RMAX) * IP_TW];

    const int tid = threadIdx.x;
    const int r = taps.n >> 1;
    const int sw = IP_TW + 2 * r, sh = IP_TH + 2 * r;
    const int x0 = blockIdx.x * IP_TW, y0 = blockIdx.y * IP_TH;

    s_lut[tid] = lut ? lut[tid] : (uint8_t)tid;
    if (tid < taps.n) s_tap[tid] = taps.w[tid];
    __syncthreads();

    for (int i = tid; i < sw * sh; i += IP_THREADS) {
        const int ly = i / sw, lx = i - ly * sw;
        const int gx = ip_reflect101(min(x0 + lx - r, w - 1 + r), w);
        const int gy = ip_reflect101(min(y0 + ly - r, h - 1 + r), h);
        s_src[i] = s_lut[src[(size_t)gy * pitch + gx]];
    }
    __syncthreads();

    for (int i = tid; i < sh * IP_TW; i += IP_THREADS) {
        const int ly = i / IP_TW, lx = i - ly * IP_TW;
        const uint8_t *p = s_src + ly * sw + lx;
        uint32_t acc = 0;
        for (int k = 0; k < taps.n; k++) acc += (uint32_t)s_tap[k] * p[k];
        s_row[i] = (uint16_t)min(acc, 65535u);
    }
    __syncthreads();

    const int band0 = h / 4, band1 = h * 3 / 4;
    // four horizontally adjacent pixels per thread and pass: 32-bit stores
    for (int q = tid; q < IP_TW * IP_TH / 4; q += IP_THREADS) {
        const int ly = q / (IP_TW / 4), lx = (q - ly * (IP_TW / 4)) * 4;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        uint32_t acc[4] = {0, 0, 0, 0};
        for (int k = 0; k < taps.n; k++) {
            const uint32_t t = s_tap[k];
            const uint16_t *p = s_row + (ly + k) * IP_TW + lx;
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += t * p[j];
        }
        uint32_t packed = 0, epacked = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            packed |= min((acc[j] + 32768u) >> 16, 255u) << (8 * j);
            epacked |= (uint32_t)s_src[(ly + r) * sw + lx + r + j] << (8 * j);
        }
        // planes are padded to whole tiles, so the full quad may be stored
        const size_t o = (size_t)gy * pitch + gx;
        if (enh) *reinterpret_cast<uint32_t *>(enh + o) = epacked;
        if (den) *reinterpret_cast<uint32_t *>(den + o) = packed;
        if (msk) *reinterpret_cast<uint32_t *>(msk + o) = (gy >= band0 && gy <= band1) ? packed : 0u;
    }
}

// cv::undistort of a gray plane: the inverse map of initUndistortRectifyMap
// (K, D, R = I, P = K) evaluated per destination pixel in double, quantised to
// 1/32 px like the CV_16SC2 + CV_16UC1 maps, then cv::remap(INTER_LINEAR,
// BORDER_CONSTANT 0) with the 15-bit fixed-point bilinear weights.  A thread
// produces four neighbouring pixels of a row (one dword store); the taps are
// gathers through L1/L2 -- neighbouring threads read neighbouring source bytes.
struct UndistortParams {
    double fx, fy, cx, cy, ifx, ify;
    double k1, k2, p1, p2, k3, k4, k5, k6;
};

__global__ __launch_bounds__(IP_THREADS) void k_undistort(const uint8_t *__restrict__ src, int w, int h, int spitch,
                                                         UndistortParams P, uint8_t *__restrict__ dst, int dpitch)
{
    const int j0 = (blockIdx.x * 32 + (threadIdx.x & 31)) * 4;
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (i >= h || j0 >= w) return;
    const double y = (i - P.cy) * P.ify;
    const double y2 = y * y;
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = j0 + q;
        const double x = (j - P.cx) * P.ifx;
        const double x2 = x * x, r2 = x2 + y2, xy2 = 2 * x * y;
        const double kr = (1 + ((P.k3 * r2 + P.k2) * r2 + P.k1) * r2) / (1 + ((P.k6 * r2 + P.k5) * r2 + P.k4) * r2);
        const double xd = x * kr + P.p1 * xy2 + P.p2 * (r2 + 2 * x2);
        const double yd = y * kr + P.p1 * (r2 + 2 * y2) + P.p2 * xy2;
        double u = (P.fx * xd + P.cx) * 32, v = (P.fy * yd + P.cy) * 32;
        u = u < -2147483648.0 ? -2147483648.0 : (u > 2147483647.0 ? 2147483647.0 : u);
        v = v < -2147483648.0 ? -2147483648.0 : (v > 2147483647.0 ? 2147483647.0 : v);
        const int iu = (u != u) ? 0 : (int)rint(u), iv = (v != v) ? 0 : (int)rint(v);
        const int sx = (int)(int16_t)(iu >> 5), sy = (int)(int16_t)(iv >> 5), a = iu & 31, b = iv & 31;
        int acc = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int xx = sx + (t & 1), yy = sy + (t >> 1);
            const int wt = ((t & 1) ? a : 32 - a) * ((t >> 1) ? b : 32 - b) * 32;
            if (xx >= 0 && xx < w && yy >= 0 && yy < h) acc += (int)src[(size_t)yy * spitch + xx] * wt;
        }
        out |= (uint32_t)((acc + (1 << 14)) >> 15) << (8 * q);
    }
    *reinterpret_cast<uint32_t *>(dst + (size_t)i * dpitch + j0) = out;  // the pitch is padded to 128
}

}  // namespace

void launch_hist256(const uint8_t *img, int w, int h, int pitch, uint32_t *hist, hipStream_t s)
{
    (void)hipMemsetAsync(hist, 0, 256 * sizeof(uint32_t), s);
    const int rows = 4;
    hipLaunchKernelGGL(k_hist256, dim3((h + rows - 1) / rows), dim3(IP_THREADS), 0, s, img, w, h, pitch, rows, hist);
}

void launch_equalize_lut(const uint32_t *hist, int total, uint8_t *lut, hipStream_t s)
{
    hipLaunchKernelGGL(k_equalize_lut, dim3(1), dim3(256), 0, s, hist, total, lut);
}

void launch_enhance_blur_mask(const uint8_t *src, int w, int h, int pitch, const uint8_t *lut, int n_taps,
                              const uint16_t *taps, uint8_t *enh, uint8_t *den, uint8_t *msk, hipStream_t s)
{
    GaussTaps t{};
    t.n = n_taps;
    for (int i = 0; i < n_taps; i++) t.w[i] = taps[i];
    hipLaunchKernelGGL(k_enhance_blur_mask, dim3((w + IP_TW - 1) / IP_TW, (h + IP_TH - 1) / IP_TH),
                       dim3(IP_THREADS), 0, s, src, w, h, pitch, lut, t, enh, den, msk);
}

void launch_undistort(const uint8_t *src, int w, int h, int spitch, double fx, double fy, double cx, double cy,
                      const double *k, uint8_t *dst, int dpitch, hipStream_t s)
{
    UndistortParams P{fx, fy, cx, cy, 1.0 / fx, 1.0 / fy, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]};
    hipLaunchKernelGGL(k_undistort, dim3((w + 127) / 128, (h + 7) / 8), dim3(IP_THREADS), 0, s, src, w, h, spitch, P,
                       dst, dpitch);
}

}  // namespace vo
