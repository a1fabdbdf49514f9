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

}  // namespace vo
