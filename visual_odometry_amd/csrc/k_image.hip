// Image-side kernels of the detector for gfx950: BGR->gray, the fused tile
// kernel (gray + FAST-9/16 pre-test, ballot-compacted scoring, non-max
// suppression, candidate emission and 7x7 Gaussian), top-N selection and
// rBRIEF-256.
//
// All arithmetic here is 8/16/32-bit integer and bit-exact by construction;
// the work is HBM/LDS/VALU integer work, there is nothing MFMA-shaped.
#include <stdlib.h>

#include "vo_internal.h"


namespace vo {

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ int reflect101(int p, int len)
{
    // BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ...
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ uint32_t gray_of(uint32_t b, uint32_t g, uint32_t r, GrayCoeffs c)
{
    return (b * c.b + g * c.g + r * c.r + (1u << (c.shift - 1))) >> c.shift;
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

// --------------------------------------------------------------------- gray

// One thread converts four horizontally adjacent pixels and stores one dword.
// reference src/frame_processor.cpp:204-208 (cv::cvtColor BGR2GRAY)
template <int CN>
__global__ __launch_bounds__(256) void k_gray(const uint8_t *__restrict__ src, int width, int height,
                                              int src_stride, size_t src_frame_stride, PlaneBatch dst,
                                              GrayCoeffs c)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = gx * 4;
    if (y >= height || x >= width) return;
    const uint8_t *row = src + blockIdx.z * src_frame_stride + (size_t)y * src_stride;
    uint8_t *drow = dst.base + blockIdx.z * dst.plane_stride + (size_t)y * dst.pitch;
    uint32_t out = 0;
    if (CN == 4 && ((((uintptr_t)row) & 15) == 0) && x + 3 < width) {
        const uint4 v = *reinterpret_cast<const uint4 *>(row + (size_t)x * 4);
        const uint32_t p[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++)
            out |= gray_of(p[i] & 255u, (p[i] >> 8) & 255u, (p[i] >> 16) & 255u, c) << (8 * i);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x + i < width) {
                const uint8_t *p = row + (size_t)(x + i) * CN;
                out |= gray_of(p[0], p[1], p[2], c) << (8 * i);
            }
        }
    }
    *reinterpret_cast<uint32_t *>(drow + x) = out;  // pitch is padded: always in bounds
}

void launch_gray(const uint8_t *src, int width, int height, int src_stride, size_t src_frame_stride,
                 int channels, PlaneBatch dst, int batch, GrayCoeffs c, hipStream_t s)
{
    dim3 grid((width + 255) / 256, (height + 3) / 4, batch);
    if (channels == 4)
        hipLaunchKernelGGL(k_gray<4>, grid, dim3(256), 0, s, src, width, height, src_stride,
                           src_frame_stride, dst, c);
    else
        hipLaunchKernelGGL(k_gray<3>, grid, dim3(256), 0, s, src, width, height, src_stride,
                           src_frame_stride, dst, c);
}

// ----------------------- fused gray + FAST + NMS + emit + blur (one tile)

constexpr int FE_TW = 128;             // interior tile width
constexpr int FE_TH = kTileH;          // interior tile height
constexpr int FE_PITCH = FE_TW + 16;   // LDS row: image x0-4 .. x0+TW+11, 16-B aligned
constexpr int FE_GROWS = FE_TH + 8;    // gray rows: image y0-4 .. y0+TH+3
constexpr int FE_SROWS = FE_TH + 2;    // score rows: image y0-1 .. y0+TH
constexpr int FE_HROWS = FE_TH + 6;    // row-sum rows: image y0-3 .. y0+TH+2
constexpr int FE_THREADS = 512;
constexpr int FE_WAVES = FE_THREADS / 64;
constexpr int FE_HALO_PIX = 2 * (FE_TW + 2) + 2 * FE_TH;             // 1-px rim around the tile
constexpr int FE_QCAP = FE_TW * FE_TH + FE_HALO_PIX;                  // queue entries a tile can produce
static_assert(FE_TW == kTileW && FE_TH == kTileH, "candidate lists are indexed by this tile grid");
constexpr int FE_QPT = FE_TW / 4 * FE_TH / FE_THREADS;  // quads (four pixels of a row) per thread: rows gy + 16 k
static_assert(FE_THREADS * FE_QPT == FE_TW / 4 * FE_TH && FE_HALO_PIX <= FE_THREADS, "whole quads and one rim pixel per thread");

// FAST-9/16 score of one pixel from its 16 ring values p[] and centre v:
//   A = max over the 16 arcs of 9 of min(v - p) = v - min_arcs(max_arc p)
//   B = max over arcs of min(p - v)           = max_arcs(min_arc p) - v
//   score = max(A, B) - 1 when that is >= threshold (the pixel is then a
//   corner for every t <= score), else 0.
// Same value as cv::FAST's cornerScore<16> (reference src/feature_detector.cpp:59).
__device__ __forceinline__ uint32_t fast_score(const uint32_t (&p)[16], uint32_t v, int thr)
{
    uint32_t mx3[16], mn3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx3[k] = umax3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        mn3[k] = umin3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
    }
    // per arc of 9: max / min of its three triples; then the min over arcs of the max and the max over arcs of
    // the min, reduced three at a time
    uint32_t mx9[16], mn9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx9[k] = umax3(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
        mn9[k] = umin3(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
    }
    uint32_t tl[5], th[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        tl[k] = umin3(mx9[3 * k], mx9[3 * k + 1], mx9[3 * k + 2]);
        th[k] = umax3(mn9[3 * k], mn9[3 * k + 1], mn9[3 * k + 2]);
    }
    const uint32_t lo = umin3(umin3(tl[0], tl[1], tl[2]), umin3(tl[3], tl[4], mx9[15]), 255u);
    const uint32_t hi = umax3(umax3(th[0], th[1], th[2]), umax3(th[3], th[4], mn9[15]), 0u);
    const int a = (int)v - (int)lo, b = (int)hi - (int)v;
    const int sc = max(a, b) - 1;
    return sc >= thr ? (uint32_t)sc : 0u;
}

// Necessary condition for a 9-arc on the 16-ring: the arc contains at least
// one pixel of every opposite pair, so both the vertical pair (p0, p8) and the
// horizontal pair (p4, p12) must hold a pixel brighter than v+t, or both a
// pixel darker than v-t.  Rejects about 92 % of the pixels of a textured frame.
__device__ __forceinline__ bool fast_quick(uint32_t v, uint32_t p0, uint32_t p8, uint32_t p4, uint32_t p12, int thr)
{
    const int hi = (int)v + thr, lo = (int)v - thr;
    const bool b = (int)max(p0, p8) > hi && (int)max(p4, p12) > hi;
    const bool d = (int)min(p0, p8) < lo && (int)min(p4, p12) < lo;
    return b || d;
}

#ifdef VO_FE_CUT
// Diagnostic build only: the kernel ends after barrier VO_FE_CUT with an empty
// corner list, so that hardware counters can be attributed to its phases.
#define FE_CUT(n)                                                                                          \
    do {                                                                                                   \
        if (VO_FE_CUT == (n)) {                                                                            \
            if (cand && threadIdx.x == 0)                                                                  \
                tile_count[(size_t)tile.frame * tiles_x * tiles_y + (tile.y0 / FE_TH) * tiles_x + tile.x0 / FE_TW] = 0; \
            return;                                                                                        \
        }                                                                                                  \
    } while (0)
#else
#define FE_CUT(n) do { } while (0)
#endif

#ifdef VO_FE_STAMPS
// Diagnostic build only: cycle stamps of wave 0 of a window of blocks.
constexpr int FE_STAMP_N = 16, FE_STAMP_BLOCKS = 4096, FE_STAMP_FIRST = 40000;
__device__ unsigned long long g_fe_stamps[FE_STAMP_BLOCKS * FE_STAMP_N];
#define FE_STAMP(i)                                                                                   \
    do {                                                                                              \
        const unsigned lb_ = blockIdx.x + blockIdx.y * gridDim.x;                                     \
        if (threadIdx.x == 0 && lb_ >= FE_STAMP_FIRST && lb_ < FE_STAMP_FIRST + FE_STAMP_BLOCKS)       \
            g_fe_stamps[(lb_ - FE_STAMP_FIRST) * FE_STAMP_N + (i)] = __builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define FE_STAMP(i) do { } while (0)
#endif

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }

// bytes 2 of four sums -> one dword
__device__ __forceinline__ uint32_t pack_byte2(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3)
{
    return __builtin_amdgcn_perm(s1, s0, 0x0c0c0602u) | __builtin_amdgcn_perm(s3, s2, 0x06020c0cu);
}

// bytes A and B (0..3 from lo, 4..7 from hi) of the 8-byte pool {hi, lo}, zero-extended to two u16
template <int A, int B>
__device__ __forceinline__ us2 pair_of(uint32_t hi, uint32_t lo)
{
    return as_us2(__builtin_amdgcn_perm(hi, lo, (uint32_t)A | 0x0c00u | ((uint32_t)B << 16) | 0x0c000000u));
}

// fast_quick() for the two pixels at window bytes 4+I and 5+I of the middle
// row, in packed 16-bit arithmetic.  d0..d2: the middle row's 12-byte window,
// up1/dn1: the dword above/below d1 at distance 3.  With
//   M = min(max(p0,p8), max(p4,p12)),  m = max(min(p0,p8), min(p4,p12))
// the test "both pairs hold a pixel > v+t, or both a pixel < v-t" is
// max(M - v, v - m) > t (saturating differences).  Returns per 16-bit half a
// non-zero value when the pixel passes.
template <int I>
__device__ __forceinline__ uint32_t quick_pair(uint32_t up1, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t dn1,
                                               us2 tt)
{
    const us2 v = pair_of<I, I + 1>(0u, d1);
    const us2 p0 = pair_of<I, I + 1>(0u, dn1);
    const us2 p8 = pair_of<I, I + 1>(0u, up1);
    const us2 p4 = pair_of<3 + I, 4 + I>(d2, d1);
    const us2 p12 = pair_of<1 + I, 2 + I>(d1, d0);
    const us2 M = __builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12));
    const us2 m = __builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12));
    const us2 e = __builtin_elementwise_max(__builtin_elementwise_sub_sat(M, v), __builtin_elementwise_sub_sat(v, m));
    return as_u32(__builtin_elementwise_sub_sat(e, tt));
}

// 7-tap row sums [18,34,49,55,49,34,18] of the four pixels at window bytes
// 4..7, two 4-byte dot products each.  Pixels 0 and 1 read the same two dwords
// (window bytes 1..8) with the taps shifted by one byte, pixels 2 and 3 likewise
// (bytes 3..10): four byte alignments for the quad.
__device__ __forceinline__ void hsum7_quad(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t (&h)[4])
{
    constexpr uint32_t WA = 18u | (34u << 8) | (49u << 16) | (55u << 24);  // taps 0..3 | window starts at tap 0
    constexpr uint32_t WB = 49u | (34u << 8) | (18u << 16);               // taps 4..6, -
    constexpr uint32_t XA = (18u << 8) | (34u << 16) | (49u << 24);       // -, taps 0..2 | window starts one byte early
    constexpr uint32_t XB = 55u | (49u << 8) | (34u << 16) | (18u << 24);  // taps 3..6
    const uint32_t a1 = __builtin_amdgcn_alignbyte(d1, d0, 1), b1 = __builtin_amdgcn_alignbyte(d2, d1, 1);
    const uint32_t a3 = __builtin_amdgcn_alignbyte(d1, d0, 3), b3 = __builtin_amdgcn_alignbyte(d2, d1, 3);
    h[0] = __builtin_amdgcn_udot4(a1, WA, __builtin_amdgcn_udot4(b1, WB, 0u, false), false);
    h[1] = __builtin_amdgcn_udot4(a1, XA, __builtin_amdgcn_udot4(b1, XB, 0u, false), false);
    h[2] = __builtin_amdgcn_udot4(a3, WA, __builtin_amdgcn_udot4(b3, WB, 0u, false), false);
    h[3] = __builtin_amdgcn_udot4(a3, XA, __builtin_amdgcn_udot4(b3, XB, 0u, false), false);
}

constexpr int FE_TASKS = FE_GROWS * (FE_PITCH / 4);  // 4-pixel staging tasks per tile
constexpr int FE_TPT = (FE_TASKS + FE_THREADS - 1) / FE_THREADS;  // tasks per thread
constexpr int FE_HPAIRS = FE_HROWS / 2;  // row pairs of row sums
static_assert(kTileCap == FE_TW * FE_TH / 4, "a tile emits at most one corner per 2x2 block");
static_assert(FE_HROWS % 2 == 0 && FE_TH % 2 == 0, "rows are processed in pairs");

// Pixels of one staging task (4 horizontally adjacent pixels), still in the
// layout they were loaded in.
template <int CN>
struct FePre {
    uint32_t v[CN == 3 ? 12 : 4];
};

struct FeTile {
    int frame, x0, y0;
    bool inside;  // the tile and its apron lie inside the image: no reflection
    bool whole;   // every pixel of the tile lies where corners are looked for: no masking
    bool fast;    // border tile of an aligned image of at least 8 x 8: branch-free reflection, vector loads
};

// gray weights split into byte planes for 4-byte dot products:
// gray = (dot4(px, lo) + (dot4(px, hi) << 8) + round) >> shift.
// When the weights scaled by 2^(16 - shift) still fit 16 bits (the usual 14- and 15-bit tables do), the planes hold
// the scaled weights instead: the sum is then gray << 16 plus a fraction and the result is its byte 2 -- no final
// shift, and four results are packed with byte permutes (byte2 is set).
struct GrayDot {
    uint32_t lo, hi, round;
    int shift;
    bool byte2;
};
__device__ __forceinline__ GrayDot gray_dot(GrayCoeffs c)
{
    GrayDot g;
    uint32_t cb = (uint32_t)c.b, cg = (uint32_t)c.g, cr = (uint32_t)c.r;
    g.round = 1u << (c.shift - 1);
    g.shift = c.shift;
    g.byte2 = false;
    if (c.shift <= 16) {
        const int up = 16 - c.shift;
        if (((cb << up) | (cg << up) | (cr << up)) < 65536u && c.b >= 0 && c.g >= 0 && c.r >= 0) {
            cb <<= up; cg <<= up; cr <<= up;
            g.round = 1u << 15;
            g.shift = 16;
            g.byte2 = true;
        }
    }
    g.lo = (cb & 255u) | ((cg & 255u) << 8) | ((cr & 255u) << 16);
    g.hi = ((cb >> 8) & 255u) | (((cg >> 8) & 255u) << 8) | (((cr >> 8) & 255u) << 16);
    return g;
}
// px holds B, G, R in its three low bytes (the fourth byte is ignored): the weighted sum before the final shift
__device__ __forceinline__ uint32_t gray_dot_sum(uint32_t px, GrayDot g)
{
    const uint32_t lo = __builtin_amdgcn_udot4(px, g.lo, g.round, false);
    const uint32_t hi = __builtin_amdgcn_udot4(px, g.hi, 0u, false);
    return lo + (hi << 8);
}
__device__ __forceinline__ uint32_t gray_dot_px(uint32_t px, GrayDot g) { return gray_dot_sum(px, g) >> g.shift; }

// BORDER_REFLECT_101 for positions at most len - 1 outside [0, len), clamped
// beyond that: branch-free.  The tile apron reaches 4 px outside the image and
// the padding of the last tile column/row further, but nothing that is kept
// depends on pixels more than 3 px outside.
__device__ __forceinline__ int reflect101_near(int p, int len)
{
    p = max(p, -p);
    p = p >= len ? 2 * len - 2 - p : p;
    return min(max(p, 0), len - 1);
}

// How a tile is staged: FE_INSIDE -- tile and apron inside an aligned image, plain vector loads;
// FE_BORDER -- border tile of an aligned image of at least 8 x 8, branch-free reflection;
// FE_GENERAL -- anything else (unaligned rows, tiny images): per-pixel loads, iterated reflection.
enum { FE_INSIDE = 0, FE_BORDER = 1, FE_GENERAL = 2 };

template <int CN>
__device__ __forceinline__ void fe_load_packed(FePre<CN> &p, const uint8_t *row, int xb)
{
    if (CN == 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + (size_t)xb * 4);
        p.v[0] = q.x; p.v[1] = q.y; p.v[2] = q.z; p.v[3] = q.w;
    } else if (CN == 1) {
        p.v[0] = *reinterpret_cast<const uint32_t *>(row + xb);
    } else {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(row + (size_t)xb * 3);
        p.v[0] = q[0]; p.v[1] = q[1]; p.v[2] = q[2];
    }
}

template <int CN>
__device__ __forceinline__ void fe_load_pixel(FePre<CN> &p, const uint8_t *row, int j, int xx)
{
    if (CN == 4) p.v[j] = *reinterpret_cast<const uint32_t *>(row + (size_t)xx * 4);
    else if (CN == 1) p.v[j] = row[xx];
    else { p.v[3 * j] = row[xx * 3]; p.v[3 * j + 1] = row[xx * 3 + 1]; p.v[3 * j + 2] = row[xx * 3 + 2]; }
}

// issue the global loads of staging task (tr, tq) of a tile (no wait); returns whether the pixels arrive packed
template <int CN, int PATH>
__device__ __forceinline__ bool fe_issue(FePre<CN> &p, const uint8_t *img, int src_stride, int width, int height,
                                         const FeTile &t, int tr, int tq)
{
    const int yb = t.y0 - 4 + tr, xb = t.x0 - 4 + tq * 4;
    if (PATH == FE_INSIDE) {
        fe_load_packed<CN>(p, img + (size_t)yb * src_stride, xb);
        return true;
    }
    if (PATH == FE_BORDER) {
        const uint8_t *row = img + (size_t)reflect101_near(yb, height) * src_stride;
        if (xb >= 0 && xb + 4 <= width) {  // inside the image in x: only the row is reflected
            fe_load_packed<CN>(p, row, xb);
            return true;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) fe_load_pixel<CN>(p, row, j, reflect101_near(xb + j, width));
        return false;
    }
    const uint8_t *row = img + (size_t)reflect101(yb, height) * src_stride;
#pragma unroll
    for (int j = 0; j < 4; j++) fe_load_pixel<CN>(p, row, j, reflect101(xb + j, width));
    return false;
}

// four gray bytes of a loaded task
template <int CN>
__device__ __forceinline__ uint32_t fe_gray4(const FePre<CN> &p, bool packed, GrayDot g)
{
    uint32_t out = 0;
    if (CN == 4) {
        if (g.byte2)
            return pack_byte2(gray_dot_sum(p.v[0], g), gray_dot_sum(p.v[1], g), gray_dot_sum(p.v[2], g),
                              gray_dot_sum(p.v[3], g));
#pragma unroll
        for (int j = 0; j < 4; j++) out |= gray_dot_px(p.v[j], g) << (8 * j);
    } else if (CN == 1) {
        if (packed) out = p.v[0];
        else out = p.v[0] | (p.v[1] << 8) | (p.v[2] << 16) | (p.v[3] << 24);
    } else if (packed) {
        // 12 bytes B0 G0 R0 B1 ... R3 in three dwords: pixel j starts at byte 3j
        const uint32_t d0 = p.v[0], d1 = p.v[1], d2 = p.v[2];
        if (g.byte2)
            return pack_byte2(gray_dot_sum(d0, g), gray_dot_sum(__builtin_amdgcn_alignbyte(d1, d0, 3), g),
                              gray_dot_sum(__builtin_amdgcn_alignbyte(d2, d1, 2), g), gray_dot_sum(d2 >> 8, g));
        out = gray_dot_px(d0, g) | (gray_dot_px(__builtin_amdgcn_alignbyte(d1, d0, 3), g) << 8) |
              (gray_dot_px(__builtin_amdgcn_alignbyte(d2, d1, 2), g) << 16) | (gray_dot_px(d2 >> 8, g) << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            out |= gray_dot_px(p.v[3 * j] | (p.v[3 * j + 1] << 8) | (p.v[3 * j + 2] << 16), g) << (8 * j);
    }
    return out;
}

// Stage the tile and its apron as 8-bit gray in LDS (all global loads of a thread are issued before the first
// use) and clear the score tile and the block's counters.
template <int CN, int PATH>
__device__ __forceinline__ void fe_stage(uint8_t *s_gray, uint8_t *s_score, uint32_t *s_counters, const uint8_t *img,
                                         int src_stride, int width, int height, const FeTile &tile, GrayCoeffs c)
{
    const int tid = threadIdx.x;
    if constexpr (CN == 1 && PATH == FE_INSIDE) {
        // gray input, tile and apron inside the image: the rows are copied as 16-byte pieces (4-byte aligned in
        // global memory, 16-byte aligned in LDS), one piece per thread
        constexpr int PPR = FE_PITCH / 16, NP = FE_GROWS * PPR, PPT = (NP + FE_THREADS - 1) / FE_THREADS;
        static_assert(FE_PITCH % 16 == 0, "whole 16-byte pieces");
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        u32x4_a4 v[PPT];
        int at[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const int i = min(tid + u * FE_THREADS, NP - 1);
            const int tr = i / PPR, tp = i - tr * PPR;
            at[u] = tr * FE_PITCH + 16 * tp;
            v[u] = *reinterpret_cast<const u32x4_a4 *>(img + (size_t)(tile.y0 - 4 + tr) * src_stride +
                                                       (tile.x0 - 4 + 16 * tp));
        }
        for (int i = tid; i < FE_SROWS * FE_PITCH / 16; i += FE_THREADS) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0, 0, 0, 0);
        if (tid < 3) s_counters[tid] = 0;
        FE_STAMP(1);
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (tid + u * FE_THREADS < NP)
                *reinterpret_cast<uint4 *>(&s_gray[at[u]]) = make_uint4(v[u].x, v[u].y, v[u].z, v[u].w);
        return;
    }
    if constexpr (CN == 1 && PATH == FE_BORDER) {
        // gray input, border tile of an aligned image: the same 16-byte pieces with the row reflected; a piece
        // that crosses the left or right image edge is gathered byte by byte
        constexpr int PPR = FE_PITCH / 16, NP = FE_GROWS * PPR, PPT = (NP + FE_THREADS - 1) / FE_THREADS;
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        uint32_t w[PPT][4];
        int at[PPT];
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const int i = min(tid + u * FE_THREADS, NP - 1);
            const int tr = i / PPR, tp = i - tr * PPR;
            at[u] = tr * FE_PITCH + 16 * tp;
            const uint8_t *row = img + (size_t)reflect101_near(tile.y0 - 4 + tr, height) * src_stride;
            const int xb = tile.x0 - 4 + 16 * tp;
            if (xb >= 0 && xb + 16 <= width) {
                const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(row + xb);
                w[u][0] = v.x; w[u][1] = v.y; w[u][2] = v.z; w[u][3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    w[u][q] = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        w[u][q] |= (uint32_t)row[reflect101_near(xb + 4 * q + j, width)] << (8 * j);
                }
            }
        }
        for (int i = tid; i < FE_SROWS * FE_PITCH / 16; i += FE_THREADS) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0, 0, 0, 0);
        if (tid < 3) s_counters[tid] = 0;
        FE_STAMP(1);
#pragma unroll
        for (int u = 0; u < PPT; u++)
            if (tid + u * FE_THREADS < NP)
                *reinterpret_cast<uint4 *>(&s_gray[at[u]]) = make_uint4(w[u][0], w[u][1], w[u][2], w[u][3]);
        return;
    }
    FePre<CN> pre[FE_TPT];
    bool packed[FE_TPT];
    int at[FE_TPT];
    const GrayDot gd = gray_dot(c);
#pragma unroll
    for (int u = 0; u < FE_TPT; u++) {
        const int i = tid + u * FE_THREADS;
        packed[u] = false;
        const int tr = i / (FE_PITCH / 4), tq = i - tr * (FE_PITCH / 4);
        at[u] = tr * FE_PITCH + tq * 4;
        if (i < FE_TASKS) packed[u] = fe_issue<CN, PATH>(pre[u], img, src_stride, width, height, tile, tr, tq);
    }
    static_assert(FE_PITCH % 16 == 0, "whole 16-byte pieces");
    for (int i = tid; i < FE_SROWS * FE_PITCH / 16; i += FE_THREADS) reinterpret_cast<uint4 *>(s_score)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 3) s_counters[tid] = 0;
    FE_STAMP(1);
#pragma unroll
    for (int u = 0; u < FE_TPT; u++)
        if (tid + u * FE_THREADS < FE_TASKS)
            *reinterpret_cast<uint32_t *>(&s_gray[at[u]]) = fe_gray4<CN>(pre[u], packed[u], gd);
}

// One block = one 128x32 tile of one frame:
//   0. the tile plus a 4-px apron is staged in LDS as 8-bit gray (BGR/BGRA are
//      converted on the way in; 16-B vector loads, all issued before first use);
//   1. cheap FAST pre-test on the tile and its 1-px rim in packed 16-bit
//      arithmetic; survivors (about 8 % of the pixels) are compacted into one
//      LDS queue (wave ballot + one LDS atomic per wave);
//   2. the queue is processed densely by the first waves: full FAST-9/16 score
//      into the LDS score tile; the other waves compute the Gaussian's row sums
//      meanwhile (4-byte dot products, two rows per task);
//   3. 3x3 strict non-max suppression (cv::FAST, nonmaxSuppression = true) of
//      the queued corners; survivors go straight to the tile's slot list in
//      global memory as keys ((255-score) << 24) | y << 12 | x (slots from an
//      LDS counter; the wave that finishes last stores the tile's count -- no
//      global atomics, no zero-initialised counters).  The other waves run the
//      Gaussian's column pass (2-element dot products over row pairs).
// tile (row r, column c) <-> image (y0 - 4 + r, x0 - 4 + c).
//
// Tile order is XCD-aware: consecutive blocks (grid x fastest) go to consecutive
// XCDs, so blocks b and b+8 share an XCD (and its L2).  By default XCD x = b % 8
// works along whole tile rows (frame * tiles_y + ty congruent to x mod 8), one
// tile after the other: a 128-px gray tile is one cache line wide, so its 4-px
// apron reads both neighbouring lines, and with the row on one XCD those are
// fetched into one L2 once.  VO_TILE_ORDER=cols selects the other order (whole
// tile columns per XCD: vertically adjacent tiles share 8 of their 40 staged
// rows), which fetches about twice as much and takes the same time.
// (col = frame * tiles_x + tile column, ty = tile row: which tile this block processes)
template <int CN>
__device__ __forceinline__ void fe_tile(const uint8_t *__restrict__ src, int width, int height, int src_stride,
                                        size_t src_frame_stride, PlaneBatch gray_out, PlaneBatch score,
                                        PlaneBatch blur, int thr, GrayCoeffs c, uint32_t *__restrict__ cand,
                                        uint32_t *__restrict__ tile_count, int batch, uint32_t inv_tiles_x,
                                        int margin, int col, int ty, int unused_edge = 0)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_gray[FE_GROWS * FE_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t s_score[FE_SROWS * FE_PITCH];
    __shared__ __attribute__((aligned(16))) uint32_t s_hpair[FE_HPAIRS * FE_TW];  // row sums of rows (2p, 2p+1)
    __shared__ uint16_t s_queue[FE_QCAP + 64];  // + one spare slot per lane for the unconditional stores
    __shared__ uint32_t s_counters[3];  // queue entries, emitted corners, waves done
    uint32_t &s_qcount = s_counters[0], &s_ecount = s_counters[1], &s_done = s_counters[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (width + FE_TW - 1) / FE_TW, tiles_y = (height + FE_TH - 1) / FE_TH;
    // rows 16-B (BGRA) / 4-B (gray, BGR) aligned: vector loads allowed
    const bool aligned = CN == 4 ? ((((uintptr_t)src | (uintptr_t)src_stride | src_frame_stride) & 15) == 0)
                                 : ((((uintptr_t)src | (uintptr_t)src_stride | src_frame_stride) & 3) == 0);

    if (col >= batch * tiles_x) return;
    FeTile tile;
    {
        tile.frame = (int)magic_div((uint32_t)col, inv_tiles_x);  // col / tiles_x
        tile.x0 = (col - tile.frame * tiles_x) * FE_TW;
        tile.y0 = ty * FE_TH;
        tile.inside = tile.x0 >= 4 && tile.x0 + FE_PITCH - 4 <= width && tile.y0 >= 4 &&
                      tile.y0 + FE_GROWS - 4 <= height;
        tile.fast = !tile.inside && aligned && width >= 8 && height >= 8;
        tile.whole = tile.x0 >= margin && tile.x0 + FE_TW <= width - margin && tile.y0 >= margin &&
                     tile.y0 + FE_TH <= height - margin;
    }
    const int x0 = tile.x0, y0 = tile.y0, f = tile.frame;

    // A tile that starts within `unused_edge` pixels of the right or bottom edge holds nothing a later stage reads
    // (no corners are looked for there: unused_edge <= margin; the caller reads the blurred plane only further
    // inside): it only reports an empty corner list.
    if (unused_edge > 0 && unused_edge <= margin && (x0 >= width - unused_edge || y0 >= height - unused_edge)) {
        if (cand && tid == 0) tile_count[(size_t)f * tiles_x * tiles_y + (y0 / FE_TH) * tiles_x + x0 / FE_TW] = 0;
        return;
    }

    FE_STAMP(0);
    // ---- 0. stage
    {
        const uint8_t *img = src + f * src_frame_stride;
        if (tile.inside && aligned)
            fe_stage<CN, FE_INSIDE>(s_gray, s_score, s_counters, img, src_stride, width, height, tile, c);
        else if (tile.fast)
            fe_stage<CN, FE_BORDER>(s_gray, s_score, s_counters, img, src_stride, width, height, tile, c);
        else
            fe_stage<CN, FE_GENERAL>(s_gray, s_score, s_counters, img, src_stride, width, height, tile, c);
    }
    FE_STAMP(2);
    __syncthreads();
    FE_STAMP(3);
    FE_CUT(1);

    const int gx = tid & 31, gy = tid >> 5;  // 32 quad columns x 16 rows: this thread's quads are at rows gy + 16 k
    const int x = x0 + gx * 4;               // image column of the first pixel of its quads
    const bool want_fast = score.base != nullptr || cand != nullptr;
    int qn = 0;  // entries in the tile queue

    if (want_fast) {
        // ---- 1. pre-test: interior (FE_QPT quads of four pixels per thread) ...
        bool pass[4 * FE_QPT];
        const us2 tt = as_us2((uint32_t)thr * 0x00010001u);
        // corners are looked for `margin` pixels inside the image (3: where cv::FAST evaluates; more when the
        // caller drops a wider border anyway)
        auto pretest = [&](int k, bool masked, uint32_t &r01, uint32_t &r23) {
            const int ry = gy + 16 * k;  // interior row of this quad
            const uint32_t up1 = *reinterpret_cast<const uint32_t *>(&s_gray[(ry + 1) * FE_PITCH + gx * 4 + 4]);
            const uint32_t dn1 = *reinterpret_cast<const uint32_t *>(&s_gray[(ry + 7) * FE_PITCH + gx * 4 + 4]);
            const uint32_t *rm = reinterpret_cast<const uint32_t *>(&s_gray[(ry + 4) * FE_PITCH + gx * 4]);
            const uint32_t d0 = rm[0], d1 = rm[1], d2 = rm[2];
            r01 = quick_pair<0>(up1, d0, d1, d2, dn1, tt);
            r23 = quick_pair<2>(up1, d0, d1, d2, dn1, tt);
            if (masked) {
                // bit i of vm: pixel x+i lies inside the margin
                const int y = y0 + ry;
                uint32_t vm = 0;
                if (y >= margin && y < height - margin) {
                    const int lo = max(margin - x, 0), hi = min(width - margin - x, 4);
                    if (hi > lo) vm = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                }
                r01 &= ((vm & 1u) ? 0xffffu : 0u) | ((vm & 2u) ? 0xffff0000u : 0u);
                r23 &= ((vm & 4u) ? 0xffffu : 0u) | ((vm & 8u) ? 0xffff0000u : 0u);
            }
        };
        uint32_t r01[FE_QPT], r23[FE_QPT];
        if (tile.whole) {  // block-uniform
#pragma unroll
            for (int k = 0; k < FE_QPT; k++) pretest(k, false, r01[k], r23[k]);
        } else {
            // a wave's two rows of quad k are y0 + 16 k + 2 * wave (+1): skipped when both lie outside the margin
            const int wrow = y0 + 2 * __builtin_amdgcn_readfirstlane(wave);
            const bool xin = x0 < width - margin && x0 + FE_TW > margin;  // some column of the tile is inside
#pragma unroll
            for (int k = 0; k < FE_QPT; k++) {
                const int yw = wrow + 16 * k;
                r01[k] = r23[k] = 0;
                if (xin && yw + 1 >= margin && yw < height - margin) pretest(k, true, r01[k], r23[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < FE_QPT; k++) {
            pass[4 * k + 0] = (r01[k] & 0xffffu) != 0; pass[4 * k + 1] = r01[k] > 0xffffu;
            pass[4 * k + 2] = (r23[k] & 0xffffu) != 0; pass[4 * k + 3] = r23[k] > 0xffffu;
        }
        // ... and the 1-px rim around the tile (one pixel per thread of the first waves)
        int hr = -1, hc = 0;
        if (tid < FE_HALO_PIX) {
            // top row, bottom row, left column, right column of the rim (selects, no branches)
            const bool top = tid < FE_TW + 2, horiz = tid < 2 * (FE_TW + 2), left = tid < 2 * (FE_TW + 2) + FE_TH;
            const int side = tid - 2 * (FE_TW + 2) - (left ? 0 : FE_TH);
            hr = horiz ? (top ? 3 : 4 + FE_TH) : 4 + side;
            hc = horiz ? 3 + tid - (top ? 0 : FE_TW + 2) : (left ? 3 : 4 + FE_TW);
            const int ix = x0 - 4 + hc, iy = y0 - 4 + hr;
            bool ok = ix >= margin && ix < width - margin && iy >= margin && iy < height - margin;
            if (ok) {
                const uint8_t *p = &s_gray[hr * FE_PITCH + hc];
                ok = fast_quick(p[0], p[3 * FE_PITCH], p[-3 * FE_PITCH], p[3], p[-3], thr);
            }
            if (!ok) hr = -1;
        }
        // queue the survivors: one LDS atomic per wave
        {
            uint64_t m[4 * FE_QPT + 1];
#pragma unroll
            for (int i = 0; i < 4 * FE_QPT; i++) m[i] = __ballot(pass[i]);
            m[4 * FE_QPT] = __ballot(hr >= 0);
            uint32_t total = 0;
#pragma unroll
            for (int i = 0; i <= 4 * FE_QPT; i++) total += (uint32_t)__popcll(m[i]);
            if (total) {
                uint32_t base = 0;
                const uint32_t q0 = (uint32_t)(((gy + 4) << 8) + 4 + 4 * gx);
                if (lane == 0) base = atomicAdd(&s_qcount, total);
                base = __builtin_amdgcn_readfirstlane(base);
                // byte offsets into the queue: 2 * (base + passing lanes below this one); lanes that did not pass
                // write their own spare slot past the queue (unconditional stores)
                uint32_t base2 = 2u * base;
                const uint32_t spare2 = 2u * (uint32_t)(FE_QCAP + lane);
                uint8_t *qb = reinterpret_cast<uint8_t *>(s_queue);
#pragma unroll
                for (int i = 0; i < 4 * FE_QPT; i++) {
                    const uint32_t below =
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(m[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[i], 0u));
                    const uint32_t q = q0 + (uint32_t)((16 * (i >> 2)) << 8 | (i & 3));  // row gy + 16 k + 4, column 4 + 4 gx + i
                    *reinterpret_cast<uint16_t *>(qb + (pass[i] ? (below << 1) + base2 : spare2)) = (uint16_t)q;
                    base2 += 2u * (uint32_t)__popcll(m[i]);
                }
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[4 * FE_QPT] >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)m[4 * FE_QPT], 0u));
                *reinterpret_cast<uint16_t *>(qb + (hr >= 0 ? (below << 1) + base2 : spare2)) = (uint16_t)((hr << 8) | hc);
            }
        }
        FE_STAMP(4);
        __syncthreads();
        FE_STAMP(5);
        FE_CUT(2);

        // ---- 2. full score of the queued pixels (the first waves)
        qn = (int)s_qcount;
        for (int e = tid; e < qn; e += FE_THREADS) {
            const int q = s_queue[e];
            const int r = q >> 8, cc = q & 255;
            const uint8_t *p = &s_gray[r * FE_PITCH + cc];
            uint32_t ring[16];
            ring[0] = p[3 * FE_PITCH];       ring[1] = p[3 * FE_PITCH + 1];   ring[2] = p[2 * FE_PITCH + 2];
            ring[3] = p[FE_PITCH + 3];       ring[4] = p[3];                  ring[5] = p[-FE_PITCH + 3];
            ring[6] = p[-2 * FE_PITCH + 2];  ring[7] = p[-3 * FE_PITCH + 1];  ring[8] = p[-3 * FE_PITCH];
            ring[9] = p[-3 * FE_PITCH - 1];  ring[10] = p[-2 * FE_PITCH - 2]; ring[11] = p[-FE_PITCH - 3];
            ring[12] = p[-3];                ring[13] = p[FE_PITCH - 3];      ring[14] = p[2 * FE_PITCH - 2];
            ring[15] = p[3 * FE_PITCH - 1];
            s_score[(r - 3) * FE_PITCH + cc] = (uint8_t)fast_score(ring, p[0], thr);
        }
    }

    // ---- Gaussian, row pass: row sums of tile rows 1 .. FE_HROWS, two rows per task, the last threads of
    // the block first so that the waves with fewer queue entries take the second round
    if (blur.base) {
#pragma unroll
        for (int it = 0; it < (FE_HPAIRS * 32 + FE_THREADS - 1) / FE_THREADS; it++) {
            const int t = FE_THREADS - 1 - tid + it * FE_THREADS;
            if (t < FE_HPAIRS * 32) {
                const int hp = t >> 5, tg = t & 31;
                const uint32_t *ra = reinterpret_cast<const uint32_t *>(&s_gray[(2 * hp + 1) * FE_PITCH + tg * 4]);
                const uint32_t *rb = reinterpret_cast<const uint32_t *>(&s_gray[(2 * hp + 2) * FE_PITCH + tg * 4]);
                uint32_t ha[4], hb[4];
                hsum7_quad(ra[0], ra[1], ra[2], ha);
                hsum7_quad(rb[0], rb[1], rb[2], hb);
                uint4 o;
                o.x = ha[0] | (hb[0] << 16); o.y = ha[1] | (hb[1] << 16);
                o.z = ha[2] | (hb[2] << 16); o.w = ha[3] | (hb[3] << 16);
                *reinterpret_cast<uint4 *>(&s_hpair[hp * FE_TW + tg * 4]) = o;
            }
        }
    }
    FE_STAMP(6);
    __syncthreads();
    FE_STAMP(7);
    FE_CUT(3);

    // ---- 3. non-max suppression of the queued corners (score row sr <-> tile row sr + 3), direct emission
    if (cand) {
        const int tile_id = (y0 / FE_TH) * tiles_x + x0 / FE_TW;
        uint32_t *out = cand + ((size_t)f * tiles_x * tiles_y + tile_id) * kTileCap;
        for (int e0 = wave * 64; e0 < qn; e0 += FE_THREADS) {
            const int e = e0 + lane;
            bool is_max = false;
            uint32_t key = 0;
            if (e < qn) {
                const int q = s_queue[e];
                const int r = q >> 8, cc = q & 255;
                // rim pixels only feed their neighbours
                if (r >= 4 && r < 4 + FE_TH && cc >= 4 && cc < 4 + FE_TW) {
                    const uint8_t *sp = &s_score[(r - 3) * FE_PITCH + cc];
                    const uint32_t sc = sp[0];
                    // all eight neighbours at once: no short-circuit branches
                    const uint32_t n0 = umax3(sp[-FE_PITCH - 1], sp[-FE_PITCH], sp[-FE_PITCH + 1]);
                    const uint32_t n1 = umax3(sp[FE_PITCH - 1], sp[FE_PITCH], sp[FE_PITCH + 1]);
                    is_max = sc > umax3(n0, n1, max((uint32_t)sp[-1], (uint32_t)sp[1]));  // implies sc != 0
                    key = corner_key(sc, x0 - 4 + cc, y0 - 4 + r);
                }
            }
            const uint64_t mm = __ballot(is_max);
            if (mm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_ecount, (uint32_t)__popcll(mm));
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t pos =
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, base));
                if (is_max) out[pos] = key;
            }
        }
        // the wave that arrives last publishes the tile's corner count
        if (lane == 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's slot reservations are done
            if (atomicAdd(&s_done, 1u) == FE_WAVES - 1)
                tile_count[(size_t)f * tiles_x * tiles_y + tile_id] = s_ecount;
        }
    }
    FE_STAMP(8);

    // ---- Gaussian, column pass: output rows (2m, 2m+1) from the row pairs m .. m+3
    static_assert((FE_TH / 2) * 32 % FE_THREADS == 0, "whole rounds of one pair of output rows per thread");
    if (blur.base)
#pragma unroll
    for (int cr = 0; cr < (FE_TH / 2) * 32 / FE_THREADS; cr++) {
        const int ct = FE_THREADS - 1 - tid + cr * FE_THREADS;
        const int m = ct >> 5, tg = ct & 31;
        uint4 P[4];
#pragma unroll
        for (int k = 0; k < 4; k++) P[k] = *reinterpret_cast<const uint4 *>(&s_hpair[(m + k) * FE_TW + tg * 4]);
        // even row: taps (18,34)(49,55)(49,34)(18,-); odd row: (-,18)(34,49)(55,49)(34,18)
        constexpr uint32_t E0 = 18u | (34u << 16), E1 = 49u | (55u << 16), E2 = 49u | (34u << 16), E3 = 18u;
        constexpr uint32_t O0 = 18u << 16, O1 = 34u | (49u << 16), O2 = 55u | (49u << 16), O3 = 34u | (18u << 16);
        const uint32_t col[4][4] = {{P[0].x, P[1].x, P[2].x, P[3].x}, {P[0].y, P[1].y, P[2].y, P[3].y},
                                    {P[0].z, P[1].z, P[2].z, P[3].z}, {P[0].w, P[1].w, P[2].w, P[3].w}};
        // The accumulators start at 0xff000000 + the rounding constant and the dot products saturate at 2^32 - 1:
        // byte 2 of the result is then min((sum + 2^15) >> 16, 255) without a shift or a compare.
        uint32_t ae[4], ao[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ae[i] = ao[i] = 0xff008000u;
            ae[i] = __builtin_amdgcn_udot2(as_us2(col[i][0]), as_us2(E0), ae[i], true);
            ae[i] = __builtin_amdgcn_udot2(as_us2(col[i][1]), as_us2(E1), ae[i], true);
            ae[i] = __builtin_amdgcn_udot2(as_us2(col[i][2]), as_us2(E2), ae[i], true);
            ae[i] = __builtin_amdgcn_udot2(as_us2(col[i][3]), as_us2(E3), ae[i], true);
            ao[i] = __builtin_amdgcn_udot2(as_us2(col[i][0]), as_us2(O0), ao[i], true);
            ao[i] = __builtin_amdgcn_udot2(as_us2(col[i][1]), as_us2(O1), ao[i], true);
            ao[i] = __builtin_amdgcn_udot2(as_us2(col[i][2]), as_us2(O2), ao[i], true);
            ao[i] = __builtin_amdgcn_udot2(as_us2(col[i][3]), as_us2(O3), ao[i], true);
        }
        // bytes 2 of four sums -> one dword
        const uint32_t oe = pack_byte2(ae[0], ae[1], ae[2], ae[3]), oo = pack_byte2(ao[0], ao[1], ao[2], ao[3]);
        const int yy = y0 + 2 * m, xx = x0 + tg * 4;
        uint8_t *bp = blur.base + f * blur.plane_stride + (size_t)yy * blur.pitch + xx;
        if (yy < height) *reinterpret_cast<uint32_t *>(bp) = oe;
        if (yy + 1 < height) *reinterpret_cast<uint32_t *>(bp + blur.pitch) = oo;
    }

#pragma unroll
    for (int k = 0; k < FE_QPT; k++) {
        const int ry = gy + 16 * k, y = y0 + ry;
        if (gray_out.base && y < height)
            *reinterpret_cast<uint32_t *>(gray_out.base + f * gray_out.plane_stride + (size_t)y * gray_out.pitch + x) =
                *reinterpret_cast<const uint32_t *>(&s_gray[(ry + 4) * FE_PITCH + gx * 4 + 4]);
        if (score.base && y < height)  // diagnostic score map
            *reinterpret_cast<uint32_t *>(score.base + f * score.plane_stride + (size_t)y * score.pitch + x) =
                *reinterpret_cast<const uint32_t *>(&s_score[(ry + 1) * FE_PITCH + gx * 4 + 4]);
    }
    FE_STAMP(9);
}

// grid (column order, by_rows == 0): x = 8 * tile row + XCD slot, y = group of eight tile columns
// (frame * tiles_x + tx); row order: see the body
template <int CN>
__global__ __launch_bounds__(FE_THREADS) void k_frontend(const uint8_t *__restrict__ src, int width,
                                                         int height, int src_stride,
                                                         size_t src_frame_stride, PlaneBatch gray_out,
                                                         PlaneBatch score, PlaneBatch blur, int thr,
                                                         GrayCoeffs c, uint32_t *__restrict__ cand,
                                                         uint32_t *__restrict__ tile_count, int batch,
                                                         uint32_t inv_tiles_x, int margin, uint32_t inv_tiles_y,
                                                         int by_rows, unsigned long long *stamp)
{
    // (optional: the device wall clock at the start of the launch, for callers that time the detector)
    if (stamp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *stamp = wall_clock64();
    if (!by_rows) {
        fe_tile<CN>(src, width, height, src_stride, src_frame_stride, gray_out, score, blur, thr, c, cand, tile_count,
                    batch, inv_tiles_x, margin, (int)(blockIdx.y * 8 + (blockIdx.x & 7)), (int)(blockIdx.x >> 3));
        return;
    }
    // row-major variant: grid x = 8 * tile column + XCD slot, y = group of eight tile rows (frame * tiles_y + ty);
    // the tiles of one tile row run on one XCD, one after the other
    const int tiles_x = (width + FE_TW - 1) / FE_TW, tiles_y = (height + FE_TH - 1) / FE_TH;
    const int row = (int)(blockIdx.y * 8 + (blockIdx.x & 7));
    if (row >= batch * tiles_y) return;
    const int frame = (int)magic_div((uint32_t)row, inv_tiles_y);
    fe_tile<CN>(src, width, height, src_stride, src_frame_stride, gray_out, score, blur, thr, c, cand, tile_count,
                batch, inv_tiles_x, margin, frame * tiles_x + (int)(blockIdx.x >> 3), row - frame * tiles_y);
}

// The tile kernel over several gray planes of different sizes in ONE launch (the upper levels of the ORB
// pyramid): a 1-D grid, eight consecutive blocks (one per XCD) form a group, groups [first_group,
// first_group + tiles_y * column groups) belong to a level, ordered column group major / tile row minor like the
// 2-D grid above.  The levels are listed largest first, so the small ones fill the tail of the large ones instead
// of each paying its own ramp-down.
struct FeLevels {
    int n;
    struct L {
        const uint8_t *src;
        uint8_t *blur;
        uint32_t *cand, *tile_count;
        int width, height, pitch, tiles_y;
        uint32_t inv_tiles_x, inv_tiles_y;
        int first_group;
    } lv[kOrbMaxLevels];
    int total_groups;
    int by_rows;          // groups enumerate (tile-row group, tile column) instead of (tile-column group, tile row)
    int unused_edge;      // tiles that start this close to the right / bottom edge are skipped (see fe_tile)
    size_t frame_stride;  // bytes between the planes of consecutive frames (same for input and output)
};

__global__ __launch_bounds__(FE_THREADS) void k_frontend_levels(FeLevels t, int thr, GrayCoeffs c, int batch, int margin)
{
    const int g = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kOrbMaxLevels; i++)
        if (i < t.n && g >= t.lv[i].first_group) l = i;
    const FeLevels::L &L = t.lv[l];
    const int local = g - L.first_group;
    if (t.by_rows) {
        const int tiles_x = (L.width + FE_TW - 1) / FE_TW;
        const int rg = (int)magic_div((uint32_t)local, L.inv_tiles_x);  // local / tiles_x
        const int tx = local - rg * tiles_x, row = rg * 8 + xcd;
        if (row >= batch * L.tiles_y) return;
        const int frame = (int)magic_div((uint32_t)row, L.inv_tiles_y);
        fe_tile<1>(L.src, L.width, L.height, L.pitch, t.frame_stride, PlaneBatch{nullptr, 0, 0},
                   PlaneBatch{nullptr, 0, 0}, PlaneBatch{L.blur, L.pitch, t.frame_stride}, thr, c, L.cand, L.tile_count,
                   batch, L.inv_tiles_x, margin, frame * tiles_x + tx, row - frame * L.tiles_y, t.unused_edge);
        return;
    }
    const int cg = (int)magic_div((uint32_t)local, L.inv_tiles_y);  // local / tiles_y
    const int ty = local - cg * L.tiles_y;
    fe_tile<1>(L.src, L.width, L.height, L.pitch, t.frame_stride, PlaneBatch{nullptr, 0, 0}, PlaneBatch{nullptr, 0, 0},
               PlaneBatch{L.blur, L.pitch, t.frame_stride}, thr, c, L.cand, L.tile_count, batch, L.inv_tiles_x, margin,
               cg * 8 + xcd, ty, t.unused_edge);
}

#ifdef VO_FE_STAMPS
extern "C" int vo_debug_fe_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fe_stamps), sizeof(unsigned long long) * FE_STAMP_BLOCKS * FE_STAMP_N);
}
#endif

// VO_TILE_ORDER=cols restores the column-major tile order (for A/B measurements)
static bool tile_order_by_rows()
{
    static const bool v = [] { const char *e = getenv("VO_TILE_ORDER"); return !(e && e[0] == 'c'); }();
    return v;
}

void launch_frontend(const uint8_t *src, int width, int height, int src_stride, size_t src_frame_stride,
                     int channels, PlaneBatch gray_out, PlaneBatch score, PlaneBatch blur, int threshold,
                     uint32_t *cand, uint32_t *tile_count, int batch, GrayCoeffs c, hipStream_t s, int margin,
                     unsigned long long *stamp)
{
    // block b serves XCD group b % 8, which owns the tile columns congruent to
    // b % 8; the grid holds the largest group's tile count for every group
    const int tiles_x = (width + FE_TW - 1) / FE_TW, ncols = tiles_x * batch, tiles_y = (height + FE_TH - 1) / FE_TH;
    const bool by_rows = tile_order_by_rows();
    dim3 grid(8 * tiles_y, (ncols + 7) / 8);
    if (by_rows) grid = dim3(8 * tiles_x, (tiles_y * batch + 7) / 8);
    const uint32_t inv_tiles_x = magic_inv(tiles_x), inv_tiles_y = magic_inv(tiles_y);
    threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    margin = margin < 3 ? 3 : margin;  // cv::FAST itself never evaluates the outer 3 pixels
    if (channels == 1)
        hipLaunchKernelGGL(k_frontend<1>, grid, dim3(FE_THREADS), 0, s, src, width, height, src_stride,
                           src_frame_stride, gray_out, score, blur, threshold, c, cand, tile_count, batch, inv_tiles_x, margin,
                           inv_tiles_y, by_rows ? 1 : 0, stamp);
    else if (channels == 3)
        hipLaunchKernelGGL(k_frontend<3>, grid, dim3(FE_THREADS), 0, s, src, width, height, src_stride,
                           src_frame_stride, gray_out, score, blur, threshold, c, cand, tile_count, batch, inv_tiles_x, margin,
                           inv_tiles_y, by_rows ? 1 : 0, stamp);
    else
        hipLaunchKernelGGL(k_frontend<4>, grid, dim3(FE_THREADS), 0, s, src, width, height, src_stride,
                           src_frame_stride, gray_out, score, blur, threshold, c, cand, tile_count, batch, inv_tiles_x, margin,
                           inv_tiles_y, by_rows ? 1 : 0, stamp);
}

void launch_frontend_levels(const OrbPlan &plan, int first_level, const uint8_t *pyr, uint8_t *pyr_blur,
                            uint32_t *cand, uint32_t *tile_count, int threshold, int batch, GrayCoeffs c,
                            hipStream_t s, int margin)
{
    FeLevels t{};
    t.frame_stride = plan.frame_bytes;
    t.by_rows = tile_order_by_rows() ? 1 : 0;
    // the steered descriptor samples at most kBriefReach pixels from a key point, and key points lie kFBorder
    // pixels inside the level: the outermost kFBorder - kBriefReach columns and rows of the blurred level are
    // never read
    t.unused_edge = kFBorder - kBriefReach;
    int groups = 0;
    for (int l = first_level; l < plan.nlevels; l++) {
        const OrbLevel &P = plan.lv[l];
        FeLevels::L &L = t.lv[t.n++];
        const int tiles_x = (P.w + FE_TW - 1) / FE_TW, tiles_y = (P.h + FE_TH - 1) / FE_TH;
        L.src = pyr + P.off;
        L.blur = pyr_blur + P.off;
        L.cand = cand + P.cand_off;
        L.tile_count = tile_count + P.count_off;
        L.width = P.w; L.height = P.h; L.pitch = P.pitch; L.tiles_y = tiles_y;
        L.inv_tiles_x = magic_inv(tiles_x);
        L.inv_tiles_y = magic_inv(tiles_y);
        L.first_group = groups;
        groups += t.by_rows ? tiles_x * ((tiles_y * batch + 7) / 8) : tiles_y * ((tiles_x * batch + 7) / 8);
    }
    if (t.n == 0) return;
    t.total_groups = groups;
    threshold = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    margin = margin < 3 ? 3 : margin;
    hipLaunchKernelGGL(k_frontend_levels, dim3((unsigned)groups * 8u), dim3(FE_THREADS), 0, s, t, threshold, c, batch,
                       margin);
}

// ---------------------------------------------------------------- select

constexpr int SEL_THREADS = 1024;

// Ascending bitonic sort of n_pow2 keys in LDS, any size: one compare-exchange per thread and step, a barrier
// per stage.
__device__ __forceinline__ void bitonic_sort_lds_any(uint32_t *keys, int n_pow2)
{
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < n_pow2 / 2; t += SEL_THREADS) {
                // t-th compare-exchange of this stage
                const int i = ((t / j) * 2 * j) + (t % j);
                const int p = i + j;
                const bool up = ((i & k) == 0);
                const uint32_t a = keys[i], b = keys[p];
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
    }
}

// Ascending bitonic sort of n_pow2 <= 2 * SEL_THREADS keys in LDS.  A thread keeps the keys at positions tid and
// tid + SEL_THREADS in registers; exchanges at a distance below the wave size are lane shuffles, the distance
// SEL_THREADS is inside the thread, and only the distances in between go through LDS with block barriers (10 of
// the 55 stages of 1024 keys; a single frame's selection is a third sort, so the barriers were most of it).
__device__ __forceinline__ void bitonic_sort_lds(uint32_t *keys, int n_pow2)
{
    const int i0 = threadIdx.x, i1 = threadIdx.x + SEL_THREADS;
    const bool two = n_pow2 > SEL_THREADS;  // block-uniform
    uint32_t v0 = i0 < n_pow2 ? keys[i0] : 0xffffffffu;
    uint32_t v1 = two ? keys[i1] : 0xffffffffu;
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == SEL_THREADS) {  // only with two keys per thread, k = 2 * SEL_THREADS: ascending everywhere
                const uint32_t lo = min(v0, v1), hi = max(v0, v1);
                v0 = lo; v1 = hi;
                continue;
            }
            uint32_t p0, p1 = 0;
            if (j >= 64) {
                __syncthreads();  // the previous stage's reads are done
                if (i0 < n_pow2) keys[i0] = v0;
                if (two) keys[i1] = v1;
                __syncthreads();
                p0 = i0 < n_pow2 ? keys[i0 ^ j] : v0;
                if (two) p1 = keys[i1 ^ j];
            } else {
                p0 = __shfl_xor(v0, j);
                if (two) p1 = __shfl_xor(v1, j);
            }
            const bool low = (i0 & j) == 0;  // (i1 & j) is the same: j < SEL_THREADS
            v0 = (((i0 & k) == 0) == low) ? min(v0, p0) : max(v0, p0);
            if (two) v1 = (((i1 & k) == 0) == low) ? min(v1, p1) : max(v1, p1);
        }
    }
    __syncthreads();
    if (i0 < n_pow2) keys[i0] = v0;
    if (two) keys[i1] = v1;
    __syncthreads();
}

// One block per frame.  reference src/feature_detector.cpp:62-68 sorts all
// corners by response and keeps max_features; here the same set (with the
// canonical tie order: response descending, raster index ascending) is found
// with a two-level histogram select (score, then image row inside the tied
// score class) so that at most max_features + width/2 keys are sorted in LDS.
// Then KeyPointsFilter::runByImageBorder(31) as done inside ORB::compute
// (reference src/feature_detector.cpp:73), order preserved.
__global__ __launch_bounds__(SEL_THREADS) void k_select(const uint32_t *__restrict__ cand,
                                                        const uint32_t *__restrict__ tile_count,
                                                        uint32_t *__restrict__ flat, int ntiles,
                                                        int width, int height, int max_features, int border,
                                                        vo_kp *__restrict__ kps, int kp_stride,
                                                        int *__restrict__ nkp)
{
    __shared__ uint32_t s_keys[kSelectCap];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_rows[kMaxDim];
    __shared__ uint32_t s_scan[SEL_THREADS];
    __shared__ int s_cut_bin, s_need, s_cut_row, s_count;

    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    // the frame's corners as one list (s_keys and s_scan are free until the gather below)
    uint32_t *c = flat + (size_t)f * ntiles * kTileCap;
    const int M = flatten_tile_lists<SEL_THREADS>(cand + (size_t)f * ntiles * kTileCap,
                                                  tile_count + (size_t)f * ntiles, ntiles, c, s_keys, s_scan);
    const bool all = M <= max_features;

    if (tid == 0) { s_count = 0; s_cut_bin = 256; s_need = 0; s_cut_row = kMaxDim; }
    if (!all) {
        for (int i = tid; i < 256; i += SEL_THREADS) s_hist[i] = 0;
        for (int i = tid; i < kMaxDim; i += SEL_THREADS) s_rows[i] = 0;
        __syncthreads();
        for (int i = tid; i < M; i += SEL_THREADS) atomicAdd(&s_hist[c[i] >> 24], 1u);
        __syncthreads();
        {
            // the score class at which the running count reaches max_features
            const uint32_t hv = tid < 256 ? s_hist[tid] : 0u;
            const uint32_t inc = block_scan_inclusive<SEL_THREADS>(hv, s_scan);
            const uint32_t target = (uint32_t)max_features;
            if (tid < 256 && inc >= target && inc - hv < target) {
                s_cut_bin = tid;
                s_need = (int)(target - (inc - hv));  // >= 1 keys to take from the tied class
            }
        }
        __syncthreads();
        const uint32_t cb = (uint32_t)s_cut_bin;
        for (int i = tid; i < M; i += SEL_THREADS) {
            const uint32_t k = c[i];
            if ((k >> 24) == cb) atomicAdd(&s_rows[key_y(k)], 1u);
        }
        __syncthreads();
        {
            // the image row at which the tied class has supplied the keys still needed (4 rows per thread)
            static_assert(kMaxDim == 4 * SEL_THREADS, "four row bins per thread");
            uint32_t rv[4], sum4 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { rv[j] = s_rows[4 * tid + j]; sum4 += rv[j]; }
            const uint32_t inc = block_scan_inclusive<SEL_THREADS>(sum4, s_scan);
            const uint32_t need = (uint32_t)s_need;
            uint32_t cum = inc - sum4;
            if (cum < need && inc >= need) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool before = cum < need;
                    cum += rv[j];
                    if (before && cum >= need) s_cut_row = 4 * tid + j;
                }
            }
        }
    }
    __syncthreads();

    // gather the keys to sort
    {
        const uint32_t cb = (uint32_t)s_cut_bin, cr = (uint32_t)s_cut_row;
        for (int i = tid; i < M; i += SEL_THREADS) {
            const uint32_t k = c[i];
            bool take;
            uint32_t key;
            if (all) {
                take = true;
                key = ((k & 0xffffffu) << 8) | (k >> 24);  // raster order
            } else {
                const uint32_t b = k >> 24;
                take = b < cb || (b == cb && (uint32_t)key_y(k) <= cr);
                key = k;
            }
            if (take) {
                const int slot = atomicAdd(&s_count, 1);
                if (slot < kSelectCap) s_keys[slot] = key;
            }
        }
    }
    __syncthreads();
    const int K = min(s_count, kSelectCap);
    int n_pow2 = 1;
    while (n_pow2 < K) n_pow2 <<= 1;
    for (int i = K + tid; i < n_pow2; i += SEL_THREADS) s_keys[i] = 0xffffffffu;
    __syncthreads();
    if (n_pow2 > 2 * SEL_THREADS)
        bitonic_sort_lds_any(s_keys, n_pow2);
    else if (n_pow2 > 1)
        bitonic_sort_lds(s_keys, n_pow2);

    // border filter + ordered compaction
    const int n_sel = min(K, max_features);
    const int per = (n_sel + SEL_THREADS - 1) / SEL_THREADS;  // consecutive keys per thread
    const int begin = tid * per, end = min(begin + per, n_sel);
    const bool size_ok = width > 2 * border && height > 2 * border;
    int cnt = 0;
    for (int i = begin; i < end; i++) {
        const uint32_t key = s_keys[i];
        const uint32_t pos = all ? key >> 8 : key;
        const int yy = key_y(pos), xx = key_x(pos);
        cnt += (size_ok && xx >= border && xx < width - border && yy >= border && yy < height - border);
    }
    const uint32_t inc_cnt = block_scan_inclusive<SEL_THREADS>((uint32_t)cnt, s_scan);
    int out = (int)inc_cnt - cnt;
    vo_kp *o = kps + (size_t)f * kp_stride;
    for (int i = begin; i < end; i++) {
        const uint32_t key = s_keys[i];
        const uint32_t pos = all ? key >> 8 : key;
        const uint32_t sc = 255u - (all ? key & 255u : key >> 24);
        const int yy = key_y(pos), xx = key_x(pos);
        if (size_ok && xx >= border && xx < width - border && yy >= border && yy < height - border) {
            vo_kp kp;
            kp.x = (float)xx; kp.y = (float)yy; kp.size = 7.f; kp.angle = -1.f;
            kp.response = (float)sc; kp.octave = 0;
            o[out++] = kp;
        }
    }
    if (tid == SEL_THREADS - 1) nkp[f] = (int)inc_cnt;
}

void launch_select(const uint32_t *cand, const uint32_t *tile_count, uint32_t *flat, int width, int height,
                   int max_features, int border, vo_kp *kps, int kp_stride, int *nkp, int batch,
                   hipStream_t s)
{
    hipLaunchKernelGGL(k_select, dim3(batch), dim3(SEL_THREADS), 0, s, cand, tile_count, flat,
                       tiles_of(width, height), width, height, max_features, border, kps, kp_stride, nkp);
}

// ----------------------------------------------------------------- BRIEF

// 32 lanes per keypoint, one descriptor byte (8 tests) per lane.
// cv::ORB::compute's computeOrbDescriptors with WTA_K = 2
// (reference src/feature_detector.cpp:73).  The pattern holds the 512 sample
// offsets already rotated by the (constant) keypoint angle and rounded; they
// stay within +-13 px, so the 27 x 27 patch of each keypoint is first staged
// in LDS with dword loads (a keypoint's 32 lanes fetch four rows per
// instruction) and the 512 samples are LDS byte reads instead of 512 scattered
// global ones.
constexpr int BR_R = 13;
constexpr int BR_PWD = (2 * BR_R + 1 + 3 + 3) / 4;  // dwords per staged row (any alignment of the first column)
constexpr int BR_KPG = 2;           // key points a 32-lane group handles one after the other
constexpr int BR_KPB = 8 * BR_KPG;  // key points per block
// (The loads of both patches of a group are issued up front; the second patch waits in registers while the first
// descriptor is computed from LDS.  A group is half a wave and only touches its own patch, so program order is all
// that is needed between its LDS writes and reads; the block barrier is for the pattern.)
__global__ __launch_bounds__(256) void k_brief(PlaneBatch blur, const vo_kp *__restrict__ kps, int kp_stride,
                                               const int *__restrict__ nkp,
                                               const int8_t *__restrict__ pattern,
                                               uint8_t *__restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) int8_t s_pat[1024];
    __shared__ __attribute__((aligned(16))) uint32_t s_patch[8][(2 * BR_R + 1) * BR_PWD];
    reinterpret_cast<uint32_t *>(s_pat)[threadIdx.x] = reinterpret_cast<const uint32_t *>(pattern)[threadIdx.x];
    const int f = blockIdx.y;
    const int g = threadIdx.x >> 5, k0 = blockIdx.x * BR_KPB + g * BR_KPG;  // the group's first key point
    const int byte = threadIdx.x & 31;
    const int n_kp = nkp[f];
    // all of a lane's loads are issued before the first LDS write (lanes past the end re-read the last dword)
    constexpr int N = (2 * BR_R + 1) * BR_PWD, PER = (N + 31) / 32;
    uint32_t v[BR_KPG][PER];
    int col0[BR_KPG];
#pragma unroll
    for (int i = 0; i < BR_KPG; i++) {
        col0[i] = 0;
        if (k0 + i < n_kp) {
            const vo_kp kp = kps[(size_t)f * kp_stride + k0 + i];
            const int cx = __float2int_rn(kp.x), cy = __float2int_rn(kp.y);
            const int x_al = (cx - BR_R) & ~3;  // keypoints are >= 31 px inside the image: never negative
            col0[i] = cx - x_al;
            const uint8_t *src = blur.base + f * blur.plane_stride + (size_t)(cy - BR_R) * blur.pitch + x_al;
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int j = min(byte + 32 * u, N - 1);
                const int r = j / BR_PWD, c = j - r * BR_PWD;
                v[i][u] = *reinterpret_cast<const uint32_t *>(src + (size_t)r * blur.pitch + 4 * c);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BR_KPG; i++) {
        const int k = k0 + i;
        if (k >= n_kp) break;  // uniform over the group (half a wave)
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (byte + 32 * u < N) s_patch[g][byte + 32 * u] = v[i][u];
        const uint8_t *center = reinterpret_cast<const uint8_t *>(s_patch[g]) + BR_R * (BR_PWD * 4) + col0[i];
        const int8_t *p = s_pat + byte * 32;
        uint32_t val = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int t0 = center[p[4 * b + 1] * (BR_PWD * 4) + p[4 * b + 0]];
            const int t1 = center[p[4 * b + 3] * (BR_PWD * 4) + p[4 * b + 2]];
            val |= (uint32_t)(t0 < t1) << b;
        }
        desc[((size_t)f * kp_stride + k) * 32 + byte] = (uint8_t)val;
    }
}

void launch_brief(PlaneBatch blur, const vo_kp *kps, int kp_stride, const int *nkp, const int8_t *pattern,
                  uint8_t *desc, int batch, hipStream_t s)
{
    dim3 grid((kp_stride + BR_KPB - 1) / BR_KPB, batch);
    hipLaunchKernelGGL(k_brief, grid, dim3(256), 0, s, blur, kps, kp_stride, nkp, pattern, desc);
}

}  // namespace vo
