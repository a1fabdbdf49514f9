// Full ORB detector kernels for gfx950 (pyramid, per-level Harris ranking,
// intensity-centroid orientation, steered rBRIEF): what
// cv::ORB::create(nfeatures, scaleFactor, nlevels)->detectAndCompute() computes
// for the object the reference builds at src/feature_detector.cpp:18-22.
// The per-level FAST + NMS + Gaussian stage is the fused tile kernel of
// k_image.hip, run once per level.
//
// Integer stages are bit-exact by construction; the Harris response, the
// orientation and the pattern rotation are float32 expressions evaluated in
// OpenCV's order (the library is built with -ffp-contract=off).
#include <float.h>

#include <vector>

#include "vo_internal.h"

namespace vo {

namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// Frame-per-XCD block order for kernels whose blocks of one frame read overlapping lines.  The grid is 1-D with
// per_frame * batch blocks and consecutive blocks go to consecutive XCDs.  When the batch is a multiple of eight,
// XCD x = block % 8 works through frames x, x + 8, ... one after the other, a frame's blocks in order, so that the
// frame's lines are fetched into one L2 only; other batches (a single frame above all) keep the plain order, which
// spreads every frame over all XCDs.
struct FrameBlock {
    int frame, block;
};
struct FrameGrid {
    int per_frame, batch;
    uint32_t inv;  // magic_inv(per_frame)
};
inline FrameGrid frame_grid(int per_frame, int batch)
{
    return FrameGrid{per_frame, batch, magic_inv(per_frame)};
}
__device__ __forceinline__ FrameBlock frame_block(const FrameGrid &g)
{
    if (g.batch & 7) {
        const int frame = (int)magic_div(blockIdx.x, g.inv);
        return FrameBlock{frame, (int)blockIdx.x - frame * g.per_frame};
    }
    const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    const int group = (int)magic_div((uint32_t)slot, g.inv);
    return FrameBlock{group * 8 + xcd, slot - group * g.per_frame};
}

// ------------------------------------------------------------------ resize

constexpr int RZ_TW = 128, RZ_TH = 32, RZ_THREADS = 256;
constexpr int RZ_FP = 288;    // LDS footprint row pitch (bytes): scale <= 2 plus alignment slack
constexpr int RZ_FROWS = 68;  // footprint rows
constexpr int RZ_SPAN = 6;    // largest source distance between the first and last of a thread's four columns
static_assert(RZ_TH == kResizeRowPad, "row tables are padded to whole resize tiles");

// Level l from level l-1 for every frame of the batch: cv::resize(...,
// INTER_LINEAR_EXACT) -- rows in 8.8 fixed point, columns in 16.16,
// (v + 2^15) >> 16.  xtab/ytab hold per destination column/row the source
// offset (low 16 bits) and the 8.8 weight of the second tap (high 16 bits),
// built on the host by orb_axis_table().  One 128x32 output tile per block;
// the source footprint is staged in LDS with 16-byte loads.  A thread computes
// the same four columns of four rows: per source row three LDS dwords are
// funnel-shifted once so that the first column's left tap sits at byte 0, then
// one v_perm_b32 with a per-lane selector (set up once per thread) extracts each
// column's tap pair and one v_dot2_u32_u16 weighs it.
// (A block that walks down a strip of four tiles and prefetches the next footprint into registers while it
// computes -- 78 registers, two barriers per tile -- was slower, 0.78 against 0.70 ms for the seven levels of 512
// HD720 frames: eight independent blocks per CU overlap loads and arithmetic better than one block does itself.)
__global__ __launch_bounds__(RZ_THREADS) void k_resize_exact(PlaneBatch src, int sw, int sh, PlaneBatch dst, int dw,
                                                            int dh, const uint32_t *__restrict__ xtab,
                                                            const uint32_t *__restrict__ ytab, int tiles_x,
                                                            uint32_t inv_tiles_x, FrameGrid fg)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[RZ_FROWS * RZ_FP + 16];

    const int tid = threadIdx.x;
    const FrameBlock fb = frame_block(fg);  // a frame's tiles in raster order on one XCD
    const int ty = (int)magic_div((uint32_t)fb.block, inv_tiles_x), tx = fb.block - ty * tiles_x;
    const int x0 = tx * RZ_TW, y0 = ty * RZ_TH, f = fb.frame;
    // block-uniform footprint (tables are padded to whole tiles with their last entry)
    const int sx_lo = (int)(xtab[x0] & 0xffffu) & ~15;
    const int sx_hi = min((int)(xtab[x0 + RZ_TW - 1] & 0xffffu) + 1, sw - 1);
    const int sy_lo = (int)(ytab[y0] & 0xffffu);
    const int sy_hi = min((int)(ytab[y0 + RZ_TH - 1] & 0xffffu) + 1, sh - 1);
    // 16-byte pieces (planes and their pitch are 128-byte aligned); host checked: fits the LDS footprint
    const int cols16 = (sx_hi - sx_lo) / 16 + 1, rows = sy_hi - sy_lo + 1;
    const uint8_t *sp = src.base + f * src.plane_stride + (size_t)sy_lo * src.pitch + sx_lo;
    {
        // 32 lanes per footprint row (at most RZ_FP / 16 = 18 pieces: the others repeat the last one and store
        // nothing), eight rows per step: one address add per load, no division
        static_assert(RZ_FP / 16 <= 32 && RZ_THREADS == 256, "a footprint row fits 32 lanes, eight rows per step");
        constexpr int STEPS = (RZ_FROWS + 7) / 8;
        const int c = tid & 31, r0 = tid >> 5;
        const bool col_ok = c < cols16;
        const uint8_t *p = sp + (size_t)r0 * src.pitch + 16 * min(c, cols16 - 1);
        const size_t step = (size_t)8 * src.pitch;
        uint4 piece[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; u++) {  // all loads in flight before the first LDS write
            piece[u] = make_uint4(0, 0, 0, 0);
            if (u * 8 < rows) {  // block-uniform; rows past the footprint re-read its last row
                piece[u] = *reinterpret_cast<const uint4 *>(r0 + u * 8 < rows ? p : sp + (size_t)(rows - 1) * src.pitch);
                p += step;
            }
        }
        uint8_t *q = &s_src[r0 * RZ_FP + 16 * c];
#pragma unroll
        for (int u = 0; u < STEPS; u++)
            if (col_ok && r0 + u * 8 < rows) *reinterpret_cast<uint4 *>(q + u * 8 * RZ_FP) = piece[u];
    }
    // this thread's four columns
    const int lx = (tid & 31) * 4;
    const uint4 xt = *reinterpret_cast<const uint4 *>(xtab + x0 + lx);
    const uint32_t xe[4] = {xt.x, xt.y, xt.z, xt.w};
    const uint32_t xo0 = xe[0] & 0xffffu;
    const uint32_t base = xo0 & ~3u, rel0 = xo0 & 3u;  // first staged dword this thread reads, byte of xo0 in it
    uint32_t sel[4];
    us2 wx[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t d = min((xe[j] & 0xffffu) - xo0, (uint32_t)RZ_SPAN);  // host checked: <= RZ_SPAN
        sel[j] = d * 0x00010001u + 0x0c010c00u;                              // bytes d, d + 1 as two u16
        const uint32_t w1 = xe[j] >> 16;
        wx[j] = __builtin_bit_cast(us2, (256u - w1) | (w1 << 16));
    }
    uint32_t yes[RZ_TH / 8];  // row table entries of this thread's rows (the table is padded to whole tiles)
#pragma unroll
    for (int k = 0; k < RZ_TH / 8; k++) yes[k] = ytab[y0 + (tid >> 5) + k * 8];
    __syncthreads();

    uint8_t *dp = dst.base + f * dst.plane_stride;
    const uint8_t *col = &s_src[base - sx_lo];
    uint32_t out[RZ_TH / 8];  // this thread's four columns of its four rows
#pragma unroll
    for (int k = 0; k < RZ_TH / 8; k++) {
        const uint32_t ye = yes[k];
        const int yo = (int)(ye & 0xffffu);
        const uint32_t wy1 = ye >> 16;
        const us2 wy = __builtin_bit_cast(us2, (256u - wy1) | (wy1 << 16));
        const uint32_t *r0 = reinterpret_cast<const uint32_t *>(col + (yo - sy_lo) * RZ_FP);
        const uint32_t *r1 = reinterpret_cast<const uint32_t *>(col + (min(yo + 1, sh - 1) - sy_lo) * RZ_FP);
        const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        const uint32_t aL = __builtin_amdgcn_alignbyte(a1, a0, rel0), aH = __builtin_amdgcn_alignbyte(a2, a1, rel0);
        const uint32_t bL = __builtin_amdgcn_alignbyte(b1, b0, rel0), bH = __builtin_amdgcn_alignbyte(b2, b1, rel0);
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t h0 =
                __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(aH, aL, sel[j])), wx[j], 0u, false);
            const uint32_t h1 =
                __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(bH, bL, sel[j])), wx[j], 0u, false);
            v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, h0 | (h1 << 16)), wy, 32768u, false);
        }
        // results are bits 16..23 of each v (v < 2^24)
        const uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), hi = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0602u);
        out[k] = lo | (hi << 16);
    }
    // 4 x 4 transpose inside each group of four lanes (16 adjacent columns): lane l of the group ends up with the
    // 16 bytes of the group's row l and stores them at once
    static_assert(RZ_TH / 8 == 4, "four rows per thread, four lanes per 16-byte store");
    {
        const bool b0 = tid & 1, b1 = tid & 2;
#pragma unroll
        for (int k = 0; k < 4; k += 2) {  // rows (k, k+1) between lanes l and l ^ 1
            const uint32_t recv = __shfl_xor(b0 ? out[k] : out[k + 1], 1);
            if (b0) out[k] = recv; else out[k + 1] = recv;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {     // rows (k, k+2) between lanes l and l ^ 2
            const uint32_t recv = __shfl_xor(b1 ? out[k] : out[k + 2], 2);
            if (b1) out[k] = recv; else out[k + 2] = recv;
        }
    }
    const int gx = x0 + (lx & ~15);                   // first column of the group
    const int gy = y0 + (tid >> 5) + (tid & 3) * 8;   // the row this lane stores
    if (gx < dw && gy < dh)                           // pitch is padded: the last group may run past the width
        *reinterpret_cast<uint4 *>(dp + (size_t)gy * dst.pitch + gx) = make_uint4(out[0], out[1], out[2], out[3]);
}

// ------------------------------------------------- per-level keypoint ranking

constexpr int OS_THREADS = 1024;
constexpr int OS_K1CAP = 4096;  // candidates that can enter the Harris ranking of one level (ordinary variant)
constexpr int OS_LCAP = 2048;   // survivors of one level (ordinary variant)
// the variant for quotas beyond the ordinary one's (100 KB of LDS, one block per CU), also the one that takes over
// levels whose tied classes overflow a smaller variant's lists
constexpr int OS_K1CAP_XL = 8192, OS_LCAP_XL = kOrbLevelCap;
// Levels with a small quota (all eight at 2000 key points) go to blocks a quarter of the size with a third of the
// LDS: eight of them per CU instead of two, and barriers among four waves instead of sixteen (ranking of 512 HD720
// frames 0.30 -> 0.23 ms).  A block of the small or the ordinary variant whose level has more boundary ties than its
// lists hold (a score class of hundreds of equal corners at the cut: dense, repetitive texture) marks the level and
// leaves; a second pass of the largest variant takes the marked levels and nothing else.
constexpr int OS_THREADS_S = 256, OS_K1CAP_S = 1536, OS_LCAP_S = 768;
constexpr int kOrbLevelRedo = -1;  // lvl_n of a level a smaller variant has left to the largest one

typedef short ss2 __attribute__((ext_vector_type(2)));

// HarrisResponses(): 7x7 block, k = 0.04, Sobel-like integer gradients.  The
// three sums are integers, so the order of evaluation is free: the nine rows of
// the 9x9 footprint come in as three aligned dwords each (12 line accesses per
// key point instead of 81 byte gathers -- the ranking kernel is bound by those),
// are unpacked to pairs of 16-bit values and the gradients and their products
// are formed two columns at a time (packed 16-bit adds, v_dot2_i32_i16).
// Requires pitch % 4 == 0 and img 4-byte aligned.
__device__ __forceinline__ float harris_response(const uint8_t *img, int pitch, int x0, int y0)
{
    const size_t off = (size_t)(y0 - 4) * pitch + (size_t)(x0 - 4);
    const uint32_t r = (uint32_t)off & 3u;  // byte of column x0 - 4 in its dword
    const uint8_t *pa = img + (off & ~(size_t)3);
    uint32_t raw[9][3];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(pa + (size_t)k * pitch);
        raw[k][0] = q[0]; raw[k][1] = q[1]; raw[k][2] = q[2];
    }
    // row k as five pairs of 16-bit values: columns (0,1) (2,3) (4,5) (6,7) (8,-)
    auto unpack = [&](int k, uint32_t P[5]) {
        const uint32_t n0 = __builtin_amdgcn_alignbyte(raw[k][1], raw[k][0], r);
        const uint32_t n1 = __builtin_amdgcn_alignbyte(raw[k][2], raw[k][1], r);
        const uint32_t n2 = __builtin_amdgcn_alignbyte(0u, raw[k][2], r);
        P[0] = __builtin_amdgcn_perm(0u, n0, 0x0c010c00u); P[1] = __builtin_amdgcn_perm(0u, n0, 0x0c030c02u);
        P[2] = __builtin_amdgcn_perm(0u, n1, 0x0c010c00u); P[3] = __builtin_amdgcn_perm(0u, n1, 0x0c030c02u);
        P[4] = n2 & 0xffu;
    };
    uint32_t U[5], M[5], D[5];
    unpack(0, U);
    unpack(1, M);
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        unpack(i + 2, D);
        // s = up + 2 mid + dn (horizontal gradient: s[j+1] - s[j-1]); d = dn - up (vertical: d[j-1] + 2 d[j] + d[j+1])
        ss2 S[5], V[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const ss2 u = __builtin_bit_cast(ss2, U[k]), m = __builtin_bit_cast(ss2, M[k]), d = __builtin_bit_cast(ss2, D[k]);
            S[k] = u + m + m + d;
            V[k] = d - u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // columns 2k+1, 2k+2
            ss2 ix = S[k + 1] - S[k];
            const ss2 e = __builtin_bit_cast(
                ss2, __builtin_amdgcn_alignbyte(__builtin_bit_cast(uint32_t, V[k + 1]), __builtin_bit_cast(uint32_t, V[k]), 2));
            ss2 iy = V[k] + e + e + V[k + 1];
            if (k == 3) {  // column 8 is outside the 7x7 block
                ix = __builtin_bit_cast(ss2, __builtin_bit_cast(uint32_t, ix) & 0xffffu);
                iy = __builtin_bit_cast(ss2, __builtin_bit_cast(uint32_t, iy) & 0xffffu);
            }
            a = __builtin_amdgcn_sdot2(ix, ix, a, false);
            b = __builtin_amdgcn_sdot2(iy, iy, b, false);
            c = __builtin_amdgcn_sdot2(ix, iy, c, false);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) { U[k] = M[k]; M[k] = D[k]; }
    }
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    return (fa * fb - fc * fc - 0.04f * (fa + fb) * (fa + fb)) * scale_sq_sq;
}

// float -> unsigned key with the same order
__device__ __forceinline__ uint32_t ordered_key(float v)
{
    const uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

#ifdef VO_FE_STAMPS
// Diagnostic build only: cycle stamps of thread 0 of the first 32 frames' blocks, per level.
__device__ unsigned long long g_os_stamps[8 * 32 * 16];
#define OS_STAMP(i)                                                                                  \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 32)                                                     \
            g_os_stamps[(blockIdx.y * 32 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define OS_STAMP(i) do { } while (0)
#endif

// One block per (level, frame): what computeKeyPoints() does to the FAST
// corners of one pyramid level --
//   runByImageBorder(31) -> retainBest(2*quota) by FAST score ->
//   Harris response -> retainBest(quota) by Harris response --
// where retainBest(n) keeps every keypoint whose response is >= the n-th
// largest (all boundary ties survive).  OpenCV leaves the survivors in
// nth_element order (unspecified); here they are written in raster order.
template <int OS_THREADS, int OS_K1CAP, int OS_LCAP>
__global__ __launch_bounds__(OS_THREADS) void k_orb_select(OrbPlan plan, int batch, int first_level, int redo_only,
                                                          const uint8_t *__restrict__ pyr,
                                                          const uint32_t *__restrict__ cand,
                                                          const uint32_t *__restrict__ tile_count,
                                                          uint32_t *__restrict__ flat,
                                                          uint32_t *__restrict__ lvl_ras,
                                                          float *__restrict__ lvl_resp, int *__restrict__ lvl_n)
{
    __shared__ uint32_t s_buf[2 * OS_K1CAP];
    __shared__ uint32_t s_scan[OS_THREADS];
    uint32_t *const s_ras = s_buf, *const s_key = s_buf + OS_K1CAP;
    __shared__ __attribute__((aligned(16))) uint32_t s_oras[OS_LCAP];
    __shared__ uint32_t s_okey[OS_LCAP];
    __shared__ uint32_t s_hist[256];
    __shared__ int s_cut, s_n1, s_n2, s_remaining;
    __shared__ uint32_t s_prefix;

    // grid x = frame, y = level: consecutive blocks go to consecutive XCDs, so every XCD sees all levels
    // (with the level in x each XCD would serve a single level and the level-0 one would finish last)
    const int f = blockIdx.x, l = first_level + (int)blockIdx.y, tid = threadIdx.x;
    constexpr bool kSmall = OS_K1CAP < OS_K1CAP_XL;  // this variant may leave a level to the largest one
    if (redo_only && lvl_n[f * plan.nlevels + l] != kOrbLevelRedo) return;
    const OrbLevel L = plan.lv[l];
    const int w = L.w, h = L.h, quota = L.quota, border = kFBorder;
    OS_STAMP(0);
    // the level's corners as one list (s_buf is free until the gather below)
    uint32_t *c = flat + L.cand_off + (size_t)f * L.tiles * kTileCap;
    const int M = flatten_tile_lists<OS_THREADS>(cand + L.cand_off + (size_t)f * L.tiles * kTileCap,
                                                 tile_count + L.count_off + (size_t)f * L.tiles, L.tiles, c, s_buf,
                                                 s_scan);
    OS_STAMP(1);
    const uint8_t *img = pyr + (size_t)f * plan.frame_bytes + L.off;
    const bool size_ok = w > 2 * border && h > 2 * border && quota > 0;

    for (int i = tid; i < 256; i += OS_THREADS) s_hist[i] = 0;
    if (tid == 0) { s_n1 = 0; s_n2 = 0; s_cut = -1; }
    __syncthreads();

    // 1. border filter + FAST-score histogram
    auto in_border = [&](uint32_t pos) {
        const int y = key_y(pos), x = key_x(pos);
        return x >= border && x < w - border && y >= border && y < h - border;
    };
    if (size_ok)
        for (int i = tid; i < M; i += OS_THREADS) {
            const uint32_t k = c[i];
            if (in_border(k & 0xffffffu)) atomicAdd(&s_hist[k >> 24], 1u);
        }
    __syncthreads();
    OS_STAMP(2);
    {
        // bins are 255 - score: the first bin at which the running count covers 2*quota corners;
        // every bin <= cut survives (all of them when fewer than 2*quota exist)
        const uint32_t hv = tid < 256 ? s_hist[tid] : 0u;
        const uint32_t inc = block_scan_inclusive<OS_THREADS>(hv, s_scan);
        const uint32_t target = 2u * (uint32_t)quota;
        if (tid < 256 && size_ok) {
            if (inc >= target && inc - hv < target) s_cut = tid;
            else if (tid == 255 && inc < target) s_cut = 255;
        }
    }
    __syncthreads();

    OS_STAMP(3);
    // 2. gather the survivors
    const int cut = s_cut;
    if (cut >= 0)
        for (int i = tid; i < M; i += OS_THREADS) {
            const uint32_t k = c[i];
            if ((int)(k >> 24) <= cut && in_border(k & 0xffffffu)) {
                const int slot = atomicAdd(&s_n1, 1);
                if (slot < OS_K1CAP) s_ras[slot] = k & 0xffffffu;
            }
        }
    __syncthreads();
    if (kSmall && s_n1 > OS_K1CAP) {  // (block-uniform)
        if (tid == 0) lvl_n[f * plan.nlevels + l] = kOrbLevelRedo;
        return;
    }
    const int K1 = min(s_n1, OS_K1CAP);

    OS_STAMP(4);
    // 3. Harris response of each survivor
    for (int i = tid; i < K1; i += OS_THREADS) {
        const uint32_t ras = s_ras[i];
        const int y = key_y(ras), x = key_x(ras);
        s_key[i] = ordered_key(harris_response(img, L.pitch, x, y));
    }
    __syncthreads();

    OS_STAMP(5);
    // 4. the quota-th largest key by an MSB-first radix select
    uint32_t threshold = 0;
    if (K1 > quota) {
        if (tid == 0) { s_prefix = 0; s_remaining = quota; }
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += OS_THREADS) s_hist[i] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            const uint32_t hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < K1; i += OS_THREADS) {
                const uint32_t k = s_key[i];
                if ((k & hi_mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            {
                // the digit d with sum(hist[j > d]) < rem <= sum(hist[j >= d]): thread t owns digit 255 - t
                const uint32_t rem = (uint32_t)s_remaining;
                const uint32_t hv = tid < 256 ? s_hist[255 - tid] : 0u;
                const uint32_t inc = block_scan_inclusive<OS_THREADS>(hv, s_scan);
                if (tid < 256 && inc >= rem && inc - hv < rem) {
                    s_remaining = (int)(rem - (inc - hv));
                    s_prefix = prefix | ((uint32_t)(255 - tid) << shift);
                }
            }
            __syncthreads();
        }
        threshold = s_prefix;
    }

    OS_STAMP(6);
    // 5. survivors of the second cut, sorted by raster index
    for (int i = tid; i < K1; i += OS_THREADS) {
        if (s_key[i] >= threshold) {
            const int slot = atomicAdd(&s_n2, 1);
            if (slot < OS_LCAP) { s_oras[slot] = s_ras[i]; s_okey[slot] = s_key[i]; }
        }
    }
    __syncthreads();
    if (kSmall && s_n2 > OS_LCAP) {
        if (tid == 0) lvl_n[f * plan.nlevels + l] = kOrbLevelRedo;
        return;
    }
    const int K2 = min(s_n2, OS_LCAP);
    OS_STAMP(7);
    // raster order by counting: the rank of a survivor is the number of survivors with a smaller raster index
    // (all distinct); every lane of a wave reads the same four entries per step (LDS broadcast)
    for (int i = K2 + tid; i < ((K2 + 3) & ~3); i += OS_THREADS) s_oras[i] = 0xffffffffu;
    __syncthreads();
    OS_STAMP(8);
    const size_t o = ((size_t)f * plan.nlevels + l) * kOrbLevelCap;
    for (int i = tid; i < K2; i += OS_THREADS) {
        const uint32_t me = s_oras[i];
        int rank = 0;
        for (int j = 0; j < K2; j += 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&s_oras[j]);
            rank += (int)(v.x < me) + (int)(v.y < me) + (int)(v.z < me) + (int)(v.w < me);
        }
        lvl_ras[o + rank] = me;
        lvl_resp[o + rank] = key_to_float(s_okey[i]);
    }
    OS_STAMP(9);
    if (tid == 0) lvl_n[f * plan.nlevels + l] = K2;
}

// -------------------------------------------------- orientation + keypoints

// half-width of the radius-15 disc at row |v| (ORB's umax table)
__device__ __forceinline__ constexpr int kUmaxC(int v)
{
    constexpr int t[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    return t[v];
}

// cv::fastAtan2, scalar path (degrees)
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

constexpr int OF_THREADS = 256;
constexpr int OF_LANES = 8;                        // lanes per keypoint
constexpr int OF_KPB = OF_THREADS / OF_LANES;      // keypoints per block

// 8 lanes per output keypoint: concatenates the levels' survivors (level by
// level), computes the intensity-centroid angle over the radius-15 disc of
// the keypoint's level image (ICAngles()) and writes the cv::KeyPoint with
// pt scaled to level-0 coordinates, size = 31 * scale, octave = level.
// Lane q of a group owns the four columns u = -15 + 4q .. -12 + 4q (the last
// one, u = 16, is masked): one unaligned dword per row, all 31 issued before
// the first use; the disc is applied as a byte mask per |v| (table in LDS) and
// the moments are formed with 4-byte dot products:
//   m10 = sum u I = sum (u + 16) I - 16 sum I,   m01 = sum_v v (S(+v) - S(-v)).
__global__ __launch_bounds__(OF_THREADS) void k_orb_finish(OrbPlan plan, const uint8_t *__restrict__ pyr,
                                                          const uint32_t *__restrict__ lvl_ras,
                                                          const float *__restrict__ lvl_resp,
                                                          const int *__restrict__ lvl_n, vo_kp *__restrict__ kps,
                                                          int kp_stride, int kp_cap, int *__restrict__ nkp,
                                                          FrameGrid fg)
{
    __shared__ uint32_t s_mask[16][OF_LANES];
    const FrameBlock fb = frame_block(fg);
    const int f = fb.frame, gl = threadIdx.x & (OF_LANES - 1);
    const int i = fb.block * OF_KPB + (threadIdx.x >> 3);
    if (threadIdx.x < 16 * OF_LANES) {
        // byte j of s_mask[v][q]: column u = -15 + 4q + j lies inside the disc at row |v|
        const int v = threadIdx.x >> 3, q = threadIdx.x & 7;
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (abs(-15 + 4 * q + j) <= kUmaxC(v)) m |= 0xffu << (8 * j);
        s_mask[v][q] = m;
    }
    int level = -1, j = 0, total = 0;
    for (int l = 0; l < plan.nlevels; l++) {
        const int n = lvl_n[f * plan.nlevels + l];
        if (level < 0 && i < total + n) { level = l; j = i - total; }
        total += n;
    }
    total = min(total, kp_cap);  // vo_cfg.max_features: what the caller's arrays hold (kp_stride is padded)
    if (fb.block == 0 && threadIdx.x == 0) nkp[f] = total;
    const bool live = level >= 0 && i < total;
    const int lv = live ? level : 0;
    const int Lpitch = plan.lv[lv].pitch;
    const float Lscale = plan.lv[lv].scale;
    const size_t Loff = plan.lv[lv].off;
    const size_t o = ((size_t)f * plan.nlevels + lv) * kOrbLevelCap + (live ? j : 0);
    const uint32_t ras = live ? lvl_ras[o] : 0u;
    const int y = key_y(ras), x = key_x(ras);
    // block-uniform base + 32-bit offsets (a frame's pyramid is far below 4 GB): one add per row
    const uint8_t *frame = pyr + (size_t)f * plan.frame_bytes;

    // the lane's last byte (u = 16 for q = 7) may lie one past the patch: rows are padded, the byte is masked
    // (idle groups re-read the first bytes of the frame)
    uint32_t off = live ? (uint32_t)Loff + (uint32_t)(y - 15) * (uint32_t)Lpitch + (uint32_t)(x - 15 + 4 * gl) : 0u;
    const uint32_t step = live ? (uint32_t)Lpitch : 0u;
    uint32_t val[31];
#pragma unroll
    for (int k = 0; k < 31; k++) {
        __builtin_memcpy(&val[k], frame + off, 4);  // unaligned dword
        off += step;
    }
    __syncthreads();
    // weights u + 16 of the lane's four columns (1 .. 32)
    const uint32_t w0 = (uint32_t)(1 + 4 * gl);
    const uint32_t wts = w0 | ((w0 + 1) << 8) | ((w0 + 2) << 16) | ((w0 + 3) << 24);
    const uint32_t c = val[15] & s_mask[0][gl];
    uint32_t accw = __builtin_amdgcn_udot4(c, wts, 0u, false);
    uint32_t acc1 = __builtin_amdgcn_udot4(c, 0x01010101u, 0u, false);
    int m01 = 0;
#pragma unroll
    for (int v = 1; v <= 15; v++) {
        const uint32_t m = s_mask[v][gl];
        const uint32_t a = val[15 + v] & m, b = val[15 - v] & m;
        accw = __builtin_amdgcn_udot4(a, wts, accw, false);
        accw = __builtin_amdgcn_udot4(b, wts, accw, false);
        const uint32_t sa = __builtin_amdgcn_udot4(a, 0x01010101u, 0u, false);
        const uint32_t sb = __builtin_amdgcn_udot4(b, 0x01010101u, 0u, false);
        acc1 += sa + sb;
        m01 += v * ((int)sa - (int)sb);
    }
    int m10 = (int)accw - 16 * (int)acc1;
#pragma unroll
    for (int sh = OF_LANES / 2; sh > 0; sh >>= 1) {
        m10 += __shfl_xor(m10, sh);
        m01 += __shfl_xor(m01, sh);
    }
    if (live && gl == 0) {
        vo_kp kp;
        kp.x = (float)x * Lscale;
        kp.y = (float)y * Lscale;
        kp.size = 31 * Lscale;
        kp.angle = fast_atan2_deg((float)m01, (float)m10);
        kp.response = lvl_resp[o];
        kp.octave = level;
        kps[(size_t)f * kp_stride + i] = kp;
    }
}

// ------------------------------------------------------------ steered BRIEF

// 32 lanes per keypoint, one descriptor byte per lane: computeOrbDescriptors()
// with WTA_K = 2 on the blurred level image of the keypoint's octave, the
// 31x31 pattern rotated by the keypoint angle per sample (float32, cvRound).
// The pattern's largest radius is 18.4 px, so every rotated sample lies within
// +-18 px: the 37 x 37 patch is staged in LDS with dword loads first.
constexpr int BS_R = kBriefReach;
// low 24 bits of the float 1.5 * 2^23 + n are 0x400000 + n
constexpr int kBriefRoundBias = 0x400000;
__device__ __forceinline__ int brief_round_bits(float v)
{
    return (int)(__float_as_uint(v + 12582912.f) & 0xffffffu);
}
constexpr int BS_PWD = (2 * BS_R + 1 + 3 + 3) / 4;
constexpr int BS_KPG = 2;             // key points a 32-lane group handles one after the other
constexpr int BS_KPB = 8 * BS_KPG;    // key points per block
// The loads of both patches of a group are issued up front; while the group computes the first descriptor from
// LDS the second patch is still in flight (it waits in registers and goes to LDS afterwards): the patch gathers,
// not the arithmetic, bound this kernel.  A group is half a wave and only touches its own patch, so
// nothing but program order is needed between its LDS writes and reads; the one block barrier is for the
// pattern and the cos / sin table.
__global__ __launch_bounds__(256) void k_brief_steered(OrbPlan plan, const uint8_t *__restrict__ pyr_blur,
                                                      const vo_kp *__restrict__ kps, int kp_stride,
                                                      const int *__restrict__ nkp,
                                                      const float *__restrict__ pattern,
                                                      uint8_t *__restrict__ desc, FrameGrid fg)
{
    __shared__ float4 s_pat[256];  // [test of a byte][byte]: x0 y0 x1 y1
    __shared__ __attribute__((aligned(16))) uint32_t s_patch[8][(2 * BS_R + 1) * BS_PWD];
    __shared__ float s_cs[BS_KPB][2];
    const FrameBlock fb = frame_block(fg);
    s_pat[threadIdx.x] = reinterpret_cast<const float4 *>(pattern)[threadIdx.x];
    const int f = fb.frame;
    const int g = threadIdx.x >> 5, k0 = fb.block * BS_KPB + g * BS_KPG;  // the group's first key point
    const int byte = threadIdx.x & 31;
    const int n_kp = nkp[f];
    // double-precision cos / sin of the block's angles: one lane each, in a single wave
    if (threadIdx.x < BS_KPB && fb.block * BS_KPB + (int)threadIdx.x < n_kp) {
        float angle = kps[(size_t)f * kp_stride + fb.block * BS_KPB + threadIdx.x].angle;
        angle *= (float)(3.14159265358979323846 / 180.f);
        s_cs[threadIdx.x][0] = (float)cos((double)angle);
        s_cs[threadIdx.x][1] = (float)sin((double)angle);
    }
    // block-uniform base + 32-bit offsets (a frame's pyramid is far below 4 GB).  The group's 32 lanes cover two
    // rows of 16 dwords per step (11 of the 16 are patch columns, the others repeat the last one and are not
    // stored): one add per load, and all loads are issued before the first LDS write
    const uint8_t *frame = pyr_blur + (size_t)f * plan.frame_bytes;
    constexpr int ROWS = 2 * BS_R + 1, PER = (ROWS + 1) / 2;
    const int rh = byte >> 4, c = min(byte & 15, BS_PWD - 1);
    const bool col_ok = (byte & 15) < BS_PWD;
    uint32_t v[BS_KPG][PER];
    // issue the patch loads of key point k (k < n_kp) into register set i; returns the patch column of its centre
    auto issue = [&](int i, int k) -> int {
        const vo_kp kp = kps[(size_t)f * kp_stride + k];
        const OrbLevel L = plan.lv[kp.octave];
        const float inv = 1.f / L.scale;
        const int cx = __float2int_rn(kp.x * inv), cy = __float2int_rn(kp.y * inv);
        const int x_al = (cx - BS_R) & ~3;  // keypoints are >= 31 px inside their level: never negative
        uint32_t off = (uint32_t)L.off + __umul24((uint32_t)(cy - BS_R + rh), (uint32_t)L.pitch) + (uint32_t)(x_al + 4 * c);
        const uint32_t step = 2u * (uint32_t)L.pitch;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            // the last step has a single row: the upper half of the group re-reads it
            if (u == PER - 1 && (ROWS & 1)) off -= rh ? (uint32_t)L.pitch : 0u;
            v[i][u] = *reinterpret_cast<const uint32_t *>(frame + off);
            off += step;
        }
        return cx - x_al;
    };
    // the loads of all of the group's patches are issued up front, one register set per key point
    int col0[BS_KPG];
#pragma unroll
    for (int i = 0; i < BS_KPG; i++) col0[i] = k0 + i < n_kp ? issue(i, k0 + i) : 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BS_KPG; i++) {
        const int k = k0 + i;
        if (k >= n_kp) break;  // uniform over the group (half a wave)
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (col_ok && 2 * u + rh < ROWS) s_patch[g][(2 * u + rh) * BS_PWD + c] = v[i][u];
        const int col_cur = col0[i];
        const float a = s_cs[g * BS_KPG + i][0], b = s_cs[g * BS_KPG + i][1];
        const uint8_t *center = reinterpret_cast<const uint8_t *>(s_patch[g]) + BS_R * (BS_PWD * 4) + col_cur -
                                kBriefRoundBias * (BS_PWD * 4 + 1);
        uint32_t val = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const float4 p = s_pat[t * 32 + byte];
            const float x0 = p.x, y0 = p.y, x1 = p.z, y1 = p.w;
            // cvRound by adding 1.5 * 2^23: the sum's low mantissa bits hold the coordinate rounded to nearest even
            // (|coordinate| < 2^22); the bias of the bit patterns is taken off the base address once
            const int ix0 = brief_round_bits(x0 * a - y0 * b), iy0 = brief_round_bits(x0 * b + y0 * a);
            const int ix1 = brief_round_bits(x1 * a - y1 * b), iy1 = brief_round_bits(x1 * b + y1 * a);
            const int t0 = center[__mul24(iy0, BS_PWD * 4) + ix0];
            const int t1 = center[__mul24(iy1, BS_PWD * 4) + ix1];
            val |= (uint32_t)(t0 < t1) << t;
        }
        desc[((size_t)f * kp_stride + k) * 32 + byte] = (uint8_t)val;
    }
}

}  // namespace

// level geometry and quotas exactly as ORB derives them:
// scale_l = float(pow(double(scaleFactor), l)), size_l = cvRound(size / scale_l),
// quota_l = cvRound(N (1-f) / (1-f^L) f^l) with f = float(1/scaleFactor), the
// last level taking the remainder.
bool orb_make_plan(int w, int h, float scale_factor, int nlevels, int nfeatures, int batch, OrbPlan *out)
{
    if (nlevels < 1 || nlevels > kOrbMaxLevels || !(scale_factor > 1.f) || scale_factor > 2.f) return false;
    OrbPlan p{};
    p.nlevels = nlevels;
    const float factor = (float)(1.0 / scale_factor);
    float per = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    size_t off = 0, coff = 0, noff = 0, toff = 0;
    for (int l = 0; l < nlevels; l++) {
        OrbLevel &L = p.lv[l];
        L.scale = (float)pow((double)scale_factor, (double)l);
        L.w = (int)lrintf((float)w / L.scale);
        L.h = (int)lrintf((float)h / L.scale);
        if (L.w < 7 || L.h < 7) return false;
        L.pitch = (L.w + 127) / 128 * 128;
        L.off = off;
        off += (size_t)L.pitch * ((L.h + 15) / 16 * 16);
        if (l < nlevels - 1) {
            L.quota = (int)lrintf(per);
            sum += L.quota;
            per *= factor;
        } else {
            L.quota = nfeatures - sum > 0 ? nfeatures - sum : 0;
        }
        if (L.quota > OS_LCAP_XL - 256 || 2 * L.quota + 512 > OS_K1CAP_XL) return false;
        L.tiles = tiles_of(L.w, L.h);
        if (L.tiles > 2 * OS_K1CAP_XL) return false;  // (a variant lists the tile counts in its candidate buffer)
        L.cand_off = coff;
        coff += (size_t)L.tiles * kTileCap * batch;
        L.count_off = noff;
        noff += (size_t)L.tiles * batch;
        L.xtab_off = toff;
        toff += (size_t)L.pitch;
        L.ytab_off = toff;
        toff += (size_t)((L.h + kResizeRowPad - 1) / kResizeRowPad * kResizeRowPad);
        if (l > 0) {
            // the resize kernel's LDS footprint of one output tile
            const OrbLevel &S = p.lv[l - 1];
            const double sx = (double)S.w / L.w, sy = (double)S.h / L.h;
            if (RZ_TW * sx + 24 > RZ_FP || RZ_TH * sy + 3 > RZ_FROWS) return false;
            // a thread's four columns must start within RZ_SPAN source pixels of the first
            std::vector<uint32_t> xt((size_t)L.pitch);
            orb_axis_table(S.w, L.w, L.pitch, xt.data());
            for (int x = 0; x + 3 < L.pitch; x += 4)
                if ((xt[x + 3] & 0xffffu) - (xt[x] & 0xffffu) > (uint32_t)RZ_SPAN) return false;
        }
    }
    p.frame_bytes = off;
    p.cand_total = coff;
    p.count_total = noff;
    p.tab_total = toff;
    *out = p;
    return true;
}

// One axis of cv::resize(INTER_LINEAR_EXACT) as a table: entry d = source
// offset | (8.8 weight of the second tap) << 16, with scale = 1 / (dst/src) in
// double, sample position scale * (d + 0.5) - 0.5 and replication outside
// [0, src-1); padded to `padded` entries with the last one.
void orb_axis_table(int src_n, int dst_n, int padded, uint32_t *out)
{
    const double inv_scale = (double)dst_n / (double)src_n;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < padded; d++) {
        const int dd = d < dst_n ? d : dst_n - 1;
        const double f = scale * ((double)dd + 0.5) - 0.5;
        const int i = (int)floor(f);
        uint32_t ofs = 0, w1 = 0;
        if (i >= 0 && src_n > 1) {
            if (i < src_n - 1) {
                ofs = (uint32_t)i;
                w1 = (uint32_t)lrint((f - (double)i) * 256.0);
            } else {
                ofs = (uint32_t)(src_n - 1);
            }
        }
        out[d] = ofs | (w1 << 16);
    }
}

#ifdef VO_FE_STAMPS
extern "C" int vo_debug_os_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_stamps), sizeof(unsigned long long) * 8 * 32 * 16);
}
#endif

void launch_resize_exact(PlaneBatch src, int sw, int sh, PlaneBatch dst, int dw, int dh, const uint32_t *xtab,
                         const uint32_t *ytab, int batch, hipStream_t s)
{
    const int tiles_x = (dw + RZ_TW - 1) / RZ_TW, tiles_y = (dh + RZ_TH - 1) / RZ_TH;
    hipLaunchKernelGGL(k_resize_exact, dim3((unsigned)(tiles_x * tiles_y * batch)), dim3(RZ_THREADS), 0, s, src, sw,
                       sh, dst, dw, dh, xtab, ytab, tiles_x, magic_inv(tiles_x),
                       frame_grid(tiles_x * tiles_y, batch));
}

void launch_orb_select(const OrbPlan &plan, int batch, const uint8_t *pyr, const uint32_t *cand,
                       const uint32_t *tile_count, uint32_t *flat, uint32_t *lvl_ras, float *lvl_resp, int *lvl_n,
                       hipStream_t s)
{
    // Quotas fall with the level: levels [0, xl) need the largest variant, [xl, sm) the ordinary one, [sm, n) fit the
    // small one (margins for boundary ties: 256 at the Harris cut, 512 at the FAST-score cut)
    auto fits = [&](int l, int lcap, int k1cap) {
        return plan.lv[l].quota <= lcap - 256 && 2 * plan.lv[l].quota + 512 <= k1cap && plan.lv[l].tiles <= 2 * k1cap;
    };
    int sm = plan.nlevels;
    // (a handful of frames cannot fill the chip with blocks anyway: there the larger block finishes a level sooner,
    // 0.54 against 0.56 ms for one HD720 frame through the facade)
    if (batch >= 32)
        while (sm > 0 && fits(sm - 1, OS_LCAP_S, OS_K1CAP_S)) sm--;
    int xl = sm;
    while (xl > 0 && fits(xl - 1, OS_LCAP, OS_K1CAP)) xl--;
    // (VO_OS_SMALL_FROM=n: levels below n do not use the small variant, for comparisons)
    static const int force = getenv("VO_OS_SMALL_FROM") ? atoi(getenv("VO_OS_SMALL_FROM")) : -1;
    if (force >= 0) sm = std::max(sm, std::min(force, plan.nlevels));
    if (xl > 0)
        hipLaunchKernelGGL((k_orb_select<OS_THREADS, OS_K1CAP_XL, OS_LCAP_XL>), dim3(batch, xl), dim3(OS_THREADS), 0, s,
                           plan, batch, 0, 0, pyr, cand, tile_count, flat, lvl_ras, lvl_resp, lvl_n);
    if (sm > xl)
        hipLaunchKernelGGL((k_orb_select<OS_THREADS, OS_K1CAP, OS_LCAP>), dim3(batch, sm - xl), dim3(OS_THREADS), 0, s,
                           plan, batch, xl, 0, pyr, cand, tile_count, flat, lvl_ras, lvl_resp, lvl_n);
    if (plan.nlevels > sm)
        hipLaunchKernelGGL((k_orb_select<OS_THREADS_S, OS_K1CAP_S, OS_LCAP_S>), dim3(batch, plan.nlevels - sm),
                           dim3(OS_THREADS_S), 0, s, plan, batch, sm, 0, pyr, cand, tile_count, flat, lvl_ras, lvl_resp,
                           lvl_n);
    if (plan.nlevels > xl)  // the levels that overflowed a smaller variant
        hipLaunchKernelGGL((k_orb_select<OS_THREADS, OS_K1CAP_XL, OS_LCAP_XL>), dim3(batch, plan.nlevels - xl),
                           dim3(OS_THREADS), 0, s, plan, batch, xl, /*redo_only=*/1, pyr, cand, tile_count, flat,
                           lvl_ras, lvl_resp, lvl_n);
}

void launch_orb_finish(const OrbPlan &plan, int batch, const uint8_t *pyr, const uint32_t *lvl_ras,
                       const float *lvl_resp, const int *lvl_n, vo_kp *kps, int kp_stride, int kp_cap, int *nkp,
                       hipStream_t s)
{
    const int per_frame = (kp_stride + OF_KPB - 1) / OF_KPB;
    hipLaunchKernelGGL(k_orb_finish, dim3((unsigned)(per_frame * batch)), dim3(OF_THREADS), 0, s, plan, pyr, lvl_ras,
                       lvl_resp, lvl_n, kps, kp_stride, kp_cap, nkp, frame_grid(per_frame, batch));
}

void launch_brief_steered(const OrbPlan &plan, int batch, const uint8_t *pyr_blur, const vo_kp *kps, int kp_stride,
                          const int *nkp, const float *pattern, uint8_t *desc, hipStream_t s)
{
    const int per_frame = (kp_stride + BS_KPB - 1) / BS_KPB;
    hipLaunchKernelGGL(k_brief_steered, dim3((unsigned)(per_frame * batch)), dim3(256), 0, s, plan, pyr_blur, kps,
                       kp_stride, nkp, pattern, desc, frame_grid(per_frame, batch));
}

}  // namespace vo
