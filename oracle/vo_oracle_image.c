/* CPU oracle, image stages and matcher.  TEST INFRASTRUCTURE ONLY -- see
 * vo_oracle.h.  PARITY UNPINNED (no OpenCV here, no reference test vectors).
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vo_orb_pattern.h"

static inline int ora_cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------ gray */

/* [OCV-recall] OpenCV 4.x RGB2Gray<uchar>: 15-bit coefficients
 * B 3735, G 19235, R 9798 (sum 32768), rounded descale.
 * reference call sites: src/frame_processor.cpp:206, :223,
 * src/feature_detector.cpp:46, src/image_processor.cpp:35 */
void ora_gray(const uint8_t *src, int w, int h, int stride, int cn,
              uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * stride;
        uint8_t *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++, s += cn)
            d[x] = (uint8_t)((s[0] * 3735 + s[1] * 19235 + s[2] * 9798 + (1 << 14)) >> 15);
    }
}

/* ------------------------------------------------------------------ FAST */

/* [OCV-recall] radius-3 Bresenham circle, OpenCV order */
static const int fast_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int fast_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* [OCV-recall] cornerScore<16> of fast_score.cpp: largest threshold for which
 * the pixel still passes the 9-of-16 segment test. */
static int fast_corner_score(const uint8_t *p, const int *off, int threshold)
{
    int d[25];
    int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k]];

    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++)
            if (d[k + j] < a) a = d[k + j];
        int t = a < d[k] ? a : d[k];
        if (t > a0) a0 = t;
        t = a < d[k + 9] ? a : d[k + 9];
        if (t > a0) a0 = t;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++)
            if (d[k + j] > b) b = d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++)
            if (d[k + j] > b) b = d[k + j];
        int t = b > d[k] ? b : d[k];
        if (t < b0) b0 = t;
        t = b > d[k + 9] ? b : d[k + 9];
        if (t < b0) b0 = t;
    }
    return -b0 - 1;
}

/* [OCV-recall] FAST_t<16>: a pixel is a corner when more than 8 contiguous
 * ring pixels are all brighter than v+t or all darker than v-t. */
static int fast_is_corner(const uint8_t *p, const int *off, int t)
{
    int v = p[0];
    int cb = v + t, c_b = v - t;
    /* ring pixels brighter / darker as 16-bit masks; a run of more than 8 set bits on the circular mask:
     * doubling the mask makes runs linear, and-ing shifted copies shrinks every run by 1, 2, 4, 1 = 8 bits */
    uint32_t mb = 0, md = 0;
    for (int k = 0; k < 16; k++) {
        int x = p[off[k]];
        mb |= (uint32_t)(x > cb) << k;
        md |= (uint32_t)(x < c_b) << k;
    }
    mb |= mb << 16; md |= md << 16;
    mb &= mb >> 1; mb &= mb >> 2; mb &= mb >> 4; mb &= mb >> 1;
    md &= md >> 1; md &= md >> 2; md &= md >> 4; md &= md >> 1;
    return (mb | md) != 0;
}

#if defined(__AVX2__)
#include <immintrin.h>
/* bit i set: pixel row[x + i] passes the compass test -- both opposite pairs (up/down at distance 3, left/right
 * at distance 3) hold a pixel brighter than v + t, or both a pixel darker than v - t.  A 9-arc of the 16-ring
 * contains one pixel of every opposite pair, so pixels that fail cannot be corners: the same role as the
 * 16-pixels-at-a-time early-out of cv::FAST's SIMD loop, 32 pixels per step here. */
static inline uint32_t fast_compass_mask32(const uint8_t *p, int stride, __m256i t)
{
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const __m256i hi = _mm256_adds_epu8(v, t), lo = _mm256_subs_epu8(v, t);
    const __m256i up = _mm256_loadu_si256((const __m256i *)(p - 3 * stride));
    const __m256i dn = _mm256_loadu_si256((const __m256i *)(p + 3 * stride));
    const __m256i lf = _mm256_loadu_si256((const __m256i *)(p - 3));
    const __m256i rt = _mm256_loadu_si256((const __m256i *)(p + 3));
    const __m256i zero = _mm256_setzero_si256();
    /* a > b  <=>  saturating a - b != 0 */
    const __m256i b1 = _mm256_subs_epu8(_mm256_max_epu8(up, dn), hi), b2 = _mm256_subs_epu8(_mm256_max_epu8(lf, rt), hi);
    const __m256i d1 = _mm256_subs_epu8(lo, _mm256_min_epu8(up, dn)), d2 = _mm256_subs_epu8(lo, _mm256_min_epu8(lf, rt));
    const __m256i nb = _mm256_or_si256(_mm256_cmpeq_epi8(b1, zero), _mm256_cmpeq_epi8(b2, zero));  /* not bright */
    const __m256i nd = _mm256_or_si256(_mm256_cmpeq_epi8(d1, zero), _mm256_cmpeq_epi8(d2, zero));  /* not dark */
    return ~(uint32_t)_mm256_movemask_epi8(_mm256_and_si256(nb, nd));
}
#endif

void ora_fast_score_map(const uint8_t *img, int w, int h, int stride, int thr,
                        uint8_t *score, int sstride)
{
    int off[25];
    if (thr < 0) thr = 0;
    if (thr > 255) thr = 255;
    for (int k = 0; k < 25; k++) off[k] = fast_dy[k & 15] * stride + fast_dx[k & 15];
    for (int y = 0; y < h; y++) memset(score + (size_t)y * sstride, 0, (size_t)w);
    for (int y = 3; y < h - 3; y++) {
        const uint8_t *row = img + (size_t)y * stride;
        uint8_t *srow = score + (size_t)y * sstride;
        int x = 3;
#if defined(__AVX2__)
        const __m256i t8 = _mm256_set1_epi8((char)thr);
        /* 32 pixels per step while the loads (x - 3 .. x + 34) stay inside the row */
        for (; x + 35 <= w; x += 32) {
            uint32_t m = fast_compass_mask32(row + x, stride, t8);
            while (m) {
                const int i = __builtin_ctz(m);
                m &= m - 1;
                const uint8_t *p = row + x + i;
                if (fast_is_corner(p, off, thr)) srow[x + i] = (uint8_t)fast_corner_score(p, off, thr);
            }
        }
#endif
        for (; x < w - 3; x++) {
            const uint8_t *p = row + x;
            /* high-speed reject on the four compass points */
            int v = p[0];
            int hi = v + thr, lo = v - thr;
            int p0 = p[off[0]], p8 = p[off[8]];
            if (!((p0 > hi || p8 > hi) || (p0 < lo || p8 < lo))) continue;
            if (fast_is_corner(p, off, thr))
                srow[x] = (uint8_t)fast_corner_score(p, off, thr);
        }
    }
}

int ora_fast(const uint8_t *img, int w, int h, int stride, int thr,
             ora_kp *out, int cap)
{
    if (w < 7 || h < 7) return 0;
    uint8_t *score = (uint8_t *)malloc((size_t)w * h);
    ora_fast_score_map(img, w, h, stride, thr, score, w);
    int n = 0;
    /* [OCV-recall] corners live in x,y in [3, dim-4]; a corner is kept when
     * its score is strictly greater than all 8 neighbours (non-corners = 0). */
    for (int y = 3; y < h - 3; y++) {
        const uint8_t *c = score + (size_t)y * w;
        const uint8_t *u = c - w, *d = c + w;
        for (int x = 3; x < w - 3; x++) {
            if (((x - 3) & 7) == 0 && x + 8 <= w - 3) {
                /* eight zero scores at once (most of the map is zero) */
                uint64_t z;
                memcpy(&z, c + x, 8);
                if (!z) { x += 7; continue; }
            }
            int s = c[x];
            if (!s) continue;
            if (s > u[x - 1] && s > u[x] && s > u[x + 1] && s > c[x - 1] &&
                s > c[x + 1] && s > d[x - 1] && s > d[x] && s > d[x + 1]) {
                if (n < cap) {
                    out[n].x = (float)x;
                    out[n].y = (float)y;
                    out[n].size = 7.f;
                    out[n].angle = -1.f;
                    out[n].response = (float)s;
                    out[n].octave = 0;
                }
                n++;
            }
        }
    }
    free(score);
    return n;
}

/* ---------------------------------------------------------------- select */

static int g_sel_w;
static int cmp_resp_desc_raster_asc(const void *pa, const void *pb)
{
    const ora_kp *a = (const ora_kp *)pa, *b = (const ora_kp *)pb;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    long ra = (long)a->y * g_sel_w + (long)a->x;
    long rb = (long)b->y * g_sel_w + (long)b->x;
    return ra < rb ? -1 : (ra > rb ? 1 : 0);
}

int ora_select_top(ora_kp *kps, int n, int w, int max_features)
{
    if (n <= max_features) return n;
    g_sel_w = w;
    qsort(kps, (size_t)n, sizeof(ora_kp), cmp_resp_desc_raster_asc);
    return max_features;
}

int ora_border_filter(ora_kp *kps, int n, int w, int h, int border)
{
    if (w <= border * 2 || h <= border * 2) return 0;
    int m = 0;
    for (int i = 0; i < n; i++) {
        float x = kps[i].x, y = kps[i].y;
        if (x >= (float)border && x < (float)(w - border) && y >= (float)border &&
            y < (float)(h - border))
            kps[m++] = kps[i];
    }
    return m;
}

/* ------------------------------------------------------------------ blur */

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* [OCV-recall] GaussianBlur(7x7, sigma 2) on CV_8U through the integer
 * sepFilter2D path: float kernel from getGaussianKernel scaled by 256 and
 * rounded (sum 257, not renormalised); rows accumulate in int32; columns add
 * 2^15 and shift by 16 with saturation. */
void ora_blur7(const uint8_t *src, int w, int h, int stride, uint8_t *dst,
               int dstride)
{
    /* row sums fit 16 bits (255 * 257); each row is first copied into a buffer with its reflected border so
     * that both passes are plain loops the compiler vectorises -- the arithmetic is unchanged */
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    uint8_t *pad = (uint8_t *)malloc((size_t)w + 6);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * stride;
        for (int k = 0; k < 3; k++) {
            pad[k] = s[reflect101(k - 3, w)];
            pad[w + 3 + k] = s[reflect101(w + k, w)];
        }
        memcpy(pad + 3, s, (size_t)w);
        uint16_t *t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++)
            t[x] = (uint16_t)(18 * (pad[x] + pad[x + 6]) + 34 * (pad[x + 1] + pad[x + 5]) +
                              49 * (pad[x + 2] + pad[x + 4]) + 55 * pad[x + 3]);
    }
    for (int y = 0; y < h; y++) {
        uint8_t *d = dst + (size_t)y * dstride;
        const uint16_t *r[7];
        for (int k = 0; k < 7; k++) r[k] = tmp + (size_t)reflect101(y + k - 3, h) * w;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 18u * ((uint32_t)r[0][x] + r[6][x]) + 34u * ((uint32_t)r[1][x] + r[5][x]) +
                           49u * ((uint32_t)r[2][x] + r[4][x]) + 55u * (uint32_t)r[3][x];
            uint32_t v = (acc + (1u << 15)) >> 16;
            d[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(pad);
    free(tmp);
}

/* ----------------------------------------------------------------- BRIEF */

/* [OCV-recall] computeOrbDescriptors, WTA_K = 2 */
void ora_brief(const uint8_t *blur, int w, int h, int stride,
               const ora_kp *kps, int n, uint8_t *desc)
{
    (void)w; (void)h;
    for (int i = 0; i < n; i++) {
        float angle = kps[i].angle;
        angle *= (float)(3.14159265358979323846 / 180.f);
        float a = (float)cos(angle), b = (float)sin(angle);
        const uint8_t *center = blur + (size_t)ora_cv_round(kps[i].y) * stride +
                                ora_cv_round(kps[i].x);
        uint8_t *d = desc + (size_t)i * 32;
        for (int byte = 0; byte < 32; byte++) {
            int val = 0;
            for (int bit = 0; bit < 8; bit++) {
                const signed char *p = vo_orb_bit_pattern_31 + (byte * 8 + bit) * 4;
                float x0 = p[0] * a - p[1] * b, y0 = p[0] * b + p[1] * a;
                float x1 = p[2] * a - p[3] * b, y1 = p[2] * b + p[3] * a;
                int t0 = center[ora_cv_round(y0) * stride + ora_cv_round(x0)];
                int t1 = center[ora_cv_round(y1) * stride + ora_cv_round(x1)];
                val |= (t0 < t1) << bit;
            }
            d[byte] = (uint8_t)val;
        }
    }
}

/* reference src/feature_detector.cpp:37-88 on a 1-channel input */
int ora_detect(const uint8_t *gray, int w, int h, int stride, int max_features,
               int fast_threshold, ora_kp *kps, uint8_t *desc)
{
    int cap = (w * h) / 4 + 16;
    ora_kp *all = (ora_kp *)malloc(sizeof(ora_kp) * (size_t)cap);
    int n = ora_fast(gray, w, h, stride, fast_threshold, all, cap);
    if (n > cap) n = cap;
    n = ora_select_top(all, n, w, max_features);
    if (n > 0) {
        /* ORB::compute: border filter, blur level 0, sample the pattern */
        n = ora_border_filter(all, n, w, h, 31);
        if (n > 0) {
            uint8_t *blur = (uint8_t *)malloc((size_t)w * h);
            ora_blur7(gray, w, h, stride, blur, w);
            ora_brief(blur, w, h, w, all, n, desc);
            free(blur);
        }
    }
    memcpy(kps, all, sizeof(ora_kp) * (size_t)n);
    free(all);
    return n;
}

/* --------------------------------------------------------------- matcher */

/* [OCV-recall] batchDistance with K = 2: a candidate enters when its distance
 * is strictly smaller than the current K-th best, so equal distances keep the
 * lower train index first. */
void ora_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt,
                      ora_knn2 *out)
{
    for (int i = 0; i < nq; i++) {
        const uint64_t *a = (const uint64_t *)(q + (size_t)i * 32);
        uint64_t a0, a1, a2, a3;
        memcpy(&a0, a, 8); memcpy(&a1, a + 1, 8); memcpy(&a2, a + 2, 8); memcpy(&a3, a + 3, 8);
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; j++) {
            uint64_t b0, b1, b2, b3;
            const uint8_t *bp = t + (size_t)j * 32;
            memcpy(&b0, bp, 8); memcpy(&b1, bp + 8, 8); memcpy(&b2, bp + 16, 8); memcpy(&b3, bp + 24, 8);
            int d = __builtin_popcountll(a0 ^ b0) + __builtin_popcountll(a1 ^ b1) +
                    __builtin_popcountll(a2 ^ b2) + __builtin_popcountll(a3 ^ b3);
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        out[i].idx0 = i0; out[i].dist0 = i0 >= 0 ? d0 : 0;
        out[i].idx1 = i1; out[i].dist1 = i1 >= 0 ? d1 : 0;
    }
}

int ora_ratio_test(const ora_knn2 *knn, int nq, int nt, float ratio,
                   const ora_kp *qk, const ora_kp *tk, ora_match *out)
{
    int m = 0;
    if (nt < 2) return 0; /* knn_match.size() < 2 -> skipped */
    for (int i = 0; i < nq; i++) {
        float d0 = (float)knn[i].dist0, d1 = (float)knn[i].dist1;
        if (d0 < ratio * d1) {
            out[m].query = i;
            out[m].train = knn[i].idx0;
            out[m].distance = d0;
            out[m].px = qk[i].x; out[m].py = qk[i].y;
            out[m].cx = tk[knn[i].idx0].x; out[m].cy = tk[knn[i].idx0].y;
            m++;
        }
    }
    return m;
}

int ora_cross_check(const ora_knn2 *fwd, int nq, const ora_knn2 *bwd, int nt,
                    uint8_t *keep)
{
    int m = 0;
    for (int i = 0; i < nq; i++) {
        int j = fwd[i].idx0;
        keep[i] = (uint8_t)(j >= 0 && j < nt && bwd[j].idx0 == i);
        m += keep[i];
    }
    return m;
}

/* --------------------------------------------------- frame-processor glue */

int ora_backproject(ora_match *m, int n, const float *depth, int w, int h,
                    int dstride, double fx, double fy, double cx, double cy,
                    float *pts3)
{
    const float min_depth = 50.0f, max_depth = 20000.0f;
    ora_match *tmp = (ora_match *)malloc(sizeof(ora_match) * (size_t)(n > 0 ? n : 1));
    int k = 0;
    for (int i = 0; i < n; i++) {
        float ptx = m[i].cx, pty = m[i].cy;
        int u = (int)roundf(ptx), v = (int)roundf(pty);
        if (u < 0 || u >= w || v < 0 || v >= h) continue;
        float z = depth[(size_t)v * dstride + u];
        if (isnan(z) || isinf(z) || z <= 0.0f || z < min_depth || z > max_depth) continue;
        float X = (float)((ptx - cx) * z / fx);
        float Y = (float)((pty - cy) * z / fy);
        tmp[k] = m[i];
        pts3[3 * k + 0] = X; pts3[3 * k + 1] = Y; pts3[3 * k + 2] = z;
        k++;
    }
    if (k > 0) memcpy(m, tmp, sizeof(ora_match) * (size_t)k);
    free(tmp);
    return k;
}

int ora_track_filter(ora_match *m, int n, const int32_t *prev_age, int n_prev,
                     int32_t *curr_age, int n_curr)
{
    const int min_track_length = 3;
    for (int i = 0; i < n_curr; i++) curr_age[i] = 0;
    for (int i = 0; i < n; i++) {
        int pa = (m[i].query >= 0 && m[i].query < n_prev) ? prev_age[m[i].query] : 0;
        if (m[i].train >= 0 && m[i].train < n_curr) curr_age[m[i].train] = pa + 1;
    }
    if (n > 80) {
        int mature = 0;
        for (int i = 0; i < n; i++) {
            int t = m[i].train;
            if (t >= 0 && t < n_curr && curr_age[t] >= min_track_length) mature++;
        }
        if (mature >= 50) {
            int k = 0;
            for (int i = 0; i < n; i++) {
                int t = m[i].train;
                if (t >= 0 && t < n_curr && curr_age[t] >= min_track_length) m[k++] = m[i];
            }
            return k;
        }
    }
    return n;
}

/* ------------------------------------------------------ ImageProcessor */

/* [OCV-recall] cv::equalizeHist (imgproc/histogram.cpp): 256-bin histogram,
 * first non-empty bin i0 maps to 0, scale = 255.f / (total - hist[i0]),
 * lut[i] = saturate_cast<uchar>(cumsum * scale) (float product, round half
 * to even); a constant image is filled with its value.
 * reference src/image_processor.cpp:46 */
void ora_equalize_hist(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride)
{
    int hist[256] = {0};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) hist[src[(size_t)y * stride + x]]++;
    const int total = w * h;
    int i = 0;
    while (!hist[i]) ++i;
    uint8_t lut[256];
    if (hist[i] == total) {
        for (int k = 0; k < 256; k++) lut[k] = (uint8_t)i;
    } else {
        const float scale = (256 - 1.f) / (float)(total - hist[i]);
        int sum = 0;
        for (int k = 0; k <= i; k++) lut[k] = 0;
        for (++i; i < 256; ++i) {
            sum += hist[i];
            long v = lrintf((float)sum * scale);
            lut[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[(size_t)y * dstride + x] = lut[src[(size_t)y * stride + x]];
}

/* [OCV-recall] 8.8 fixed-point Gaussian taps as cv::GaussianBlur builds them
 * for 8-bit images (getGaussianKernelBitExact + error-diffused rounding in
 * imgproc/smooth.dispatch.cpp): sigma <= 0 selects 0.3*((n-1)*0.5-1)+0.8 and,
 * for n <= 7, the small fixed tables; the side taps are rounded to 1/256 with
 * the rounding error carried to the next tap, the centre takes what is left
 * of 256.  OpenCV evaluates exp() in softfloat; libm's exp can differ from it
 * only in the last ulp, which moves a tap only at an exact .5 boundary. */
int ora_gaussian_taps_q8(int n, double sigma, int *taps)
{
    static const double small_tab[4][7] = {
        {1.0},
        {0.25, 0.5, 0.25},
        {0.0625, 0.25, 0.375, 0.25, 0.0625},
        {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125}};
    if (n < 1 || n > 31 || n % 2 == 0) return 0;
    double k[31];
    if (sigma <= 0 && n <= 7) {
        for (int i = 0; i < n; i++) k[i] = small_tab[n >> 1][i];
    } else {
        const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
        const double scale2x = -0.5 / (sx * sx);
        const int n2 = (n - 1) / 2;
        double sum = 0;
        for (int i = 0; i < n2; i++) {
            const double x = i - n2;
            k[i] = exp(scale2x * x * x);
            sum += k[i];
        }
        sum *= 2;
        sum += 1.0; /* centre tap, exp(0) */
        const double inv = 1.0 / sum;
        for (int i = 0; i < n2; i++) k[i] = k[n - 1 - i] = k[i] * inv;
        k[n2] = inv;
    }
    const int n2 = n / 2;
    double err = 0;
    int sum = 0;
    for (int i = 0; i < n2; i++) {
        const double adj = k[i] * 256.0 + err;
        const int v = (int)lrint(adj);
        err = adj - (double)v;
        taps[i] = taps[n - 1 - i] = v;
        sum += v;
    }
    taps[n2] = 256 - 2 * sum;
    return 1;
}

/* cv::GaussianBlur(src, dst, Size(n,n), sigma) on a whole 8-bit image: the
 * fixed-point path -- rows in 8.8, columns in 16.16, (sum + 2^15) >> 16,
 * BORDER_REFLECT_101.  reference src/image_processor.cpp:57-59 */
int ora_gaussian_blur(const uint8_t *src, int w, int h, int stride, int n, double sigma, uint8_t *dst,
                      int dstride)
{
    int taps[31];
    if (!ora_gaussian_taps_q8(n, sigma, taps)) return 0;
    const int r = n / 2;
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = -r; i <= r; i++) s += (uint32_t)taps[i + r] * src[(size_t)y * stride + reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = (uint16_t)(s > 65535u ? 65535u : s);
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t s = 0;
            for (int i = -r; i <= r; i++) s += (uint32_t)taps[i + r] * tmp[(size_t)reflect101(y + i, h) * w + x];
            s = (s + 32768u) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s > 255u ? 255u : s);
        }
    free(tmp);
    return 1;
}

/* the centre-band mask of ImageProcessor::createMask: a filled cv::rectangle
 * from (0, h/4) to (w, 3h/4), corners inclusive, ANDed with the image.
 * reference src/image_processor.cpp:65-79 */
void ora_roi_mask(const uint8_t *src, int w, int h, int stride, uint8_t *dst, int dstride)
{
    const int y0 = h / 4, y1 = h * 3 / 4;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[(size_t)y * dstride + x] = (y >= y0 && y <= y1) ? src[(size_t)y * stride + x] : 0;
}

/* ---- undistort -------------------------------------------------------------- */

/* [OCV-recall] cv::undistort(src, dst, K, D) on an 8-bit single-channel image:
 * initUndistortRectifyMap(K, D, I, K, CV_16SC2) followed by
 * remap(INTER_LINEAR, BORDER_CONSTANT, 0).
 *   map: for destination pixel (j, i): x = (j - cx)/fx, y = (i - cy)/fy,
 *        r2 = x^2 + y^2, kr = (1 + ((k3 r2 + k2) r2 + k1) r2) / (1 + ((k6 r2 + k5) r2 + k4) r2),
 *        xd = x kr + p1 2xy + p2 (r2 + 2x^2), yd = y kr + p1 (r2 + 2y^2) + p2 2xy,
 *        u = fx xd + cx, v = fy yd + cy, all in double;
 *        iu = cvRound(u * 32), iv = cvRound(v * 32): integer parts (>> 5, kept
 *        as int16) and 5-bit fractions.
 *   remap: weights (32-a)(32-b), a(32-b), (32-a)b, ab times 32 (sum 2^15; the
 *        int16 table clamps the a = b = 0 weight to 32767, which changes no
 *        8-bit result), taps outside the image read 0,
 *        result (sum + 2^14) >> 15.
 * dist holds k1 k2 p1 p2 [k3 [k4 k5 k6]] (n_dist = 4, 5 or 8).
 * OpenCV evaluates x by repeated addition of 1/fx along a row, processes the
 * image in stripes with a shifted principal point and inverts K numerically;
 * those differ from the closed forms used here by rounding only, which moves a
 * result only where u * 32 or v * 32 falls on a rounding boundary. */
int ora_undistort(const uint8_t *src, int w, int h, int stride, double fx, double fy, double cx, double cy,
                  const double *dist, int n_dist, uint8_t *dst, int dstride)
{
    if (n_dist != 4 && n_dist != 5 && n_dist != 8) return 0;
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const double k3 = n_dist >= 5 ? dist[4] : 0.0;
    const double k4 = n_dist >= 8 ? dist[5] : 0.0, k5 = n_dist >= 8 ? dist[6] : 0.0, k6 = n_dist >= 8 ? dist[7] : 0.0;
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    for (int i = 0; i < h; i++) {
        const double y = (i - cy) * ify;
        for (int j = 0; j < w; j++) {
            const double x = (j - cx) * ifx;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = x * kr + p1 * xy2 + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * xy2;
            double u = (fx * xd + cx) * 32, v = (fy * yd + cy) * 32;
            /* saturate_cast<int>, then the int16 store of the integer part */
            u = u < -2147483648.0 ? -2147483648.0 : (u > 2147483647.0 ? 2147483647.0 : u);
            v = v < -2147483648.0 ? -2147483648.0 : (v > 2147483647.0 ? 2147483647.0 : v);
            const int iu = isnan(u) ? 0 : (int)lrint(u), iv = isnan(v) ? 0 : (int)lrint(v);
            const int sx = (int16_t)(iu >> 5), sy = (int16_t)(iv >> 5), a = iu & 31, b = iv & 31;
            int acc = 0;
            for (int t = 0; t < 4; t++) {
                const int xx = sx + (t & 1), yy = sy + (t >> 1);
                const int wt = ((t & 1) ? a : 32 - a) * ((t >> 1) ? b : 32 - b) * 32;
                if (xx >= 0 && xx < w && yy >= 0 && yy < h) acc += src[(size_t)yy * stride + xx] * wt;
            }
            dst[(size_t)i * dstride + j] = (uint8_t)((acc + (1 << 14)) >> 15);
        }
    }
    return 1;
}
