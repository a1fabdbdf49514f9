/* CPU oracle, full ORB detector ("mode B"): what cv::ORB::create(nfeatures,
 * scaleFactor, nlevels)->detectAndCompute(gray, noArray(), kps, desc) computes
 * with the other parameters at their defaults (edgeThreshold 31, firstLevel 0,
 * WTA_K 2, HARRIS_SCORE, patchSize 31, fastThreshold 20).  The reference
 * constructs exactly this object (reference src/feature_detector.cpp:18-22)
 * but only ever calls compute() on it (:73), so this path is an extension of
 * the reference's behaviour.  TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
 * PARITY UNPINNED; every block below is [OCV-recall] of
 * modules/features2d/src/orb.cpp, modules/imgproc/src/resize.cpp and
 * modules/core/src/mathfuncs_core.simd.hpp.
 */
#include "vo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_round_f(float v) { return (int)lrintf(v); }

/* getScale(): float(pow(double(scaleFactor), level)) */
int ora_orb_level_sizes(int w, int h, float scale_factor, int nlevels, int *lw, int *lh, float *lscale)
{
    for (int l = 0; l < nlevels; l++) {
        const float s = (float)pow((double)scale_factor, (double)l);
        lscale[l] = s;
        lw[l] = cv_round_f((float)w / s);
        lh[l] = cv_round_f((float)h / s);
    }
    return nlevels;
}

/* computeKeyPoints(): geometric split of nfeatures over the levels */
void ora_orb_quotas(int nfeatures, float scale_factor, int nlevels, int *quota)
{
    const float factor = (float)(1.0 / scale_factor);
    float per = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        quota[l] = cv_round_f(per);
        sum += quota[l];
        per *= factor;
    }
    quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
}

/* One axis of the bit-exact bilinear resize: source offset and the 8.8
 * weight of the second tap.  scale = 1 / (dst/src), all in IEEE double
 * (OpenCV uses softdouble, which rounds identically). */
static void linear_exact_axis(int src_n, int dst_n, int *ofs, int *w1)
{
    const double inv_scale = (double)dst_n / (double)src_n;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst_n; d++) {
        const double f = scale * ((double)d + 0.5) - 0.5;
        const int i = (int)floor(f);
        if (i >= 0 && src_n > 1) {
            if (i < src_n - 1) {
                ofs[d] = i;
                w1[d] = cv_round_d((f - (double)i) * 256.0);
            } else {
                ofs[d] = src_n - 1; /* right/bottom replicate */
                w1[d] = 0;
            }
        } else {
            ofs[d] = 0; /* left/top replicate */
            w1[d] = 0;
        }
    }
}

/* cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR_EXACT) on 8-bit data: rows
 * in 8.8 fixed point, columns in 16.16, (v + 2^15) >> 16. */
void ora_resize_linear_exact(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                             int dstride)
{
    int *xo = (int *)malloc(sizeof(int) * (size_t)dw * 2), *xw = xo + dw;
    int *yo = (int *)malloc(sizeof(int) * (size_t)dh * 2), *yw = yo + dh;
    linear_exact_axis(sw, dw, xo, xw);
    linear_exact_axis(sh, dh, yo, yw);
    for (int y = 0; y < dh; y++) {
        const uint8_t *r0 = src + (size_t)yo[y] * sstride;
        const uint8_t *r1 = src + (size_t)(yo[y] + 1 < sh ? yo[y] + 1 : sh - 1) * sstride;
        const uint32_t wy1 = (uint32_t)yw[y], wy0 = 256u - wy1;
        for (int x = 0; x < dw; x++) {
            const int o = xo[x], o1 = o + 1 < sw ? o + 1 : sw - 1;
            const uint32_t wx1 = (uint32_t)xw[x], wx0 = 256u - wx1;
            const uint32_t h0 = wx0 * r0[o] + wx1 * r0[o1];
            const uint32_t h1 = wx0 * r1[o] + wx1 * r1[o1];
            dst[(size_t)y * dstride + x] = (uint8_t)((wy0 * h0 + wy1 * h1 + 32768u) >> 16);
        }
    }
    free(xo);
    free(yo);
}

/* cv::fastAtan2 (scalar path), degrees in [0, 360] */
float ora_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
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

/* HarrisResponses(): 7x7 block of Sobel-like integer gradients, k = 0.04 */
float ora_harris_response(const uint8_t *img, int stride, int x0, int y0)
{
    const int bs = 7, r = bs / 2;
    const float scale = 1.f / ((1 << 2) * bs * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    const float harris_k = 0.04f;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < bs; i++)
        for (int j = 0; j < bs; j++) {
            const uint8_t *p = img + (size_t)(y0 - r + i) * stride + (x0 - r + j);
            const int ix = (p[1] - p[-1]) * 2 + (p[-stride + 1] - p[-stride - 1]) + (p[stride + 1] - p[stride - 1]);
            const int iy = (p[stride] - p[-stride]) * 2 + (p[stride - 1] - p[-stride - 1]) +
                           (p[stride + 1] - p[-stride + 1]);
            a += ix * ix;
            b += iy * iy;
            c += ix * iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

static const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

/* the same table built the way computeKeyPoints() builds it (self-check) */
int ora_orb_umax(int *umax)
{
    const int half = 15;
    const int vmax = (int)floor(half * sqrt(2.f) / 2 + 1), vmin = (int)ceil(half * sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
    return half + 1;
}

/* ICAngles(): intensity-centroid orientation over the radius-15 disc */
float ora_ic_angle(const uint8_t *img, int stride, int x0, int y0)
{
    const uint8_t *center = img + (size_t)y0 * stride + x0;
    int m01 = 0, m10 = 0;
    for (int u = -15; u <= 15; ++u) m10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0;
        const int d = k_umax[v];
        for (int u = -d; u <= d; ++u) {
            const int plus = center[u + v * stride], minus = center[u - v * stride];
            v_sum += plus - minus;
            m10 += u * (plus + minus);
        }
        m01 += v * v_sum;
    }
    return ora_fast_atan2((float)m01, (float)m10);
}

static int cmp_float_desc(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x > y ? -1 : (x < y ? 1 : 0);
}

/* KeyPointsFilter::retainBest(): when more than n are present keep every
 * keypoint whose response is >= the n-th largest response (all ties at the
 * boundary survive).  OpenCV leaves the survivors in nth_element/partition
 * order, which is unspecified; here the incoming (raster) order is kept. */
static int retain_best(ora_kp *kps, int n_kps, int n)
{
    if (n < 0 || n_kps <= n) return n_kps;
    if (n == 0) return 0;
    float *r = (float *)malloc(sizeof(float) * (size_t)n_kps);
    for (int i = 0; i < n_kps; i++) r[i] = kps[i].response;
    qsort(r, (size_t)n_kps, sizeof(float), cmp_float_desc);
    const float cut = r[n - 1];
    free(r);
    int m = 0;
    for (int i = 0; i < n_kps; i++)
        if (kps[i].response >= cut) kps[m++] = kps[i];
    return m;
}

/* ORB::detectAndCompute.  Keypoints come out level by level, raster order
 * inside a level.  Returns the number found; at most cap are written (desc
 * may be NULL).  pyr_out (optional) receives the level images back to back,
 * blur_out likewise their 7x7 Gaussian versions (row stride = level width). */
int ora_orb_detect(const uint8_t *gray, int w, int h, int stride, int nfeatures, float scale_factor,
                   int nlevels, int fast_threshold, ora_kp *kps, uint8_t *desc, int cap, uint8_t *pyr_out,
                   uint8_t *blur_out)
{
    enum { MAXL = 16 };
    if (nlevels < 1 || nlevels > MAXL) return -1;
    int lw[MAXL], lh[MAXL], quota[MAXL];
    float lscale[MAXL];
    ora_orb_level_sizes(w, h, scale_factor, nlevels, lw, lh, lscale);
    ora_orb_quotas(nfeatures, scale_factor, nlevels, quota);

    uint8_t *lvl[MAXL], *blr[MAXL];
    for (int l = 0; l < nlevels; l++) {
        lvl[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] + 1);
        blr[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l] + 1);
        if (l == 0)
            for (int y = 0; y < h; y++) memcpy(lvl[0] + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
        else
            ora_resize_linear_exact(lvl[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lvl[l], lw[l], lh[l], lw[l]);
    }

    int total = 0;
    for (int l = 0; l < nlevels; l++) {
        const int W = lw[l], H = lh[l];
        const int fcap = (W * H) / 4 + 16;
        ora_kp *k = (ora_kp *)malloc(sizeof(ora_kp) * (size_t)fcap);
        int n = ora_fast(lvl[l], W, H, W, fast_threshold, k, fcap);
        if (n > fcap) n = fcap;
        n = ora_border_filter(k, n, W, H, 31);
        n = retain_best(k, n, 2 * quota[l]);
        for (int i = 0; i < n; i++)
            k[i].response = ora_harris_response(lvl[l], W, cv_round_f(k[i].x), cv_round_f(k[i].y));
        n = retain_best(k, n, quota[l]);
        for (int i = 0; i < n; i++) {
            k[i].octave = l;
            k[i].size = 31 * lscale[l];
            k[i].angle = ora_ic_angle(lvl[l], W, cv_round_f(k[i].x), cv_round_f(k[i].y));
            k[i].x *= lscale[l];
            k[i].y *= lscale[l];
            if (total < cap) kps[total] = k[i];
            total++;
        }
        free(k);
    }
    const int n_out = total < cap ? total : cap;

    size_t off = 0;
    for (int l = 0; l < nlevels; l++) {
        ora_blur7(lvl[l], lw[l], lh[l], lw[l], blr[l], lw[l]);
        if (pyr_out) memcpy(pyr_out + off, lvl[l], (size_t)lw[l] * lh[l]);
        if (blur_out) memcpy(blur_out + off, blr[l], (size_t)lw[l] * lh[l]);
        off += (size_t)lw[l] * lh[l];
    }
    if (desc) {
        for (int i = 0; i < n_out; i++) {
            const int l = kps[i].octave;
            const float inv = 1.f / lscale[l];
            ora_kp at = kps[i]; /* the keypoint in level coordinates, as computeOrbDescriptors addresses it */
            at.x = (float)cv_round_f(kps[i].x * inv);
            at.y = (float)cv_round_f(kps[i].y * inv);
            ora_brief(blr[l], lw[l], lh[l], lw[l], &at, 1, desc + (size_t)i * 32);
        }
    }
    for (int l = 0; l < nlevels; l++) { free(lvl[l]); free(blr[l]); }
    return total;
}
