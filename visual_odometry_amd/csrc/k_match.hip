// Matching kernels for gfx950: brute-force Hamming kNN-2 over LDS-staged
// descriptor tiles (XOR + v_bcnt_u32_b32), merge + Lowe ratio test with
// ordered compaction.  Integer work only; results are bit-exact.
#include "vo_internal.h"

namespace vo {

constexpr int KNN_THREADS = 128;
constexpr int KNN_QPL = 2;                           // queries per lane
constexpr int KNN_QPB = KNN_THREADS * KNN_QPL;       // queries per block
constexpr int KNN_TILE = 128;                        // train descriptors per LDS tile

// Per query the two smallest keys are tracked, k0 <= k1, with
// key = (distance << 16) | train index: the order is (distance, then lower
// train index), the order in which cv::BFMatcher::knnMatch reports equal
// distances (reference src/feature_matcher.cpp:27-31).
// popcount(x) + acc in one instruction (v_bcnt_u32_b32); written out because the compiler otherwise
// computes independent popcounts and adds them up with extra instructions
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// median of three unsigned values (v_med3_u32; the compiler has no builtin for the integer form)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__global__ __launch_bounds__(KNN_THREADS) void k_knn2(const uint8_t *__restrict__ desc, size_t slot_stride,
                                                      const int *__restrict__ nkp,
                                                      const int *__restrict__ q_slot,
                                                      const int *__restrict__ t_slot, int kp_stride,
                                                      uint2 *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) uint4 s_tile[KNN_TILE * 2];

    const int pair = blockIdx.z, split = blockIdx.y;
    // block-uniform values, kept in scalar registers so that the loops below are controlled by the scalar unit
    const int qs = __builtin_amdgcn_readfirstlane(q_slot[pair]), ts = __builtin_amdgcn_readfirstlane(t_slot[pair]);
    const int nq = __builtin_amdgcn_readfirstlane(nkp[qs]), nt = __builtin_amdgcn_readfirstlane(nkp[ts]);
    const int qbase = blockIdx.x * KNN_QPB;
    if (qbase >= nq) return;

    const uint4 *qd = reinterpret_cast<const uint4 *>(desc + qs * slot_stride);
    const uint4 *td = reinterpret_cast<const uint4 *>(desc + ts * slot_stride);

    uint4 qa[KNN_QPL], qb[KNN_QPL];
    uint32_t k0[KNN_QPL], k1[KNN_QPL];
#pragma unroll
    for (int i = 0; i < KNN_QPL; i++) {
        const int q = min(qbase + i * KNN_THREADS + (int)threadIdx.x, nq - 1);
        qa[i] = qd[2 * q];
        qb[i] = qd[2 * q + 1];
        k0[i] = k1[i] = 0xffffffffu;
    }

    // this block's share of the train set: an even cut of the descriptors
    const int j_lo = (int)(((long long)nt * split) / kKnnSplits);
    const int j_hi = (int)(((long long)nt * (split + 1)) / kKnnSplits);

    for (int j0 = j_lo; j0 < j_hi; j0 += KNN_TILE) {
        const int jn = min(KNN_TILE, j_hi - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < jn * 2; i += KNN_THREADS) s_tile[i] = td[2 * j0 + i];
        __syncthreads();
        auto one = [&](int j) {
            const uint4 ta = s_tile[2 * j], tb = s_tile[2 * j + 1];  // wave-uniform: LDS broadcast
            const uint32_t jj = (uint32_t)(j0 + j);
#pragma unroll
            for (int i = 0; i < KNN_QPL; i++) {
                // eight XOR + eight accumulating popcounts (the accumulation is chained on purpose)
                uint32_t d = popc_acc(qa[i].x ^ ta.x, 0u);
                d = popc_acc(qa[i].y ^ ta.y, d);
                d = popc_acc(qa[i].z ^ ta.z, d);
                d = popc_acc(qa[i].w ^ ta.w, d);
                d = popc_acc(qb[i].x ^ tb.x, d);
                d = popc_acc(qb[i].y ^ tb.y, d);
                d = popc_acc(qb[i].z ^ tb.z, d);
                d = popc_acc(qb[i].w ^ tb.w, d);
                const uint32_t key = (d << 16) | jj;
                // k0 <= k1: the two smallest of {k0, k1, key} are min(k0, key) and the median of the three
                k1[i] = umed3(k0[i], k1[i], key);
                k0[i] = min(k0[i], key);
            }
        };
        int j = 0;
        for (; j + 4 <= jn; j += 4) {
            one(j); one(j + 1); one(j + 2); one(j + 3);
        }
        for (; j < jn; j++) one(j);
    }

#pragma unroll
    for (int i = 0; i < KNN_QPL; i++) {
        const int q = qbase + i * KNN_THREADS + (int)threadIdx.x;
        if (q < nq) partial[((size_t)pair * kKnnSplits + split) * kp_stride + q] = make_uint2(k0[i], k1[i]);
    }
}

void launch_knn2(const uint8_t *desc, size_t slot_stride_bytes, const int *nkp, const int *q_slot,
                 const int *t_slot, int kp_stride, uint2 *partial, int max_n, int pairs, hipStream_t s)
{
    dim3 grid((max_n + KNN_QPB - 1) / KNN_QPB, kKnnSplits, pairs);
    hipLaunchKernelGGL(k_knn2, grid, dim3(KNN_THREADS), 0, s, desc, slot_stride_bytes, nkp, q_slot, t_slot,
                       kp_stride, partial);
}

// ------------------------------------------------------------- finalize

constexpr int FIN_THREADS = 1024;

// One block per frame pair: merge the split partial top-2 lists, emit the
// vo_knn2 record of every query, run the ratio test
// (reference src/feature_matcher.cpp:34-49: float32 d0 < ratio * d1) and
// append survivors in query order.  With partial_rev (kNN of the swapped
// pair) a survivor must in addition be the mutual nearest neighbour
// (cross-check, an extension: the reference builds its BFMatcher with
// crossCheck = false, src/feature_matcher.cpp:8).
__global__ __launch_bounds__(FIN_THREADS) void k_match_finalize(const uint2 *__restrict__ partial,
                                                                const vo_kp *__restrict__ kps, int kp_stride,
                                                                const int *__restrict__ nkp,
                                                                const int *__restrict__ q_slot,
                                                                const int *__restrict__ t_slot, float ratio,
                                                                const uint2 *__restrict__ partial_rev,
                                                                vo_knn2 *__restrict__ knn,
                                                                vo_match *__restrict__ good,
                                                                int *__restrict__ n_good)
{
    __shared__ int s_wave[FIN_THREADS / kWave];
    __shared__ int s_base;
    const int pair = blockIdx.x;
    const int qs = q_slot[pair], ts = t_slot[pair];
    const int nq = nkp[qs], nt = nkp[ts];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const vo_kp *qk = kps + (size_t)qs * kp_stride;
    const vo_kp *tk = kps + (size_t)ts * kp_stride;
    if (tid == 0) s_base = 0;
    __syncthreads();

    for (int q0 = 0; q0 < nq; q0 += FIN_THREADS) {
        const int q = q0 + tid;
        bool keep = false;
        vo_knn2 r = {-1, 0, -1, 0};
        if (q < nq && nt > 0) {
            uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu;
#pragma unroll
            for (int s = 0; s < kKnnSplits; s++) {
                // every split writes its slot (0xffffffff keys when its share is empty)
                const uint2 p = partial[((size_t)pair * kKnnSplits + s) * kp_stride + q];
                uint32_t key = p.x;
                uint32_t n0 = min(k0, key);
                k1 = min(k1, max(k0, key));
                k0 = n0;
                key = p.y;
                n0 = min(k0, key);
                k1 = min(k1, max(k0, key));
                k0 = n0;
            }
            if (k0 != 0xffffffffu) { r.idx0 = (int)(k0 & 0xffffu); r.dist0 = (int)(k0 >> 16); }
            if (k1 != 0xffffffffu) { r.idx1 = (int)(k1 & 0xffffu); r.dist1 = (int)(k1 >> 16); }
            if (knn) knn[(size_t)pair * kp_stride + q] = r;
            keep = r.idx1 >= 0 && (float)r.dist0 < ratio * (float)r.dist1;
            if (keep && partial_rev) {
                // cross-check: q must also be the nearest query of its best train descriptor
                uint32_t b0 = 0xffffffffu;
#pragma unroll
                for (int s = 0; s < kKnnSplits; s++)
                    b0 = min(b0, partial_rev[((size_t)pair * kKnnSplits + s) * kp_stride + r.idx0].x);
                keep = (int)(b0 & 0xffffu) == q;
            }
        } else if (q < nq && knn) {
            knn[(size_t)pair * kp_stride + q] = r;
        }
        // ordered compaction across the block
        const uint64_t m = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (keep) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            vo_match o;
            o.query = q; o.train = r.idx0; o.distance = (float)r.dist0;
            o.px = qk[q].x; o.py = qk[q].y;
            o.cx = tk[r.idx0].x; o.cy = tk[r.idx0].y;
            good[(size_t)pair * kp_stride + slot] = o;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < FIN_THREADS / kWave; w++) tot += s_wave[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) n_good[pair] = s_base;
}

void launch_match_finalize(const uint2 *partial, const vo_kp *kps, int kp_stride, const int *nkp,
                           const int *q_slot, const int *t_slot, float ratio, const uint2 *partial_rev,
                           vo_knn2 *knn, vo_match *good, int *n_good, int pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_match_finalize, dim3(pairs), dim3(FIN_THREADS), 0, s, partial, kps, kp_stride, nkp,
                       q_slot, t_slot, ratio, partial_rev, knn, good, n_good);
}

}  // namespace vo
