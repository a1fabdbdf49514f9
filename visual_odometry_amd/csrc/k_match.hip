// Matching kernels for gfx950: brute-force Hamming kNN-2 on the matrix cores
// (descriptor bits as int8 vectors, v_mfma_i32_32x32x32_i8), merge + Lowe ratio
// test with ordered compaction.  Integer work only; results are bit-exact.
#include "vo_internal.h"

namespace vo {

constexpr int KNN_THREADS = 256;                 // four waves
constexpr int KNN_QPW = 64;                      // queries per wave: two sets of 32 MFMA columns
constexpr int KNN_QPB = (KNN_THREADS / 64) * KNN_QPW;  // queries per block
constexpr int KNN_TILE = 32;                     // train descriptors per step (MFMA rows)
constexpr int KNN_ROW = 256 + 16;                // bytes per unpacked train row in LDS: 16 consecutive rows of a
                                                 // 16-byte read fall into distinct banks

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// median of three unsigned values (v_med3_u32; the compiler has no builtin for the integer form)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// the same for signed values (v_med3_i32)
__device__ __forceinline__ int imed3(int a, int b, int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Key scale: a key is kKeyScale * distance + train index, so the scale must exceed every train index
// (vo_cfg.max_features <= 6144) -- and it is twice the magnitude of the byte product of the MFMA operands,
// 64 * 48 = 6144 / 2.
constexpr int kKeyScale = 6144;
constexpr int kKeyBias = 256 * (kKeyScale / 2);  // 256 equal bits: the products sum to minus this
static_assert(kKeyScale == 2 * 64 * 48, "train bytes +-64, query bytes -+48");
static_assert(kKeyScale >= VO_MAX_FEATURES_LIMIT, "the key scale must exceed every train index");
constexpr uint32_t kKeyNone = 0x7e000000u;       // keys at or above: no train

// bits 4i .. 4i+3 of x spread to bit `pos` of four bytes (bit 4i+j in byte j): the multiplier puts bit j at 8j
template <int pos>
__device__ __forceinline__ uint32_t nibble_spread(uint32_t x, int i)
{
    const int sh = 4 * i - pos;  // the nibble ends up at bits pos .. pos+3 before the multiplication
    const uint32_t n = (sh >= 0 ? x >> (sh & 31) : x << ((-sh) & 31)) & (15u << pos);
    return __umul24(n, 0x204081u) & (0x01010101u << pos);  // (15 << 7) * 0x204081 < 2^32
}

// Brute-force Hamming kNN-2 on the matrix cores.  With descriptor bits as
// +-1 the inner product of two descriptors is 256 - 2 d (d = Hamming
// distance), so all pairs of a 32-train x 32-query tile come out of eight
// v_mfma_i32_32x32x32_i8 (K = 256 bits as bytes).  The byte values are chosen
// so that the accumulator IS the sort key: train bits become +64 / -64, query
// bits -48 / +48 (the products sum to 6144 d - 786432), and the accumulator
// of train t starts at 786432 + index(t): after the eight MFMAs it holds
//   6144 d + index(t)
// -- ordered by (distance, then lower train index), the order in which
// cv::BFMatcher::knnMatch reports equal distances (reference
// src/feature_matcher.cpp:27-31).  With the query on the lane (MFMA column)
// and 16 trains in a lane's accumulator registers, the running two smallest
// keys of a query cost two vector instructions per pair:
//   k1 = med3(k0, k1, key), k0 = min(k0, key);
// distance and index are taken apart once per query at the end.  A block
// holds 256 queries in registers for its whole life (the MFMA B operands) and
// streams the trains of its share 32 at a time: the packed descriptors are
// expanded to bytes in LDS (double-buffered; the next tile's packed words are
// loaded a step ahead).  Integer arithmetic throughout: bit-exact.
__global__ __launch_bounds__(KNN_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_knn2(const uint8_t *__restrict__ desc, size_t slot_stride, const int *__restrict__ nkp,
            const int *__restrict__ q_slot, const int *__restrict__ t_slot, int kp_stride,
            uint2 *__restrict__ partial, unsigned long long *stamp)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_a[3][KNN_TILE * KNN_ROW];  // tiles t, t+1, t+2
    // (optional: the device wall clock at the start of the launch, for callers that time the matcher)
    if (stamp && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *stamp = wall_clock64();
    __shared__ uint32_t s_lut[16];  // four descriptor bits -> four bytes +64 (bit set) / -64

    const int pair = blockIdx.z, split = blockIdx.y;
    // block-uniform values, kept in scalar registers so that the loops below are controlled by the scalar unit
    const int qs = __builtin_amdgcn_readfirstlane(q_slot[pair]), ts = __builtin_amdgcn_readfirstlane(t_slot[pair]);
    const int nq = __builtin_amdgcn_readfirstlane(nkp[qs]), nt = __builtin_amdgcn_readfirstlane(nkp[ts]);
    const int qbase = blockIdx.x * KNN_QPB;
    if (qbase >= nq) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, half = lane >> 5;  // MFMA column (query) / row (train) and K half of this lane
    const uint4 *qd = reinterpret_cast<const uint4 *>(desc + qs * slot_stride);
    const uint32_t *td = reinterpret_cast<const uint32_t *>(desc + ts * slot_stride);
    if (tid < 16) s_lut[tid] = nibble_spread<7>((uint32_t)tid, 0) ^ 0xc0c0c0c0u;

    // B operands: bits 32 ks + 16 half .. + 15 of the lane's two queries as bytes -48 (bit set) / +48
    v4i B[2][8];
    // Running two smallest keys of the lane's two queries, RELATIVE to the first train of the tile they were last
    // updated with (signed: an earlier train at distance 0 has a negative relative key).  Keeping the tile's
    // accumulator start values fixed and moving these four registers by one tile per step costs 4 vector
    // instructions per step instead of 16.
    int k0[2], k1[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int q = min(qbase + wave * KNN_QPW + 32 * s + col, nq - 1);
        const uint4 lo = qd[2 * q], hi = qd[2 * q + 1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            const uint32_t bits = (w[ks] >> (16 * half)) & 0xffffu;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t m = nibble_spread<5>(bits, j);          // 0x20 per set bit
                B[s][ks][j] = (int)(0x30303030u ^ ((m << 3) - m));     // 0x30 ^ 0xe0 = 0xd0 = -48
            }
        }
        k0[s] = k1[s] = 0x7fffffff;
    }

    // this block's share of the train set: an even cut of the descriptors
    const int j_lo = (int)(((long long)nt * split) / (int)gridDim.y);
    const int j_hi = (int)(((long long)nt * (split + 1)) / (int)gridDim.y);
    const int ntiles = (j_hi - j_lo + KNN_TILE - 1) / KNN_TILE;

    // staging role of this thread: word sc of train sr of the tile
    const int sr = tid >> 3, sc = tid & 7;
    int next_row = j_lo + sr;  // train row of the next packed word this thread loads
    auto load_packed = [&]() -> uint32_t {
        const uint32_t x = td[(size_t)min(next_row, nt - 1) * 8 + sc];  // rows past the share: any train
        next_row += KNN_TILE;
        return x;
    };
    // the two 16-byte halves of words 4 .. 7 are swapped, so that the eight lanes of a train write (and the
    // MFMA operand reads below fetch) distinct banks
    const int stage_at = sr * KNN_ROW + 32 * sc, stage_sw = (sc >> 2) * 16;
    auto stage_store = [&](int buf, const uint32_t (&e)[8]) {
        uint8_t *row = &s_a[buf][stage_at];
        *reinterpret_cast<uint4 *>(row + stage_sw) = make_uint4(e[0], e[1], e[2], e[3]);
        *reinterpret_cast<uint4 *>(row + (16 - stage_sw)) = make_uint4(e[4], e[5], e[6], e[7]);
    };
    // expansion of a packed word to bytes +64 (bit set) / -64: arithmetic (prologue) or through the table in LDS
    // (steady state: two vector instructions and one LDS read per four bits instead of five instructions)
    auto expand = [&](uint32_t x, uint32_t (&e)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = nibble_spread<7>(x, i) ^ 0xc0c0c0c0u;
    };
    auto expand_lut = [&](uint32_t x, uint32_t (&e)[8]) {
        const uint8_t *lut = reinterpret_cast<const uint8_t *>(s_lut);
        e[0] = *reinterpret_cast<const uint32_t *>(lut + ((x << 2) & 0x3cu));
#pragma unroll
        for (int i = 1; i < 8; i++) e[i] = *reinterpret_cast<const uint32_t *>(lut + ((x >> (4 * i - 2)) & 0x3cu));
    };

    // One step = one tile.  The two accumulation chains of a tile (query sets 0 and 1) are issued one after
    // the other; the top-2 updates of a finished chain are placed in the issue gaps of the next chain (two
    // accumulator registers = four vector instructions per MFMA), so set 1 of tile t-1 is consumed under chain 0
    // of tile t and set 0 of tile t under chain 1 of tile t.
    auto update = [&](int s, int key) {
        // k0 <= k1: the two smallest of {k0, k1, key} are min(k0, key) and the median of the three
        k1[s] = imed3(k0[s], k1[s], key);
        k0[s] = min(k0[s], key);
    };
    // the chain of query set s over the tile in A, with the updates of set 1 - s (accumulator `done`)
    // (gap(ks): further work for the issue gap after MFMA ks)
    auto chain = [&](int s, v16i &acc, const v16i &start, v4i (&A)[8], const v16i &done, auto gap) {
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[ks], B[s][ks], ks == 0 ? start : acc, 0, 0, 0);
            update(1 - s, done[2 * ks]);
            update(1 - s, done[2 * ks + 1]);
            gap(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int op_lo = col * KNN_ROW + 16 * half, op_hi = col * KNN_ROW + 16 * (half ^ 1);
    auto operand = [&](int buf, int ks) {
        return *reinterpret_cast<const v4i *>(&s_a[buf][(ks < 4 ? op_lo : op_hi) + 32 * ks]);
    };

    v16i acc0, acc1, start;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        acc0[r] = acc1[r] = 0x7fffffff;  // "tile -1": keys that change nothing
        // accumulator register r of a lane is train row (r & 3) + 8 (r >> 2) + 4 half of the tile: the key of that
        // train relative to the tile's first one
        start[r] = kKeyBias + (r & 3) + 8 * (r >> 2) + 4 * half;
    }
    // Three tiles live in LDS.  Step t expands tile t+2 (from packed words loaded during step t-1), runs the two
    // chains of tile t from registers and, as chain 1 releases them one by one, refills the operand registers
    // with tile t+1 (expanded during step t-1): the operand reads of a step are in flight during the MFMAs of
    // the step before, nothing is read right after the block barrier.
    {
        uint32_t e[8];
        if (ntiles > 0) { expand(load_packed(), e); stage_store(0, e); }
        if (ntiles > 1) { expand(load_packed(), e); stage_store(1, e); }
    }
    uint32_t next = ntiles > 2 ? load_packed() : 0u;
    __syncthreads();
    v4i A[8];
#pragma unroll
    for (int ks = 0; ks < 8; ks++) A[ks] = operand(0, ks);
    int bcur = 0;  // t % 3
    for (int t = 0; t < ntiles; t++) {
        const int bnext = bcur == 2 ? 0 : bcur + 1, bnext2 = bnext == 2 ? 0 : bnext + 1;
        const bool fill = t + 2 < ntiles;             // block-uniform
        uint32_t e[8];
        if (fill) expand_lut(next, e);                // table reads in flight under chain 0
        if (t + 3 < ntiles) next = load_packed();
        if (t == ntiles - 1) {
            // rows past the share (their bytes are some other train's): keys above every real one
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (j_lo + t * KNN_TILE + start[r] - kKeyBias >= j_hi) start[r] = (int)kKeyNone + kKeyBias;
        }
        // the running keys move on by one tile: set 0 becomes relative to tile t, whose keys it takes under chain 1;
        // set 1 becomes relative to tile t-1, whose keys it takes under chain 0
#pragma unroll
        for (int s = 0; s < 2; s++) { k0[s] -= KNN_TILE; k1[s] -= KNN_TILE; }
        __builtin_amdgcn_sched_barrier(0);
        chain(0, acc0, start, A, acc1, [](int) {});
        if (fill) stage_store(bnext2, e);
        __builtin_amdgcn_sched_barrier(0);
        // (after the last tile the refill reads a stale buffer: harmless, and no branch per operand)
        chain(1, acc1, start, A, acc0, [&](int ks) { A[ks] = operand(bnext, ks); });
        // block barrier for the LDS tiles only: __syncthreads() would also wait for the packed words of tile
        // t+3, a global load that is meant to stay in flight across the barrier
#ifdef VO_KNN_NO_BARRIER
        // timing-only diagnostic build (wrong results: the tiles race): what no wave ever waiting for another at
        // this point would be worth -- the bound on any finer-grained hand-over (per-tile ready flags)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        bcur = bnext;
    }
    // Set 0 took tile t's keys in step t and is relative to the last tile now.  Set 1 takes tile t-1's keys in
    // step t, so it is relative to the last tile but one: one more move, then the last tile's keys.
    if (ntiles > 0) {
        k0[1] -= KNN_TILE;
        k1[1] -= KNN_TILE;
#pragma unroll
        for (int r = 0; r < 16; r++) update(1, acc1[r]);
    }
    const int key_base = j_lo + (ntiles - 1) * KNN_TILE;

#pragma unroll
    for (int s = 0; s < 2; s++) {
        // the other half of the wave holds the other 16 rows of every tile for the same query
        const int o0 = __shfl_xor(k0[s], 32), o1 = __shfl_xor(k1[s], 32);
        const int n0 = min(k0[s], o0);
        const int n1 = min(max(k0[s], o0), min(k1[s], o1));
        // key -> (distance << 16) | train index
        auto finish = [&](int rel) {
            if (rel >= (int)kKeyNone) return 0xffffffffu;
            const uint32_t k = (uint32_t)(rel + key_base);
            const uint32_t d = k / (uint32_t)kKeyScale, idx = k - d * (uint32_t)kKeyScale;
            return (d << 16) | idx;
        };
        const int q = qbase + wave * KNN_QPW + 32 * s + col;
        if (half == 0 && q < nq)
            partial[((size_t)pair * gridDim.y + split) * kp_stride + q] = make_uint2(finish(n0), finish(n1));
    }
}

// ---- the same search on the vector ALU (selectable with vo_set_matcher(); the matrix-core kernel is the default)
constexpr int KNP_THREADS = 128;
constexpr int KNP_QPL = 2;                           // queries per lane
constexpr int KNP_QPB = KNP_THREADS * KNP_QPL;       // queries per block
constexpr int KNP_TILE = 128;                        // train descriptors per LDS tile

// popcount(x) + acc in one instruction (v_bcnt_u32_b32); written out because the compiler otherwise
// computes independent popcounts and adds them up with extra instructions
__device__ __forceinline__ uint32_t popc_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}


// Lane = query (two per lane, descriptors in registers), train tiles of 128
// descriptors staged in LDS and read as wave-uniform broadcasts; per pair
// eight XOR + eight chained accumulating popcounts, then the same key / min /
// median-of-three update as above.  Same keys, same results.
__global__ __launch_bounds__(KNP_THREADS) void k_knn2_popcount(const uint8_t *__restrict__ desc, size_t slot_stride,
                                                      const int *__restrict__ nkp,
                                                      const int *__restrict__ q_slot,
                                                      const int *__restrict__ t_slot, int kp_stride,
                                                      uint2 *__restrict__ partial, unsigned long long *stamp)
{
    __shared__ __attribute__((aligned(16))) uint4 s_tile[KNP_TILE * 2];
    if (stamp && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *stamp = wall_clock64();

    const int pair = blockIdx.z, split = blockIdx.y;
    // block-uniform values, kept in scalar registers so that the loops below are controlled by the scalar unit
    const int qs = __builtin_amdgcn_readfirstlane(q_slot[pair]), ts = __builtin_amdgcn_readfirstlane(t_slot[pair]);
    const int nq = __builtin_amdgcn_readfirstlane(nkp[qs]), nt = __builtin_amdgcn_readfirstlane(nkp[ts]);
    const int qbase = blockIdx.x * KNP_QPB;
    if (qbase >= nq) return;

    const uint4 *qd = reinterpret_cast<const uint4 *>(desc + qs * slot_stride);
    const uint4 *td = reinterpret_cast<const uint4 *>(desc + ts * slot_stride);

    uint4 qa[KNP_QPL], qb[KNP_QPL];
    uint32_t k0[KNP_QPL], k1[KNP_QPL];
#pragma unroll
    for (int i = 0; i < KNP_QPL; i++) {
        const int q = min(qbase + i * KNP_THREADS + (int)threadIdx.x, nq - 1);
        qa[i] = qd[2 * q];
        qb[i] = qd[2 * q + 1];
        k0[i] = k1[i] = 0xffffffffu;
    }

    // this block's share of the train set: an even cut of the descriptors
    const int j_lo = (int)(((long long)nt * split) / (int)gridDim.y);
    const int j_hi = (int)(((long long)nt * (split + 1)) / (int)gridDim.y);

    for (int j0 = j_lo; j0 < j_hi; j0 += KNP_TILE) {
        const int jn = min(KNP_TILE, j_hi - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < jn * 2; i += KNP_THREADS) s_tile[i] = td[2 * j0 + i];
        __syncthreads();
        auto one = [&](int j) {
            const uint4 ta = s_tile[2 * j], tb = s_tile[2 * j + 1];  // wave-uniform: LDS broadcast
            const uint32_t jj = (uint32_t)(j0 + j);
#pragma unroll
            for (int i = 0; i < KNP_QPL; i++) {
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
    for (int i = 0; i < KNP_QPL; i++) {
        const int q = qbase + i * KNP_THREADS + (int)threadIdx.x;
        if (q < nq) partial[((size_t)pair * gridDim.y + split) * kp_stride + q] = make_uint2(k0[i], k1[i]);
    }
}

// The train set is cut in two across blocks when the launch has pairs enough to fill the chip, in eight when it
// has not (a single pair of 1000 x 1000 is 8 blocks against 32: 12.6 -> about 6 us).
int knn_splits(int pairs, int max_n)
{
    const int query_blocks = (max_n + KNN_QPB - 1) / KNN_QPB;
    // (fewer than 64 pairs whenever the larger number is chosen: the partial buffers are sized for that)
    return pairs >= 64 || (long long)pairs * query_blocks * kKnnSplits >= 128 ? kKnnSplits : kKnnSplitsFew;
}

void launch_knn2(const uint8_t *desc, size_t slot_stride_bytes, const int *nkp, const int *q_slot,
                 const int *t_slot, int kp_stride, uint2 *partial, int max_n, int pairs, int splits, bool popcount,
                 hipStream_t s, unsigned long long *stamp)
{
    if (popcount) {
        dim3 grid((max_n + KNP_QPB - 1) / KNP_QPB, splits, pairs);
        hipLaunchKernelGGL(k_knn2_popcount, grid, dim3(KNP_THREADS), 0, s, desc, slot_stride_bytes, nkp, q_slot,
                           t_slot, kp_stride, partial, stamp);
        return;
    }
    dim3 grid((max_n + KNN_QPB - 1) / KNN_QPB, splits, pairs);
    hipLaunchKernelGGL(k_knn2, grid, dim3(KNN_THREADS), 0, s, desc, slot_stride_bytes, nkp, q_slot, t_slot,
                       kp_stride, partial, stamp);
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
                                                                int *__restrict__ n_good, int splits)
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
            for (int s = 0; s < splits; s++) {
                // every split writes its slot (0xffffffff keys when its share is empty)
                const uint2 p = partial[((size_t)pair * splits + s) * kp_stride + q];
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
                for (int s = 0; s < splits; s++)
                    b0 = min(b0, partial_rev[((size_t)pair * splits + s) * kp_stride + r.idx0].x);
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
                           vo_knn2 *knn, vo_match *good, int *n_good, int pairs, int splits, hipStream_t s)
{
    hipLaunchKernelGGL(k_match_finalize, dim3(pairs), dim3(FIN_THREADS), 0, s, partial, kps, kp_stride, nkp,
                       q_slot, t_slot, ratio, partial_rev, knn, good, n_good, splits);
}

}  // namespace vo
