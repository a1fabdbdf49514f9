// Device-side state of one vo_ctx.
#pragma once

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

#include "vo_internal.h"

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // uploads that overlap the kernels on `stream` (depth planes from the host)
    hipEvent_t ev_depth = nullptr;      // the depth planes of the batch are on the device
    hipEvent_t ev_frames = nullptr;     // the frames of the batch are on the device
    vo_cfg cfg{};
    vo::GrayCoeffs gc{};
    std::string err;

    // geometry of the internal planes: rows padded to whole 128x16 tiles
    int pitch = 0, hpad = 0;
    size_t plane_bytes = 0;
    size_t in_frame_bytes = 0;
    int kp_stride = 0;  // feature records per slot
    int slots = 0;      // feature slots (keypoints + descriptors + count)
    int max_tiles = 0;  // tiles of the fused image kernel in the largest frame

    uint8_t *d_in = nullptr;     // staging for frames handed over from the host
    uint8_t *d_gray = nullptr, *d_score = nullptr, *d_blur = nullptr;
    uint8_t *d_aux = nullptr;    // fourth plane of the image-processing operator (masked output)
    uint32_t *d_hist = nullptr;  // 256 bins + 256-byte equalisation table behind them
    uint32_t *d_cand = nullptr, *d_cand_flat = nullptr, *d_cand_count = nullptr;  // [frame][tile][kTileCap], [frame][tile]
    vo_kp *d_kps = nullptr;
    uint8_t *d_desc = nullptr;
    int *d_nkp = nullptr;
    int8_t *d_pattern = nullptr;
    uint2 *d_partial = nullptr, *d_partial_rev = nullptr;
    vo_knn2 *d_knn = nullptr;
    vo_match *d_good = nullptr;
    int *d_ngood = nullptr;
    uint8_t *d_mask = nullptr;
    double *d_F = nullptr;
    int *d_info = nullptr;
    vo_match *d_final = nullptr;
    int *d_qslot = nullptr, *d_tslot = nullptr;
    uint32_t *d_rng = nullptr;
    int rng_len = 0;

    // full ORB detector (allocated on first use)
    vo::OrbPlan orb{};
    int orb_w = 0, orb_h = 0, orb_features = 0, orb_batch = 0;
    float orb_scale = 0.f;
    size_t orb_pyr_cap = 0, orb_cand_cap = 0, orb_count_cap = 0;
    int orb_lvl_cap = 0;
    uint8_t *d_pyr = nullptr, *d_pyr_blur = nullptr;
    uint32_t *d_orb_cand = nullptr, *d_orb_cand_flat = nullptr, *d_orb_cand_count = nullptr, *d_lvl_ras = nullptr;
    float *d_lvl_resp = nullptr;
    int *d_lvl_n = nullptr;
    float *d_pattern0 = nullptr;   // the unrotated 31x31 test pattern as floats, [test of a byte][byte][x0 y0 x1 y1]
    uint32_t *d_orb_tab = nullptr;  // resize tables of all levels
    size_t orb_tab_cap = 0;

    // essential-matrix operators (allocated on first use)
    double *d_eq = nullptr, *d_eE = nullptr;
    int *d_eidx = nullptr, *d_enm = nullptr, *d_ecnt = nullptr, *d_einfo = nullptr;
    uint8_t *d_emask = nullptr;
    int e_pts_cap = 0, e_hyp_cap = 0;
    int ensure_essential(int n_points, int n_hyp);

    // stereo operator (allocated on first use)
    int32_t *d_st_idx = nullptr;
    float *d_st_disp = nullptr, *d_st_depth = nullptr, *d_st_plane = nullptr;
    int *d_st_n = nullptr;
    uint8_t *d_st_index = nullptr;  // row lists of both key-point sets of every frame (launch_stereo_match)
    size_t st_rows_off = 0;
    size_t st_plane_cap = 0;
    int ensure_stereo();
    bool depth_planes_nan = false;  // d_depth holds nothing but NaN (stereo batches keep it that way)
    bool marks_on = true;           // timing events are recorded by mark()
    bool matcher_popcount = false;  // kNN-2 on the vector ALU instead of the matrix cores (vo_set_matcher)
    vo_stereo_cfg stereo_cfg{};
    bool last_batch_stereo = false;

    // pose stage
    float *d_depth = nullptr;  // staging for depth handed over from the host
    int *d_ages = nullptr;     // track ages, one row per feature slot
    int *d_has_prev = nullptr, *d_filter_flag = nullptr;
    vo_match *d_tracked = nullptr, *d_valid = nullptr;
    float *d_pts3 = nullptr;
    int *d_bpinfo = nullptr, *d_inliers = nullptr, *d_pose = nullptr;
    double *d_rt = nullptr;
    void *d_pnp2 = nullptr;    // scratch of the PnP stage's follow-up launches (pnp_phase2_bytes(max_batch))
    vo_frame_result *d_results = nullptr;

    // batched pipeline state
    vo_batch_cfg bcfg{};
    bool configured = false, submitted = false;
    bool streams_started = false;     // carry slots hold a previous frame
    bool first_pairs_linked = false;  // q_slot/has_prev already point at the carry slots
    int op_slot0 = 0, op_slot1 = 0;   // feature slots used by the single-frame operators
    int empty_slot = 0;            // feature slot that always holds 0 keypoints

    // packed output and graph replay of the batched pipeline (latency path)
    static constexpr size_t kPackMaxBytes = 32u << 20;
    bool packed = false;
    bool packed_valid = false;            // h_pack holds the last completed batch
    vo::PackLayout pack{};
    uint8_t *d_pack = nullptr;
    void *h_pack = nullptr;
    size_t pack_cap_bytes = 0;            // bytes allocated behind d_pack / h_pack
    unsigned long long *d_stamps = nullptr;
    unsigned long long *pending_stamp = nullptr;  // stored by the first block of the next detector / matcher launch
    double wall_clock_khz = 0;
    bool use_graph = false;
    hipGraphExec_t graph_exec = nullptr;       // detection, matching, track ages
    hipGraphExec_t graph_exec_back = nullptr;  // back-projection, pose, outputs (waits for the depth upload)
    // what the captured graph was built for.  `generation` counts every event that invalidates a recorded chain
    // without showing in the other fields: a new batch configuration or camera, a re-planned pyramid, reallocated
    // buffers.  Compared field by field (struct padding is never looked at).
    unsigned long long generation = 0;
    bool last_submit_graph = false;  // the last batch ran as a graph replay: no timing events were recorded
    struct GraphKey {
        const void *frames = nullptr, *depth = nullptr, *right = nullptr;
        int row_stride = 0, depth_row_stride = 0;
        size_t frame_stride = 0, depth_frame_stride = 0;
        unsigned long long generation = 0;
        double st_baseline = 0;
        float st_band = 0, st_min_disparity = 0, st_max_disparity = 0, st_ratio = 0;
        bool packed = false, popcount = false, depth_u16 = false;
        bool operator==(const GraphKey &o) const
        {
            return frames == o.frames && depth == o.depth && right == o.right && row_stride == o.row_stride &&
                   depth_row_stride == o.depth_row_stride && frame_stride == o.frame_stride &&
                   depth_frame_stride == o.depth_frame_stride && generation == o.generation &&
                   st_baseline == o.st_baseline && st_band == o.st_band && st_min_disparity == o.st_min_disparity &&
                   st_max_disparity == o.st_max_disparity && st_ratio == o.st_ratio && packed == o.packed &&
                   popcount == o.popcount && depth_u16 == o.depth_u16;
        }
    } graph_key{};
    // forget the recorded chain; every caller that frees, reallocates or re-plans something the chain refers to
    // goes through here
    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph_exec_back) (void)hipGraphExecDestroy(graph_exec_back);
        graph_exec = graph_exec_back = nullptr;
        generation++;
    }

    // optional per-stage timing of the batched pipeline (HIP events on `stream`)
    static constexpr int kStages = 12;
    bool timing = false;
    bool capturing = false;  // the stream is recording a graph: no event records
    hipEvent_t ev[kStages + 1] = {};
    void mark(int i)
    {
        if (timing && marks_on && !capturing) (void)hipEventRecord(ev[i], stream);
    }

    // pinned staging for frames and depth planes handed over in ordinary host memory (small batches only)
    static constexpr size_t kStageMaxBytes = 64u << 20;
    void *h_stage = nullptr;
    size_t h_stage_bytes = 0;

    // Helper threads of the single-stream path.  A frame and a depth plane that arrive in ordinary host memory are
    // packed into the pinned staging area before they are sent; one core packs at 15-20 GB/s, a third of what the
    // link takes, so the pieces of a plane set are dealt out among the calling thread and up to three helpers
    // (started on first use, asleep on the condition variable between frames).  The helpers only move bytes; the
    // calling thread makes every HIP call: it queues the copy of a chunk (eight pieces) as soon as all pieces of
    // the chunk are packed -- a copy call costs 10-20 us of host time, so they are few and large.  A job is one
    // plane set; the jobs of a call are worked through in order (the frame first, then the depth plane, which
    // nothing reads before the back-projection).
    struct StageJob {
        static constexpr size_t kPiece = 256u << 10;
        static constexpr int kChunkPieces = 8;
        static constexpr size_t kChunk = kPiece * kChunkPieces;
        const uint8_t *src = nullptr;
        uint8_t *stage = nullptr, *dst = nullptr;
        size_t row_bytes = 0, row_stride = 0, plane_stride = 0, total = 0;
        int rows = 0, count = 0, n_pieces = 0, n_chunks = 0;
        hipStream_t stream = nullptr;
        std::atomic<int> next{0};                                   // next piece to take
        std::atomic<int> packed[kStageMaxBytes / kChunk + 1] = {};  // packed pieces per chunk
        int sent = 0;                                               // chunks whose copy is queued (calling thread only)
        int err = 0;                                                // first hipError_t (calling thread only)
    };
    struct StagePool {
        static constexpr int kMaxHelpers = 3;
        std::thread th[kMaxHelpers];
        int n_threads = 0;
        std::mutex m;
        std::condition_variable cv;
        uint64_t epoch = 0;   // bumped for every posted set of jobs
        int active = 0;       // helpers working on the current epoch
        bool quit = false;
        StageJob jobs[2];
        int n_jobs = 0;
    } stager;
    void stager_loop();
    void stager_stop();
    void stager_post(int n_jobs);        // jobs[0..n_jobs) are filled in: wake the helpers
    int stager_send(StageJob &j, bool block);   // calling thread: queue the copies of the chunks that are packed (block: of all)
    int stager_work(StageJob &j, bool caller);  // pack pieces of j until none is left; caller: and queue the copies

    static constexpr size_t kPinnedBytes = 1 << 20;
    size_t pinned_bytes = 0;  // max(kPinnedBytes, max_batch result records)
    void *h_pinned = nullptr;

    vo::PlaneBatch plane(uint8_t *base) const { return vo::PlaneBatch{base, pitch, plane_bytes}; }

    int allocate();
    void release();

    // gray/BGR frames already on the device -> keypoints + descriptors in
    // feature slots first_slot .. first_slot+batch-1
    void enqueue_detect(const uint8_t *src, int w, int h, int src_stride, size_t src_frame_stride,
                        int channels, int max_features, int fast_threshold, int batch, int first_slot);
    // plan + buffers of the full ORB detector for this geometry; VO_OK or an error code
    int ensure_orb(int w, int h, float scale_factor, int nlevels, int nfeatures, int batch);
    // the same contract as enqueue_detect, with cv::ORB::detectAndCompute semantics
    void enqueue_detect_orb(const uint8_t *src, int w, int h, int src_stride, size_t src_frame_stride,
                            int channels, int fast_threshold, int batch, int first_slot);
    // kNN-2 + ratio + F-RANSAC + motion filter for `pairs` (d_qslot/d_tslot) pairs
    void enqueue_match(int pairs, float ratio, bool cross_check, vo::FRansacParams prm);
};
