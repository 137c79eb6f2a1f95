// Whole-shard driver: every stage of the hot path for a range of frames in a handful of launches.
//
// Replaces the loop body of find_features_in_consecutive_frames_whole_movie (utils/utills.py:834-873) and of
// find_loop_cand_by_consensus (:1118-1146) minus detection.  The reference walks frame pairs one by one in
// Python; pairs are independent given the detections, so here every stage runs once for ALL pairs of the shard:
//   1. one batched all-pairs launch: F stereo problems (left_f vs right_f) + P temporal problems (left_f0 vs left_f1)
//   2. rect filter + 4-view join (one workgroup per pair)       3. DLT triangulation (one lane per point)
//   4. RANSAC rounds: every pair's sample indices from its MT19937 stream (on the device), AP3P per draw, consensus count per draw, then the
//      adaptive stopping rule is replayed on the host over the (status, count) arrays; pairs that need more draws
//      than were evaluated speculatively go to the next, larger round
//   5. supporter mask of the winning hypothesis + LM refit.
// Host <-> device synchronisation: one per RANSAC round (normally one) and one at the end.
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstring>
#include <vector>

#include <sys/syscall.h>
#include <unistd.h>

#include "common.h"
#include "host_pool.h"
#include "kernels.h"
#include "ransac_host.h"

namespace {

constexpr long HYP_BUDGET = 1 << 20;       // hypotheses per sub-batch of a later RANSAC round (pose buffer: 100 MB)
constexpr int ROUND_QUANTUM = 64;          // later-round draw counts are multiples of this

struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <typename T>
    T *take(size_t n) {
        off = slam_align_up(off, 256);
        T *p = base ? (T *)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct Buffers {
    // device
    uint32_t *stereo_key, *temporal_key, *key2;
    void *match_ws;
    size_t match_ws_bytes;
    int32_t *setup, *prob_q, *prob_t, *pair_f0, *pair_f1, *order, *tcount, *aux;
    float *xy0, *xy1;
    double *X;
    float *packed, *scale;          // the points' single-precision records for the scoring kernels (score.hip)
    int32_t *sample0, *produced0, *cnt0, *best_idx, *best_cnt;
    double *pose0, *best_pose, *refit_pose;
    uint8_t *status0, *best_status, *refit_status;
    uint64_t *mask;
    // Everything the host reads after round 0 sits in ONE device block (r0: tcount | aux | produced0 | dev_best | gate |
    // cnt0 | status0) with a pinned mirror, and the per-pair results that do not go straight into the caller's arrays
    // in another (tail: best_cnt | refit_status): one copy each instead of five and two.  The sub-arrays are placed for
    // the chunk's own pair count (bind_blocks), so a copy moves exactly the chunk's bytes.
    int32_t *r0, *tail, *h_r0, *h_tail;
    int32_t *dev_best, *gate, *h_dev_best, *h_gate;
    // later-round sub-batches alternate between two slots, so that the host prepares / replays one while the GPU
    // evaluates the other
    int32_t *seg_of[2], *beginR[2], *sampleR[2], *cntR[2], *best_idxR[2];
    uint32_t *seed;
    double *poseR[2];
    uint8_t *statusR[2];
    // host (pinned)
    int32_t *h_prob, *h_tcount, *h_aux, *h_produced0, *h_cnt0, *h_best_idx, *h_best_cnt;
    int32_t *h_seg_of[2], *h_beginR[2], *h_cntR[2], *h_best_idxR[2];
    uint8_t *h_status0, *h_statusR[2], *h_refit_status;
    int D0, RB;
    size_t setup_len;
    // temporal keys of a chunk with F frames start right after its F stereo key rows
    uint32_t *temporal_key_at(int F, int N) const { return stereo_key + (size_t)F * N; }
};

int first_chunk(const SlamTrackConfig *cfg) { return cfg && cfg->first_chunk > 0 ? cfg->first_chunk : 64; }

size_t r0_words(size_t P, int D0) { return P * (size_t)(5 + D0) + 4 + (P * (size_t)D0 + 3) / 4; }
size_t tail_words(size_t P) { return P + (P + 3) / 4; }

struct R0View { int32_t *tcount, *aux, *produced0, *dev_best, *gate, *cnt0; uint8_t *status0; };
R0View r0_view(int32_t *base, size_t P, int D0) {
    R0View v{};
    if (!base) return v;
    v.tcount = base;
    v.aux = base + P;
    v.produced0 = base + 3 * P;
    v.dev_best = base + 4 * P;
    v.gate = base + 5 * P;                    // 4 words, the first is the gate
    v.cnt0 = base + 5 * P + 4;
    v.status0 = (uint8_t *)(v.cnt0 + P * (size_t)D0);
    return v;
}

// places the packed blocks' sub-arrays for a chunk of P pairs (P <= the capacity the blocks were carved for)
void bind_blocks(Buffers &b, size_t P) {
    const R0View d = r0_view(b.r0, P, b.D0), h = r0_view(b.h_r0, P, b.D0);
    b.tcount = d.tcount, b.aux = d.aux, b.produced0 = d.produced0, b.dev_best = d.dev_best, b.gate = d.gate;
    b.cnt0 = d.cnt0, b.status0 = d.status0;
    b.h_tcount = h.tcount, b.h_aux = h.aux, b.h_produced0 = h.produced0, b.h_dev_best = h.dev_best, b.h_gate = h.gate;
    b.h_cnt0 = h.cnt0, b.h_status0 = h.status0;
    b.best_cnt = b.tail;
    b.refit_status = b.tail ? (uint8_t *)(b.tail + P) : nullptr;
    b.h_best_cnt = b.h_tail;
    b.h_refit_status = b.h_tail ? (uint8_t *)(b.h_tail + P) : nullptr;
}

void layout(int F, int P, int n_max, const SlamTrackConfig *cfg, Carver &d, Carver &h, Buffers &b) {
    const size_t Fs = F, Ps = P, N = n_max;
    b.D0 = first_chunk(cfg);
    b.RB = (int)(HYP_BUDGET / ROUND_QUANTUM);
    b.stereo_key = d.take<uint32_t>((Fs + Ps) * N);   // one key array: F stereo problems, then P temporal ones
    b.temporal_key = b.stereo_key ? b.stereo_key + Fs * N : nullptr;
    const int mode = cfg ? cfg->match_mode : SLAM_MATCH_CROSSCHECK;
    b.key2 = mode == SLAM_MATCH_KNN2 ? d.take<uint32_t>((Fs + Ps) * N) : nullptr;    // second nearest rows (kNN mode)
    b.match_ws_bytes = slam_match_batched_workspace_bytes(F + P, n_max, mode);       // per-row nearest keys before the rule
    b.match_ws = d.take<uint8_t>(b.match_ws_bytes);
    // per-call set-up arrays in ONE block, uploaded with one copy: problem lists, pair frames, pair seeds
    b.setup_len = 3 * (Fs + Ps) + 3 * Ps;
    b.setup = d.take<int32_t>(b.setup_len);
    b.prob_q = b.setup;
    b.prob_t = b.setup ? b.setup + (Fs + Ps) : nullptr;
    b.pair_f0 = b.setup ? b.setup + 2 * (Fs + Ps) : nullptr;
    b.pair_f1 = b.setup ? b.pair_f0 + Ps : nullptr;
    b.seed = b.setup ? (uint32_t *)(b.pair_f1 + Ps) : nullptr;
    b.order = b.setup ? b.pair_f1 + 2 * Ps : nullptr;
    b.r0 = d.take<int32_t>(r0_words(Ps, b.D0));
    b.tail = d.take<int32_t>(tail_words(Ps));
    b.xy0 = d.take<float>(Ps * N * 4);
    b.xy1 = d.take<float>(Ps * N * 4);
    b.X = d.take<double>(Ps * N * 3);
    b.packed = d.take<float>(Ps * N * 8);
    b.scale = d.take<float>(Ps);
    b.sample0 = d.take<int32_t>(Ps * b.D0 * 4);
    b.pose0 = d.take<double>(Ps * b.D0 * 12);
    b.best_idx = d.take<int32_t>(Ps);
    b.best_pose = d.take<double>(Ps * 12);
    b.best_status = d.take<uint8_t>(Ps);
    b.mask = d.take<uint64_t>(Ps * (N / 64));
    b.refit_pose = d.take<double>(Ps * 12);
    for (int k = 0; k < 2; ++k) {
        b.seg_of[k] = d.take<int32_t>(b.RB);
        b.beginR[k] = d.take<int32_t>(b.RB);
        b.sampleR[k] = d.take<int32_t>((size_t)HYP_BUDGET * 4);
        b.poseR[k] = d.take<double>((size_t)HYP_BUDGET * 12);
        b.statusR[k] = d.take<uint8_t>(HYP_BUDGET);
        b.cntR[k] = d.take<int32_t>(HYP_BUDGET);
        b.best_idxR[k] = d.take<int32_t>(2 * (size_t)b.RB);      // a sub-batch's new winners | its rows' pairs
    }

    b.h_prob = h.take<int32_t>(3 * (Fs + Ps) + 3 * Ps);   // mirrors `setup`
    b.h_r0 = h.take<int32_t>(r0_words(Ps, b.D0));
    b.h_tail = h.take<int32_t>(tail_words(Ps));
    b.h_best_idx = h.take<int32_t>(Ps);
    for (int k = 0; k < 2; ++k) {
        b.h_seg_of[k] = h.take<int32_t>(b.RB);
        b.h_beginR[k] = h.take<int32_t>(b.RB);
        b.h_statusR[k] = h.take<uint8_t>(HYP_BUDGET);
        b.h_cntR[k] = h.take<int32_t>(HYP_BUDGET);
        b.h_best_idxR[k] = h.take<int32_t>(2 * (size_t)b.RB);
    }
    bind_blocks(b, Ps);
}

// best_idx[row] >= 0: copy pose/status of that hypothesis to the segment's "best" slot
__global__ void gather_best_kernel(const double *pose, const uint8_t *status, const int *best_idx, const int *seg_of,
                                   int n_rows, int n_hyp, double *best_pose, uint8_t *best_status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = t / 12, k = t % 12;
    if (row >= n_rows) return;
    const int bi = best_idx[row];
    if (bi < 0) return;
    const int seg = seg_of ? seg_of[row] : row;
    const size_t o = (size_t)row * n_hyp + bi;
    best_pose[(size_t)seg * 12 + k] = pose[o * 12 + k];
    if (k == 0) best_status[seg] = status[o];
}

// Round 0 of the adaptive loop replayed ON THE DEVICE, one wavefront per pair, so that the winner's mask and the refit
// can be queued behind it without a host round trip.  It is a PREDICTION: the host replays the same loop from the
// downloaded (status, count) arrays (RansacReplay, the authority -- its log / pow are libm's, as numpy's) while the
// speculative kernels run, and accepts their results only if every pair's loop finished inside round 0 with the same
// winner; otherwise the call falls back to the host-driven path.  `gate` counts the pairs whose loop wants more draws:
// the speculative mask / refit kernels return at once when it is not zero, so a shard that needs later rounds
// (loop-closure validation of unrelated frames) pays two empty launches, not a wasted refit.
//
// The loop (utils/utills.py:498-523) as prefix sums over a pair's draws d = 0 .. end (lane = draw): counted iterations,
// cumulative inliers / outliers and hence the outlier ratio BEFORE draw d are exclusive scans; the loop stops at the
// first d whose "more" test fails (d = end stands for the state after the last evaluated draw); the winner is the
// first maximum of the counts among the solved draws before the stop.
__global__ __launch_bounds__(64) void replay0_kernel(const int *tcount, const int *produced, const uint8_t *status, const int *cnt,
                                                     const double *pose, int D0, double log1mp, double max_it, int fixed_it,
                                                     int *dev_best, double *best_pose, uint8_t *best_status, int *gate) {
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = tcount[p];
    const int end = n > 0 ? produced[p] : 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int it = 0;
    long long inl = 0, outl = 0, best_key = -1;
    bool finished = n <= 0;
    for (int base = 0; base <= end && !finished; base += 64) {
        const int d = base + lane;
        const bool ok = d < end && status[(size_t)p * D0 + d] != 0;
        const int c = ok ? cnt[(size_t)p * D0 + d] : 0;
        const unsigned long long okmask = __ballot(ok);
        int incl = c;                                   // inclusive scan of the counts over the lanes
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const int up = __shfl_up(incl, k);
            if (lane >= k) incl += up;
        }
        const int before = __popcll(okmask & lt);       // counted iterations of this tile before draw d
        const int it_b = it + before;
        const long long inl_b = inl + (incl - c), outl_b = outl + (long long)n * before - (incl - c);
        bool more;
        if (fixed_it > 0) {
            more = it_b < fixed_it;
        } else {
            double perc = 0.99;
            if (it_b > 0) {
                const double q = (double)outl_b / (double)(inl_b + outl_b);
                perc = q < 0.99 ? q : 0.99;
            }
            if (perc == 0.0) {
                more = false;
            } else {
                const double est = log1mp / log(1.0 - pow(1.0 - perc, 4.0));
                more = (double)it_b < est && (double)it_b < max_it;
            }
        }
        const unsigned long long stopmask = __ballot(d <= end && !more);
        int stop = base + 64;                           // first draw of the tile the loop does not consume
        if (stopmask) {
            stop = base + __ffsll((long long)stopmask) - 1;
            finished = true;
        }
        long long key = ok && d < stop ? ((long long)c << 32) | (long long)(0x7fffffff - d) : -1;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const long long o = __shfl_xor(key, m);
            key = o > key ? o : key;
        }
        if (key > best_key) best_key = key;             // equal counts: the earlier tile's draw stays (strict >, :514)
        const int total = __shfl(incl, 63), n_ok = __popcll(okmask);
        it += n_ok;
        inl += total;
        outl += (long long)n * n_ok - total;
    }
    const int best = best_key >= 0 ? 0x7fffffff - (int)(best_key & 0x7fffffff) : -1;
    if (lane == 0) {
        dev_best[p] = best;
        best_status[p] = best >= 0 ? status[(size_t)p * D0 + best] : 0;
        if (!finished) atomicAdd(gate, 1);
    }
    if (lane < 12)
        best_pose[(size_t)p * 12 + lane] = best >= 0 ? pose[((size_t)p * D0 + best) * 12 + lane]
                                                     : __longlong_as_double(-1ll);      // the 0xFF.. NaN pattern of the memset
}

template <typename F>
void parallel_for(int n, int grain, F fn) {
    parallel_for(HostPool::get(), n, grain, fn);      // host_pool.h: persistent workers, capped per rank
}

thread_local double g_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Shared {
    const SlamTrackConfig *cfg;
    const SlamTrackIO *io;
    Mat33 K;
    Mat34 M2, Pm, Qm;
    RansacParams rp;
    double thr, ratio;
    int norm, N, match_mode;
    bool outputs_pinned;             // h_pose / h_mask are page-locked: copies into them can be queued without blocking
    const uint32_t *stereo_key;      // [n_frames][n_max] stereo keys brought by the caller (slam_track_pairs_stereo_keys), or null
};

// page-locked (hipHostMalloc / hipHostRegister) host memory?  A copy into pageable memory blocks the queueing thread
// until it has been carried out, so work is only queued speculatively ahead of such a copy when this holds.
bool host_pinned(const void *p) {
    hipPointerAttribute_t a{};
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// One workspace + one stream: a chunk of the shard goes through front() -> mid() -> back().
struct Lane {
    Buffers b;
    hipStream_t st = nullptr;
    hipEvent_t ev[6];
    hipEvent_t evR[2];                               // results of a later-round sub-batch have arrived (per slot)
    hipEvent_t evK[2];                               // kernels of a later-round sub-batch have finished (per slot)
    hipStream_t cs = nullptr;                        // copy stream: a sub-batch's results travel while the next one is evaluated
    hipStream_t ds = nullptr;                        // draw stream: the next sub-batch's sample indices are generated meanwhile
    hipEvent_t evD[2];                               // a later-round sub-batch's sample indices are there (per slot)
    hipEvent_t evA = nullptr;                        // round 0 has been downloaded
    hipEvent_t ev_matched = nullptr;                 // the chunk's matcher has finished
    hipEvent_t match_after = nullptr;                // event the chunk's matcher waits for (serial matchers), or null
    bool ev_ready = false;
    bool speculative = false;                        // the chunk's mask / refit were queued behind the device-side replay
    // the chunk in flight
    int F = 0, P = 0, pair0 = 0;
    const uint8_t *d_desc = nullptr;
    const float *d_xy = nullptr;
    const int32_t *d_count = nullptr;
    const uint32_t *stereo_in = nullptr;             // the chunk's frames' stereo keys brought by the caller, or null
    std::vector<int32_t> f0, f1, order_start, order_fill;
    const uint32_t *seed = nullptr;
    std::vector<RansacReplay> rs;
    std::vector<int> active;
    double host_ms = 0, stage_ms[5] = {0, 0, 0, 0, 0};
    hipEvent_t base = nullptr;                       // common time origin of the call (caller's stream)
    std::vector<std::pair<float, float>> match_iv;   // [start, end] of every match launch, ms since `base`

    int ensure_events() {
        if (!ev_ready) {
            for (auto &e : ev) SLAM_HIP_CHECK(hipEventCreate(&e));
            for (auto &e : evR) SLAM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : evK) SLAM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            SLAM_HIP_CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            SLAM_HIP_CHECK(hipStreamCreateWithFlags(&ds, hipStreamNonBlocking));
            for (auto &e : evD) SLAM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            SLAM_HIP_CHECK(hipEventCreateWithFlags(&evA, hipEventDisableTiming));
            SLAM_HIP_CHECK(hipEventCreateWithFlags(&ev_matched, hipEventDisableTiming));
            ev_ready = true;
        }
        return SLAM_OK;
    }

    void set_chunk(const Shared &sh, int a, int e, int lo, int hi) {
        const SlamTrackIO *io = sh.io;
        pair0 = a;
        P = e - a;
        F = hi - lo;
        d_desc = io->d_desc + (size_t)2 * lo * sh.N * 64;
        d_xy = io->d_xy + (size_t)2 * lo * sh.N * 2;
        d_count = io->d_count + 2 * lo;
        stereo_in = sh.stereo_key ? sh.stereo_key + (size_t)lo * sh.N : nullptr;
        f0.resize(P);
        f1.resize(P);
        for (int p = 0; p < P; ++p) {
            f0[p] = io->h_pair_f0[a + p] - lo;
            f1[p] = io->h_pair_f1[a + p] - lo;
        }
        seed = io->h_pair_seed + a;
        bind_blocks(b, (size_t)P);
    }

    // everything up to the first host decision, asynchronously: match, join, triangulate, RANSAC round 0, D2H
    int front(const Shared &sh) {
        const int N = sh.N, D0 = b.D0;
        int rc = ensure_events();
        if (rc) return rc;
        // The set-up block is laid out for the lane's capacity, so this chunk's arrays sit at the capacity's offsets in
        // the pinned mirror too; one copy uploads the block.
        {
            int32_t *hq = b.h_prob, *ht = b.h_prob + (b.prob_t - b.setup);
            int32_t *h_f0 = b.h_prob + (b.pair_f0 - b.setup), *h_f1 = b.h_prob + (b.pair_f1 - b.setup);
            uint32_t *h_seed = (uint32_t *)(b.h_prob + ((int32_t *)b.seed - b.setup));
            for (int f = 0; f < F; ++f) {
                hq[f] = 2 * f;              // stereo problems: left image of frame f ...
                ht[f] = 2 * f + 1;          // ... against its right image
            }
            for (int p = 0; p < P; ++p) {
                hq[F + p] = 2 * f0[p];      // temporal problems: left images of the pair's two frames
                ht[F + p] = 2 * f1[p];
                h_f0[p] = f0[p];
                h_f1[p] = f1[p];
                h_seed[p] = seed[p];
            }
            // Launch order of the problems: frame by frame, a frame's stereo problem and then the temporal problems
            // whose later frame it is.  A left image is an operand of up to three problems (stereo f, temporal f-1 -> f
            // and f -> f+1); next to each other they run on one XCD at about the same time and the image comes from HBM
            // once instead of three times.  Counting sort, O(F + P).
            // With the frames' stereo keys brought by the caller (slam_track_pairs_stereo_keys) only the temporal problems
            // are launched: the order then lists them alone, numbered from 0 (the problem arrays are passed from F on).
            int32_t *h_order = b.h_prob + (b.order - b.setup);
            const int own = stereo_in ? 0 : 1;                                             // stereo problems of this call per frame
            order_start.assign(F + 1, 0);
            for (int p = 0; p < P; ++p) ++order_start[std::max(f0[p], f1[p]) + 1];
            for (int f = 0; f < F; ++f) order_start[f + 1] += order_start[f] + own;        // + the frame's stereo problem
            if (own)
                for (int f = 0; f < F; ++f) h_order[order_start[f]] = f;
            order_fill.assign(order_start.begin(), order_start.end() - 1);
            for (int p = 0; p < P; ++p) {
                const int f = std::max(f0[p], f1[p]);
                if (own) h_order[++order_fill[f]] = F + p;
                else h_order[order_fill[f]++] = p;
            }
            SLAM_HIP_CHECK(hipMemcpyAsync(b.setup, b.h_prob, sizeof(int32_t) * b.setup_len, hipMemcpyHostToDevice, st));
            SLAM_HIP_CHECK(hipMemsetAsync(b.gate, 0, 4 * sizeof(int32_t), st));
        }
        if (match_after) SLAM_HIP_CHECK(hipStreamWaitEvent(st, match_after, 0));    // serial matchers: the previous chunk's has finished
        SLAM_HIP_CHECK(hipEventRecord(ev[0], st));
        // ---- 1. matching: F stereo + P temporal problems, ONE launch over the contiguous key array ----
        //         (P temporal problems alone when the caller brought the stereo keys)
        {
            const size_t skip = stereo_in ? (size_t)F : 0;                                 // problems not launched
            rc = slam_launch_match_batched(d_desc, d_count, N, b.prob_q + skip, b.prob_t + skip, b.order, F + P - (int)skip, sh.norm,
                                           sh.match_mode, sh.ratio, b.stereo_key + skip * N, b.key2 ? b.key2 + skip * N : nullptr,
                                           b.match_ws, b.match_ws_bytes, st);
        }
        if (rc) return rc;
        SLAM_HIP_CHECK(hipEventRecord(ev[1], st));
        SLAM_HIP_CHECK(hipEventRecord(ev_matched, st));
        // ---- 2. join, 3. triangulate ----
        rc = slam_launch_rect_join(d_xy, d_count, N, stereo_in ? stereo_in : b.stereo_key, b.temporal_key_at(F, N), b.pair_f0, b.pair_f1, P,
                                   sh.cfg->rec_err, b.tcount, sh.io->d_tracked_kp + (size_t)pair0 * N * 4, b.xy0, b.xy1,
                                   b.aux, st);
        if (rc) return rc;
        SLAM_HIP_CHECK(hipEventRecord(ev[2], st));
        rc = slam_launch_triangulate(sh.Pm, sh.Qm, b.xy0, b.tcount, P, N, b.X, st);
        if (rc) return rc;
        SLAM_HIP_CHECK(hipEventRecord(ev[3], st));
        // ---- 4. RANSAC round 0: D0 speculative draws for every pair ----
        rc = slam_launch_pack_points(b.X, b.xy1, b.tcount, P, N, b.packed, b.scale, st);
        if (rc) return rc;
        rc = slam_launch_draw_samples(b.seed, b.tcount, nullptr, nullptr, P, D0, b.sample0, b.produced0, st);
        if (rc) return rc;
        rc = slam_launch_p3p(sh.K, b.X, b.xy1, 0, b.tcount, nullptr, P, N, b.sample0, D0, 0, D0, b.pose0, b.status0, st);
        if (rc) return rc;
        rc = slam_launch_score(sh.K, sh.M2, b.X, b.xy1, 0, b.tcount, nullptr, P, N, b.pose0, b.status0, D0, sh.thr, b.cnt0,
                               nullptr, st, nullptr, b.packed, b.scale);
        if (rc) return rc;
        // device-side replay of round 0 (a prediction the host verifies in mid()), then ONE download of everything the
        // host replay reads, then -- speculatively, gated on "no pair wants another round" -- the winner's mask, the
        // refit and the result downloads, so that a chunk whose pairs all finish in round 0 needs no host round trip
        speculative = sh.rp.max_draws > D0 && sh.outputs_pinned;
        if (speculative) {
            hipLaunchKernelGGL(replay0_kernel, dim3(P), dim3(64), 0, st, b.tcount, b.produced0, b.status0, b.cnt0, b.pose0, D0,
                               std::log(1.0 - sh.rp.prob), sh.rp.max_iterations, sh.rp.fixed_iterations, b.dev_best, b.best_pose,
                               b.best_status, b.gate);
            SLAM_HIP_CHECK(hipGetLastError());
        }
        SLAM_HIP_CHECK(hipMemcpyAsync(b.h_r0, b.r0, sizeof(int32_t) * r0_words(P, D0), hipMemcpyDeviceToHost, st));
        SLAM_HIP_CHECK(hipEventRecord(evA, st));
        if (speculative) {
            SLAM_HIP_CHECK(hipEventRecord(ev[4], st));
            rc = finish_chunk(sh, b.gate);
            if (rc) return rc;
        }
        return SLAM_OK;
    }

    // the winner's supporter mask (pre-refit, as the reference returns) + the refit + the downloads of the results;
    // with `gate` the two kernels do nothing unless *gate == 0 (see replay0_kernel)
    int finish_chunk(const Shared &sh, const int *gate) {
        const int N = sh.N;
        int rc = slam_launch_score(sh.K, sh.M2, b.X, b.xy1, 0, b.tcount, nullptr, P, N, b.best_pose, b.best_status, 1, sh.thr,
                                   b.best_cnt, b.mask, st, gate);
        if (rc) return rc;
        rc = slam_launch_refit(sh.K, b.X, b.xy1, 0, b.tcount, P, N, b.mask, b.best_pose, b.refit_pose, b.refit_status, st, gate);
        if (rc) return rc;
        SLAM_HIP_CHECK(hipEventRecord(ev[5], st));
        const SlamTrackIO *io = sh.io;
        SLAM_HIP_CHECK(hipMemcpyAsync(io->h_pose + (size_t)pair0 * 12, b.refit_pose, sizeof(double) * 12 * (size_t)P, hipMemcpyDeviceToHost, st));
        SLAM_HIP_CHECK(hipMemcpyAsync(io->h_mask + (size_t)pair0 * (N / 64), b.mask, sizeof(uint64_t) * (size_t)P * (N / 64), hipMemcpyDeviceToHost, st));
        SLAM_HIP_CHECK(hipMemcpyAsync(b.h_tail, b.tail, sizeof(int32_t) * tail_words(P), hipMemcpyDeviceToHost, st));
        return SLAM_OK;
    }

    // host replay of the adaptive loop, later rounds where needed, then launch mask + refit + D2H
    int mid(const Shared &sh) {
        const int N = sh.N, D0 = b.D0;
        const RansacParams &rp = sh.rp;
        SLAM_HIP_CHECK(hipEventSynchronize(evA));           // round 0 has been downloaded (the speculative tail may still run)
        double t_h0 = now_ms();
        rs.assign(P, RansacReplay());
        active.clear();
        parallel_for(P, 64, [&](int a, int e) {          // pairs are independent: replay them on all host cores
            for (int p = a; p < e; ++p) {
                RansacReplay &r = rs[p];
                const int n = b.h_tcount[p];
                if (n <= 0)
                    r.finished = true;
                else
                    r.consume(rp, n, b.h_status0 + (size_t)p * D0, b.h_cnt0 + (size_t)p * D0, 0, b.h_produced0[p]);
                b.h_best_idx[p] = r.best_draw;
            }
        });
        bool agree = speculative && b.h_gate[0] == 0;
        for (int p = 0; p < P; ++p) {
            if (!rs[p].finished) active.push_back(p);
            else if (agree && b.h_dev_best[p] != rs[p].best_draw) agree = false;
        }
        host_ms += now_ms() - t_h0;
        if (agree && active.empty()) return SLAM_OK;          // the speculative mask / refit / downloads stand
        SLAM_HIP_CHECK(hipMemsetAsync(b.best_status, 0, P, st));
        SLAM_HIP_CHECK(hipMemsetAsync(b.best_pose, 0xFF, sizeof(double) * 12 * (size_t)P, st));   // NaN pattern
        SLAM_HIP_CHECK(hipMemcpyAsync(b.best_idx, b.h_best_idx, sizeof(int32_t) * P, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_best_kernel, dim3(slam_div_up((long)P * 12, 256)), dim3(256), 0, st, b.pose0, b.status0,
                           b.best_idx, nullptr, P, D0, b.best_pose, b.best_status);
        SLAM_HIP_CHECK(hipGetLastError());

        // ---- later rounds for the pairs whose loop wanted more draws than were evaluated ----
        // The host now knows every pair's point count, so it draws the sample indices itself (same stream, restarted
        // from the pair's seed) and uploads only the new range [draws consumed so far, + n_eval).  Pairs are sorted by
        // how many draws their loop still asks for (RansacReplay::draws_hint) and cut into sub-batches of at most
        // HYP_BUDGET hypotheses; a sub-batch evaluates what its hungriest pair asks for.  A pair whose estimate grows
        // afterwards simply stays active for another round.
        // Sub-batches are software-pipelined over two buffer slots: while the GPU evaluates sub-batch k the host draws
        // the indices of k+1 and queues it, then waits for k and replays it, so host and device time overlap.
        int rc;
        struct Batch { size_t a0; int rows, n_eval; };
        std::vector<std::pair<int, int>> need;
        std::vector<Batch> batches;
        std::vector<uint8_t> fin;
        while (!active.empty()) {
            t_h0 = now_ms();
            need.resize(active.size());
            parallel_for((int)active.size(), 256, [&](int a, int e) {
                for (int k = a; k < e; ++k) need[k] = {rs[active[k]].draws_hint(rp), active[k]};
            });
            std::stable_sort(need.begin(), need.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first > y.first; });
            batches.clear();
            for (size_t a0 = 0; a0 < need.size();) {
                long ne = ((long)need[a0].first + ROUND_QUANTUM - 1) / ROUND_QUANTUM * ROUND_QUANTUM;
                if (ne > HYP_BUDGET) ne = HYP_BUDGET;
                int rows = (int)std::min<size_t>(need.size() - a0, (size_t)(HYP_BUDGET / ne));
                if (rows > b.RB) rows = b.RB;
                batches.push_back({a0, rows, (int)ne});
                a0 += rows;
            }
            host_ms += now_ms() - t_h0;
            std::vector<int> still;

            // host: which pairs and which draw ranges go into staging slot s (the indices themselves are generated on
            // the device from each pair's seed: drawing them here cost more than evaluating them there)
            auto draw = [&](const Batch &B, int s) {
                const double t0 = now_ms();
                for (int r = 0; r < B.rows; ++r) {
                    const int p = need[B.a0 + r].second;
                    b.h_seg_of[s][r] = p;
                    b.h_beginR[s][r] = rs[p].draws;
                }
                host_ms += now_ms() - t0;
            };
            // device: upload, solve, score, download; evR[s] marks the results' arrival
            auto launch = [&](const Batch &B, int s) -> int {
                const int rows = B.rows, n_eval = B.n_eval;
                // The sample indices are generated on the draw stream: one wavefront per pair regenerating its MT19937
                // stream is a latency chain that fills a hundredth of the chip, and on `st` it would sit between the
                // previous sub-batch's scoring and this one's AP3P (0.1 ms of every 0.8).  Here it runs under the previous
                // sub-batch's kernels.  What it reads (seeds, point counts) was complete before round 0 was downloaded.
                SLAM_HIP_CHECK(hipMemcpyAsync(b.seg_of[s], b.h_seg_of[s], sizeof(int32_t) * rows, hipMemcpyHostToDevice, ds));
                SLAM_HIP_CHECK(hipMemcpyAsync(b.beginR[s], b.h_beginR[s], sizeof(int32_t) * rows, hipMemcpyHostToDevice, ds));
                int r2 = slam_launch_draw_samples(b.seed, b.tcount, b.seg_of[s], b.beginR[s], rows, n_eval, b.sampleR[s], nullptr, ds);
                if (r2) return r2;
                SLAM_HIP_CHECK(hipEventRecord(evD[s], ds));
                SLAM_HIP_CHECK(hipStreamWaitEvent(st, evD[s], 0));
                r2 = slam_launch_p3p(sh.K, b.X, b.xy1, 0, b.tcount, b.seg_of[s], rows, N, b.sampleR[s], n_eval, 0, n_eval,
                                         b.poseR[s], b.statusR[s], st);
                if (r2) return r2;
                r2 = slam_launch_score(sh.K, sh.M2, b.X, b.xy1, 0, b.tcount, b.seg_of[s], rows, N, b.poseR[s], b.statusR[s],
                                       n_eval, sh.thr, b.cntR[s], nullptr, st, nullptr, b.packed, b.scale);
                if (r2) return r2;
                // the results go down on the copy stream, so the kernels of the next sub-batch (queued on `st` before the
                // host waits for these) run under the copies instead of behind them: 5 bytes per hypothesis, ~0.1 ms of
                // PCIe time per 2^20-hypothesis sub-batch against ~1 ms of kernels
                SLAM_HIP_CHECK(hipEventRecord(evK[s], st));
                SLAM_HIP_CHECK(hipStreamWaitEvent(cs, evK[s], 0));
                SLAM_HIP_CHECK(hipMemcpyAsync(b.h_statusR[s], b.statusR[s], (size_t)rows * n_eval, hipMemcpyDeviceToHost, cs));
                SLAM_HIP_CHECK(hipMemcpyAsync(b.h_cntR[s], b.cntR[s], sizeof(int32_t) * (size_t)rows * n_eval, hipMemcpyDeviceToHost, cs));
                SLAM_HIP_CHECK(hipEventRecord(evR[s], cs));
                return SLAM_OK;
            };
            // host: replay the loop over the downloaded results, then queue the copy of any new winner
            auto finish = [&](const Batch &B, int s) -> int {
                const int rows = B.rows, n_eval = B.n_eval;
                SLAM_HIP_CHECK(hipEventSynchronize(evR[s]));
                const double t0 = now_ms();
                fin.assign(rows, 0);
                parallel_for(rows, 16, [&](int a, int e) {
                    for (int r = a; r < e; ++r) {
                        const int p = need[B.a0 + r].second;
                        RansacReplay &q = rs[p];
                        const int before = q.best_draw, begin = q.draws;
                        q.consume(rp, b.h_tcount[p], b.h_statusR[s] + (size_t)r * n_eval, b.h_cntR[s] + (size_t)r * n_eval, begin,
                                  begin + n_eval);
                        b.h_best_idxR[s][r] = q.best_draw != before ? q.best_draw - begin : -1;
                        b.h_best_idxR[s][rows + r] = p;
                        fin[r] = q.finished;
                    }
                });
                for (int r = 0; r < rows; ++r)
                    if (!fin[r]) still.push_back(need[B.a0 + r].second);
                host_ms += now_ms() - t0;
                // the gather takes its rows' pairs from its own upload, not from seg_of[s]: the draw stream may refill that
                // array for the slot's next sub-batch before the gather, queued behind the other slot's kernels, has run
                SLAM_HIP_CHECK(hipMemcpyAsync(b.best_idxR[s], b.h_best_idxR[s], sizeof(int32_t) * 2 * rows, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(gather_best_kernel, dim3(slam_div_up((long)rows * 12, 256)), dim3(256), 0, st, b.poseR[s],
                                   b.statusR[s], b.best_idxR[s], b.best_idxR[s] + rows, rows, n_eval, b.best_pose, b.best_status);
                SLAM_HIP_CHECK(hipGetLastError());
                return SLAM_OK;
            };

            const int nb = (int)batches.size();
            // slot reuse is safe by construction: a slot's host buffers are rewritten only after finish() of the
            // sub-batch that last used them (its uploads and results are done; its small best-index upload precedes, in
            // stream order, the launch whose completion finish() of the slot's next user waits for), and its device
            // buffers only by work the host queues after finish() has seen the slot's copies on `cs` complete (evR: the
            // sub-batch's kernels are done) -- on `st` behind that sub-batch's gather (poses, statuses, counts), on `ds`
            // at once (pair list, draw ranges, sample indices: nothing queued still reads them, the gather has its own
            // copy of the pair list).
            draw(batches[0], 0);
            rc = launch(batches[0], 0);
            if (rc) return rc;
            for (int k = 0; k < nb; ++k) {
                const int s = k & 1;
                if (k + 1 < nb) {
                    draw(batches[k + 1], s ^ 1);
                    rc = launch(batches[k + 1], s ^ 1);
                    if (rc) return rc;
                }
                rc = finish(batches[k], s);
                if (rc) return rc;
            }
            SLAM_HIP_CHECK(hipStreamSynchronize(st));      // the next round (or the caller) reuses both slots
            std::sort(still.begin(), still.end());
            active.swap(still);
        }
        SLAM_HIP_CHECK(hipEventRecord(ev[4], st));
        // ---- 5. supporter mask of the winner + refit, on the host replay's winners ----
        return finish_chunk(sh, nullptr);
    }

    // wait for the chunk's results and write its per-pair counters
    int back(const Shared &sh) {
        SLAM_HIP_CHECK(hipStreamSynchronize(st));
        const SlamTrackIO *io = sh.io;
        for (int p = 0; p < P; ++p) {
            const RansacReplay &r = rs[p];
            int32_t *s = io->h_stats + (size_t)(pair0 + p) * 8;
            const int n = b.h_tcount[p];
            int status = 0;
            if (n <= 0) status |= SLAM_PAIR_EMPTY;
            if (r.hit_cap) status |= SLAM_PAIR_DRAW_CAP;
            if (n > 0 && !b.h_refit_status[p]) status |= SLAM_PAIR_FEW;
            if (b.h_refit_status[p]) status |= SLAM_PAIR_OK;
            s[0] = n;
            s[1] = n > 0 && r.best_draw >= 0 ? b.h_best_cnt[p] : 0;
            s[2] = r.it;
            s[3] = r.draws;
            s[4] = r.best_draw;
            s[5] = status;
            s[6] = b.h_aux[2 * p];
            s[7] = b.h_aux[2 * p + 1];
            if (n <= 0)
                for (int k = 0; k < 12; ++k) io->h_pose[(size_t)(pair0 + p) * 12 + k] = std::nan("");
        }
        float ms = 0;
        for (int k = 0; k < 5; ++k) {
            SLAM_HIP_CHECK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            stage_ms[k] += ms;
        }
        float t0 = 0, t1 = 0;
        SLAM_HIP_CHECK(hipEventElapsedTime(&t0, base, ev[0]));
        SLAM_HIP_CHECK(hipEventElapsedTime(&t1, base, ev[1]));
        match_iv.emplace_back(t0, t1);
        return SLAM_OK;
    }
};

// total time during which at least one match kernel was in flight (lanes overlap, so durations must not be summed)
double union_ms(std::vector<std::pair<float, float>> iv) {
    std::sort(iv.begin(), iv.end());
    double total = 0, end = -1e30;
    for (auto &p : iv) {
        if (p.first > end) {
            total += p.second - p.first;
            end = p.second;
        } else if (p.second > end) {
            total += p.second - end;
            end = p.second;
        }
    }
    return total;
}

// Per-host-thread HIP objects of slam_track_pairs: two lanes (events + a copy stream each), two side streams, fork / base events.  They
// belong to the device that was current when they were created.  A host thread that moves to another device between
// calls gets a fresh set there (the old one is released: nothing of a finished call is still queued on it, every call
// drains its lanes before it returns), and a worker thread that exits releases its set (a caller that runs
// slam_track_pairs from short-lived threads would otherwise leak 20 events and 4 streams per thread).  The process'
// main thread is left alone at exit: its thread-locals are destroyed while the HIP runtime is shutting down, and the
// process is about to return everything anyway.
struct ThreadState {
    Lane lane[2];
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr;
    hipEvent_t base = nullptr;
    int device = -1;                      // device the streams / events above were created on

    void release() {
        if (device < 0) return;
        for (Lane &ln : lane)
            if (ln.ev_ready) {
                for (auto &e : ln.ev) (void)hipEventDestroy(e);
                for (auto &e : ln.evR) (void)hipEventDestroy(e);
                for (auto &e : ln.evK) (void)hipEventDestroy(e);
                for (auto &e : ln.evD) (void)hipEventDestroy(e);
                if (ln.cs) (void)hipStreamDestroy(ln.cs);
                if (ln.ds) (void)hipStreamDestroy(ln.ds);
                ln.cs = ln.ds = nullptr;
                (void)hipEventDestroy(ln.evA);
                (void)hipEventDestroy(ln.ev_matched);
                ln.ev_ready = false;
            }
        if (fork) {
            (void)hipEventDestroy(fork);
            fork = nullptr;
            for (auto &s : side_stream) {
                if (s) (void)hipStreamDestroy(s);
                s = nullptr;
            }
        }
        if (base) {
            (void)hipEventDestroy(base);
            base = nullptr;
        }
        device = -1;
    }
    // drops the handles without destroying them (their device is unreachable)
    void forget() {
        for (Lane &ln : lane) ln.ev_ready = false, ln.cs = ln.ds = nullptr;
        fork = base = nullptr;
        for (auto &s : side_stream) s = nullptr;
        device = -1;
    }
    ~ThreadState() {
        if ((long)syscall(SYS_gettid) == (long)getpid()) return;      // main thread at process exit
        int cur = -1;
        if (device < 0 || hipGetDevice(&cur) != hipSuccess) return;   // runtime already gone: nothing to release
        if (cur != device && hipSetDevice(device) != hipSuccess) return;
        release();                                                    // errors (hipErrorDeinitialized, ...) are ignored
        if (cur != device) (void)hipSetDevice(cur);
    }
};
thread_local ThreadState g_ts;

int bind_device() {
    int dev = -1;
    SLAM_HIP_CHECK(hipGetDevice(&dev));
    if (dev == g_ts.device) return SLAM_OK;
    if (g_ts.device >= 0) {
        // objects of the previous device are destroyed with that device current; if that device cannot be made current
        // any more its handles are forgotten, not reused: the new device gets a fresh set either way
        if (hipSetDevice(g_ts.device) == hipSuccess) g_ts.release();
        else g_ts.forget();
        SLAM_HIP_CHECK(hipSetDevice(dev));
    }
    g_ts.device = dev;
    return SLAM_OK;
}

struct Chunk { int a, e, lo, hi; };

// contiguous pair blocks; each needs the frames [lo, hi)
// Pairs per chunk.  taper (permille) = size of every chunk relative to the one before it: 1000 (or 0) = equal chunks;
// below 1000 the chunks shrink geometrically, so the tail that nothing overlaps (the last chunk's join / RANSAC /
// refit / host replay) belongs to a small chunk.  Sizes are >= 1 and sum to P.
std::vector<int> chunk_sizes(int P, int n_chunks, int taper) {
    std::vector<int> out;
    if (n_chunks <= 1 || P < 2 * n_chunks) return out;
    const double r = taper > 0 && taper < 1000 ? taper / 1000.0 : 1.0;
    double total = 0, w = 1;
    for (int k = 0; k < n_chunks; ++k, w *= r) total += w;
    int used = 0;
    w = 1;
    for (int k = 0; k < n_chunks; ++k, w *= r) {
        const int left_after = n_chunks - 1 - k;                     // every later chunk still needs one pair
        int n = k == n_chunks - 1 ? P - used : (int)(P * w / total + 0.5);
        n = std::max(1, std::min(n, P - used - left_after));
        out.push_back(n);
        used += n;
    }
    return out;
}

// pipeline_chunks <= 0: the library's own plan.  Two lanes pay when each half still fills the chip for long enough
// that one lane's latency-bound steps (small kernels, copies, the host replay) hide behind the other's matcher: from
// about a third of a 3450-frame x 2048-row shard; below that the halves' small kernels only wait for each other
// (433 frames: 2.36 ms in one lane, 2.60 in two; 3450 frames: 15.8 against 15.4).
int planned_chunks(int n_pairs, int n_max, const SlamTrackConfig *cfg) {
    if (cfg->pipeline_chunks > 0) return cfg->pipeline_chunks;
    return (double)n_pairs * n_max * n_max >= 1200.0 * 2048.0 * 2048.0 ? 2 : 1;
}

std::vector<Chunk> plan_chunks(const SlamTrackIO *io, const SlamTrackConfig *cfg) {
    std::vector<Chunk> out;
    const std::vector<int> sizes = chunk_sizes(io->n_pairs, planned_chunks(io->n_pairs, io->n_max, cfg), cfg->pipeline_taper_permille);
    int a = 0;
    for (int n : sizes) {
        const int e = a + n;
        int lo = io->n_frames, hi = 0;
        for (int p = a; p < e; ++p) {
            lo = std::min(lo, std::min(io->h_pair_f0[p], io->h_pair_f1[p]));
            hi = std::max(hi, std::max(io->h_pair_f0[p], io->h_pair_f1[p]) + 1);
        }
        out.push_back({a, e, lo, hi});
        a = e;
    }
    return out;
}

size_t lane_bytes(int F, int P, int N, const SlamTrackConfig *cfg, bool host) {
    Carver d(nullptr), h(nullptr);
    Buffers b;
    layout(F, P, N, cfg, d, h, b);
    return slam_align_up((host ? h.off : d.off) + 256, 4096);
}

// worst-case bytes: the whole shard in one lane, or two lanes of consecutive-frame chunks
size_t workspace_bytes(int F, int P, int N, const SlamTrackConfig *cfg, bool host) {
    size_t need = lane_bytes(F, P, N, cfg, host);
    if (cfg) {
        int per = 0;
        for (int n : chunk_sizes(P, planned_chunks(P, N, cfg), cfg->pipeline_taper_permille)) per = std::max(per, n);
        if (per) need = std::max(need, 2 * lane_bytes(per + 1, per, N, cfg, host));
    }
    return need;
}

}  // namespace

extern "C" size_t slam_track_device_workspace_bytes(int n_frames, int n_pairs, int n_max, const SlamTrackConfig *cfg) {
    return workspace_bytes(n_frames, n_pairs, n_max, cfg, false);
}

extern "C" size_t slam_track_host_workspace_bytes(int n_frames, int n_pairs, int n_max, const SlamTrackConfig *cfg) {
    return workspace_bytes(n_frames, n_pairs, n_max, cfg, true);
}

// Explicit teardown, for callers that want it done at a moment of their choosing instead of during process exit:
// stops and joins the host worker pool and releases the CALLING thread's streams / events while the HIP runtime is
// still up.  Nothing in this library joins threads or calls HIP from a static destructor (host_pool.h, ThreadState), so
// skipping the call is safe; making it leaves no thread and no HIP object of this library behind when the exit
// handlers of other libraries (the HIP runtime, a profiler's tool library) run.  The library stays usable afterwards
// (host steps then run on the calling thread only).
extern "C" int slam_shutdown(void) try {
    if (g_ts.device >= 0) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess) {
            if (cur == g_ts.device || hipSetDevice(g_ts.device) == hipSuccess) g_ts.release();
            if (cur != g_ts.device && cur >= 0) (void)hipSetDevice(cur);
        }
        g_ts.device = -1;
    }
    HostPool::get().shutdown();
    return SLAM_OK;
}
SLAM_CATCH_ALL("slam_shutdown")

extern "C" int slam_track_last_timing(double *h_ms8) {
    if (!h_ms8) return SLAM_E_INVALID;
    for (int i = 0; i < 8; ++i) h_ms8[i] = g_timing[i];
    return SLAM_OK;
}

namespace {
int track_pairs(const SlamTrackConfig *cfg, const SlamTrackIO *io, const uint32_t *d_stereo_key, void *stream) {
    const double t_begin = now_ms();
    hipStream_t st = (hipStream_t)stream;
    SLAM_REQUIRE(cfg && io, "slam_track_pairs: null config/io");
    const int F = io->n_frames, P = io->n_pairs, N = io->n_max;
    SLAM_REQUIRE(F > 0 && P >= 0 && N > 0 && N % 64 == 0 && N < SLAM_MAX_ROWS, "slam_track_pairs: bad sizes F=%d P=%d n_max=%d (n_max must be a multiple of 64)", F, P, N);
    SLAM_REQUIRE(F + P <= 1 << 24, "slam_track_pairs: too many problems");
    SLAM_REQUIRE(cfg->match_mode == SLAM_MATCH_CROSSCHECK || cfg->match_mode == SLAM_MATCH_KNN2 ||
                     cfg->match_mode == SLAM_MATCH_CROSSCHECK_MUTUAL,
                 "slam_track_pairs: match_mode %d unsupported", cfg->match_mode);
    if (P == 0) return SLAM_OK;
    SLAM_REQUIRE(io->d_desc && io->d_xy && io->d_count && io->h_pair_f0 && io->h_pair_f1 && io->h_pair_seed &&
                     io->d_tracked_kp && io->h_pose && io->h_stats && io->h_mask && io->d_workspace && io->h_workspace,
                 "slam_track_pairs: null pointer in io");
    SLAM_REQUIRE(P <= 65535, "slam_track_pairs: at most 65535 pairs per call (split the shard)");
    for (int p = 0; p < P; ++p)
        SLAM_REQUIRE(io->h_pair_f0[p] >= 0 && io->h_pair_f0[p] < F && io->h_pair_f1[p] >= 0 && io->h_pair_f1[p] < F,
                     "slam_track_pairs: pair %d references a frame outside [0, %d)", p, F);

    Shared sh;
    sh.cfg = cfg;
    sh.io = io;
    sh.N = N;
    for (int i = 0; i < 9; ++i) sh.K.m[i] = cfg->K[i];
    for (int i = 0; i < 12; ++i) sh.M2.m[i] = cfg->M2[i];
    // K @ M1, K @ M2 (numpy/BLAS accumulation order)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            sh.Pm.m[i * 4 + j] = std::fma(cfg->K[i * 3 + 2], cfg->M1[8 + j], std::fma(cfg->K[i * 3 + 1], cfg->M1[4 + j], cfg->K[i * 3] * cfg->M1[j]));
            sh.Qm.m[i * 4 + j] = std::fma(cfg->K[i * 3 + 2], cfg->M2[8 + j], std::fma(cfg->K[i * 3 + 1], cfg->M2[4 + j], cfg->K[i * 3] * cfg->M2[j]));
        }
    sh.rp.prob = cfg->prob;
    sh.rp.max_iterations = cfg->max_iterations;
    sh.rp.fixed_iterations = cfg->fixed_iterations;
    sh.rp.max_draws = cfg->max_draws > 0 ? cfg->max_draws
                                         : (int)std::fmax(64.0 * cfg->max_iterations, 8.0 * cfg->fixed_iterations);
    sh.thr = cfg->acc * cfg->acc;
    sh.norm = cfg->norm ? cfg->norm : SLAM_NORM_HAMMING2;
    sh.match_mode = cfg->match_mode;
    sh.ratio = cfg->ratio > 0 ? cfg->ratio : 0.6;
    sh.outputs_pinned = host_pinned(io->h_pose) && host_pinned(io->h_mask);
    sh.stereo_key = d_stereo_key;

    // ---- plan: the whole shard in one lane on the caller's stream, or chunks pipelined over two side streams ----
    std::vector<Chunk> chunks = plan_chunks(io, cfg);
    int lane_F = 0, lane_P = 0;
    for (const Chunk &c : chunks) {
        lane_F = std::max(lane_F, c.hi - c.lo);
        lane_P = std::max(lane_P, c.e - c.a);
    }
    if (!chunks.empty()) {
        const size_t dl = lane_bytes(lane_F, lane_P, N, cfg, false), hl = lane_bytes(lane_F, lane_P, N, cfg, true);
        if (2 * dl > io->d_workspace_bytes || 2 * hl > io->h_workspace_bytes) chunks.clear();   // pairs not local in time
    }
    Lane *L = g_ts.lane;
    {
        const int rc = bind_device();
        if (rc) return rc;
    }
    if (!g_ts.base) SLAM_HIP_CHECK(hipEventCreate(&g_ts.base));
    SLAM_HIP_CHECK(hipEventRecord(g_ts.base, st));
    for (int k = 0; k < 2; ++k) {
        L[k].host_ms = 0;
        L[k].base = g_ts.base;
        L[k].match_iv.clear();
        for (double &v : L[k].stage_ms) v = 0;
    }
    int n_launch = 0;
    if (chunks.empty()) {
        Carver dc(io->d_workspace), hc(io->h_workspace);
        layout(F, P, N, cfg, dc, hc, L[0].b);
        if (dc.off > io->d_workspace_bytes || hc.off > io->h_workspace_bytes) {
            slam_set_error("slam_track_pairs: workspace too small (device %zu < %zu or host %zu < %zu)",
                           io->d_workspace_bytes, dc.off, io->h_workspace_bytes, hc.off);
            return SLAM_E_WORKSPACE;
        }
        L[0].st = st;
        L[0].match_after = nullptr;
        L[0].set_chunk(sh, 0, P, 0, F);
        int rc = L[0].front(sh);
        if (!rc) rc = L[0].mid(sh);
        if (!rc) rc = L[0].back(sh);
        if (rc) {
            (void)hipStreamSynchronize(st);                      // no work of a failed call outlives it
            if (L[0].cs) (void)hipStreamSynchronize(L[0].cs);
            if (L[0].ds) (void)hipStreamSynchronize(L[0].ds);
            return rc;
        }
        n_launch = 1;
    } else {
        if (!g_ts.fork) {
            SLAM_HIP_CHECK(hipEventCreateWithFlags(&g_ts.fork, hipEventDisableTiming));
            for (auto &s : g_ts.side_stream) SLAM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        }
        const size_t dl = lane_bytes(lane_F, lane_P, N, cfg, false), hl = lane_bytes(lane_F, lane_P, N, cfg, true);
        SLAM_HIP_CHECK(hipEventRecord(g_ts.fork, st));              // inputs were produced on the caller's stream
        for (int k = 0; k < 2; ++k) {
            Carver dc((char *)io->d_workspace + k * dl), hc((char *)io->h_workspace + k * hl);
            layout(lane_F, lane_P, N, cfg, dc, hc, L[k].b);
            L[k].st = g_ts.side_stream[k];
            SLAM_HIP_CHECK(hipStreamWaitEvent(L[k].st, g_ts.fork, 0));
        }
        const int C = (int)chunks.size();
        const bool serial = cfg->pipeline_serial_match != 0;
        int rc = SLAM_OK;
        for (int k = 0; k < 2; ++k) {
            rc = L[k].ensure_events();                           // lane k's matcher may wait for lane k^1's event
            if (rc) return rc;
        }
        for (int c = 0; c < 2 && c < C && !rc; ++c) {
            L[c].set_chunk(sh, chunks[c].a, chunks[c].e, chunks[c].lo, chunks[c].hi);
            L[c].match_after = serial && c > 0 ? L[c - 1].ev_matched : nullptr;
            rc = L[c].front(sh);
        }
        bool mid_done = false;                                   // mid() of chunk c already ran (see below)
        for (int c = 0; c < C && !rc; ++c) {
            Lane &ln = L[c & 1];
            if (!mid_done) rc = ln.mid(sh);                      // the other lane's kernels keep the GPU busy meanwhile
            mid_done = false;
            if (!rc && c + 2 >= C && c + 1 < C) {
                // nothing left to queue on this lane: replay the other lane's chunk BEFORE waiting for this one's
                // refit, so that its mask / refit launch does not sit behind that wait
                rc = L[(c + 1) & 1].mid(sh);
                mid_done = true;
            }
            if (!rc) rc = ln.back(sh);
            if (!rc && c + 2 < C) {
                ln.set_chunk(sh, chunks[c + 2].a, chunks[c + 2].e, chunks[c + 2].lo, chunks[c + 2].hi);
                ln.match_after = serial ? L[(c + 1) & 1].ev_matched : nullptr;    // chunk c + 1's matcher, queued earlier
                rc = ln.front(sh);
            }
        }
        if (rc) {
            // the first error is the one reported; still drain both lanes so no work outlives the call
            (void)hipStreamSynchronize(g_ts.side_stream[0]);
            (void)hipStreamSynchronize(g_ts.side_stream[1]);
            for (int k = 0; k < 2; ++k) {
                if (L[k].cs) (void)hipStreamSynchronize(L[k].cs);
                if (L[k].ds) (void)hipStreamSynchronize(L[k].ds);
            }
            return rc;
        }
        n_launch = C;
    }
    for (int k = 0; k < 5; ++k) g_timing[k] = L[0].stage_ms[k] + L[1].stage_ms[k];
    {   // matching: union of the launch intervals (with two lanes the per-launch durations overlap in time)
        std::vector<std::pair<float, float>> iv = L[0].match_iv;
        iv.insert(iv.end(), L[1].match_iv.begin(), L[1].match_iv.end());
        g_timing[0] = union_ms(iv);
    }
    g_timing[5] = now_ms() - t_begin;
    g_timing[6] = L[0].host_ms + L[1].host_ms;
    g_timing[7] = n_launch;
    return SLAM_OK;
}
}  // namespace

extern "C" int slam_track_pairs(const SlamTrackConfig *cfg, const SlamTrackIO *io, void *stream) try {
    return track_pairs(cfg, io, nullptr, stream);
}
// a host failure (allocation, a worker's exception): wait for whatever the call had queued, on every stream it may have used
SLAM_CATCH_ALL_WITH("slam_track_pairs", (void)hipDeviceSynchronize())

extern "C" int slam_track_pairs_stereo_keys(const SlamTrackConfig *cfg, const SlamTrackIO *io, const uint32_t *d_stereo_key,
                                            void *stream) try {
    SLAM_REQUIRE(((uintptr_t)d_stereo_key & 3) == 0, "slam_track_pairs_stereo_keys: d_stereo_key must be 4-byte aligned");
    return track_pairs(cfg, io, d_stereo_key, stream);
}
SLAM_CATCH_ALL_WITH("slam_track_pairs_stereo_keys", (void)hipDeviceSynchronize())
