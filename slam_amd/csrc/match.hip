// All-pairs binary-descriptor matching for gfx950.
//
// Replaces cv2.BFMatcher(NORM_HAMMING2, crossCheck=True).match and .knnMatch(k=2) as called from
// utils/utills.py:57-70, :92-102, :105-119 of the reference.
//
// Shape of the work: integer VALU (xor / or / popcount), not a GEMM and not HBM bound
// (2000 x 2000 rows x 64 B = 256 KB in, 35 lane-ops per row pair).  Mapping:
//   - every lane owns ROWS "A" rows in registers (16 VGPRs each), pre-transformed so that a
//     NORM_HAMMING2 distance over 64 bytes costs 8 x (v_xor, v_bitop3 "(a^b)|c", v_bcnt accumulate):
//     each dword pair (w0, w1) is re-packed into E = even bits of w0 | even bits of w1 << 1 and
//     O = odd bits of w0 >> 1 | odd bits of w1, so a 2-bit cell differs iff (Ea^Eb)|(Oa^Ob) has
//     its bit set (16 + 16 cells per dword pair, one popcount each).
//   - "B" rows are staged through LDS in tiles (double buffered, transformed on the way in) and
//     read back with ds_read_b128 broadcasts: one LDS read feeds ROWS x 64 distance updates.
//   - running best (cross-check) or top-2 (kNN) per A row as packed keys (distance << 20 | B row); lowest index
//     wins ties because the index is in the key's low bits.  The key/min/med3 updates are half-rate instructions
//     on gfx950 (tools/valu_peak.hip), so they sit behind a wave-uniform "does any lane improve" test and are
//     skipped for most B rows.
//   - a wavefront owns 64 * ROWS consecutive A rows; the last wavefront of an image, whose trailing row slots are
//     empty, runs a variant compiled for fewer rows.
//   - the 16 distance chains of a B row (2 rows x 8 words) are issued side by side, one step of all of them at a
//     time (16 x v_xor, 16 x v_bitop3, 16 x v_bcnt): the order matters by 15 % on this part.
//   - tile shape (2 rows/lane, 256 threads, 128-row B tiles) chosen by sweeps on the MI355X.
//   - the all-pairs kernel only finds, for every A row, its nearest (kNN: two nearest) B row, and writes those keys
//     with plain stores; a small second kernel (resolve_kernel, one workgroup per problem) merges the B splits and
//     applies the matching rule on them:
//       SLAM_MATCH_CROSSCHECK (OpenCV batchDistance(crosscheck=true), the one-directional rule of OpenCV <= 4.5):
//         A = train rows; (distance, train row) of every train row is min-ed into its nearest query's slot in LDS
//         (min is order independent, so the result is deterministic), then the slots are stored as whole rows;
//       SLAM_MATCH_CROSSCHECK_MUTUAL (newer OpenCV: both directions computed, a pair is kept only if each row is the
//         other's first-minimum nearest): both directions go through the all-pairs kernel, resolve compares them;
//       SLAM_MATCH_KNN2 (+ optional ratio test d1 < ratio * d2 in float64): merge of the per-split top-2 lists.
//     (Round 1 scattered straight from the all-pairs kernel with global atomicMin: 6.8 M device-scope atomics per
//     shard, each a 64-byte memory transaction, 0.39 GB of writes for 57 MB of keys.)
//   - a batch of few problems (a loop-closure call with 3-5 candidates) splits B over blockIdx.y until the grid fills
//     the chip; the resolve kernel merges the splits.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

#ifndef SLAM_ROWS
#define SLAM_ROWS 2
#endif
#ifndef SLAM_BLOCK
#define SLAM_BLOCK 256
#endif
#ifndef SLAM_TILE_B
#define SLAM_TILE_B 128
#endif
#ifndef SLAM_INTERLEAVE
#define SLAM_INTERLEAVE 8
#endif
#ifndef SLAM_ROW_VARIANTS
#define SLAM_ROW_VARIANTS 1
#endif
#ifndef SLAM_JUNROLL
#define SLAM_JUNROLL 1
#endif
constexpr int ROWS = SLAM_ROWS;    // A rows per lane
constexpr int BLOCK = SLAM_BLOCK;  // threads per workgroup
constexpr int TILE_A = ROWS * BLOCK;
constexpr int TILE_B = SLAM_TILE_B;   // B rows per LDS tile (64 B each)
constexpr int STAGE = TILE_B * 4 / BLOCK;   // uint4 moved per thread per tile
static_assert(STAGE * BLOCK == TILE_B * 4 && STAGE >= 1, "tile must divide evenly over the workgroup");
constexpr unsigned NONE = SLAM_KEY_NONE;

struct MatchBatch {
    const uint4 *desc;        // batched: image k at desc + k * n_max * 4
    const int *count;         // batched: rows per image
    const int *prob_a;        // batched: A image per problem
    const int *prob_b;        // batched: B image per problem
    const int *order;         // batched, optional: the problem the k-th group of A tiles works on (a permutation)
    const uint4 *a_ptr;       // single problem
    const uint4 *b_ptr;
    int na, nb;               // single problem sizes
    int n_max;                // rows per image slot
    unsigned *out1, *out2;    // nearest / second nearest B row of every A row: [problem][split][part_rows] keys
    int out_stride;           // keys per problem = nsplit * part_rows
    int part_rows;            // keys per split (>= A rows of any problem)
    unsigned *col_out;        // COLMIN: nearest A row of every B row, per A tile: [problem][A tile][part_rows] keys
    int tiles_per_prob;       // A tiles per problem
    int nsplit;               // B splits (blockIdx.y)
    int n_prob;
};

// (x, y, z, w) -> (E(x,y), E(z,w), O(x,y), O(z,w)) for NORM_HAMMING2; identity for NORM_HAMMING
template <bool NORM2>
__device__ __forceinline__ uint4 pack_cells(uint4 v) {
    if (!NORM2) return v;
    uint4 r;
    r.x = (v.x & 0x55555555u) | ((v.y & 0x55555555u) << 1);
    r.y = (v.z & 0x55555555u) | ((v.w & 0x55555555u) << 1);
    r.z = ((v.x >> 1) & 0x55555555u) | (v.y & 0xAAAAAAAAu);
    r.w = ((v.z >> 1) & 0x55555555u) | (v.w & 0xAAAAAAAAu);
    return r;
}

// popcount(x) + acc in one instruction (the compiler otherwise splits pairs into bcnt, bcnt, add3)
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

template <bool NORM2>
__device__ __forceinline__ unsigned dist_quad(const uint4 &a, const uint4 &b, unsigned acc) {
    if (NORM2) {
        // gfx950 v_bitop3_b32, truth table 0xde = (s0 ^ s2) | s1: differing cells of an (E, O) pair in 2 ops
        acc = bcnt_acc(__builtin_amdgcn_bitop3_b32(a.x, a.z ^ b.z, b.x, 0xde), acc);
        acc = bcnt_acc(__builtin_amdgcn_bitop3_b32(a.y, a.w ^ b.w, b.y, 0xde), acc);
    } else {
        acc = bcnt_acc(a.x ^ b.x, acc);
        acc = bcnt_acc(a.y ^ b.y, acc);
        acc = bcnt_acc(a.z ^ b.z, acc);
        acc = bcnt_acc(a.w ^ b.w, acc);
    }
    return acc;
}

// calls f(integral_constant<int, n>) for the run-time n in [1, N]; n <= 0 does nothing
template <int N, typename F>
__device__ __forceinline__ void dispatch_rows(int n, F &f) {
    if constexpr (N >= 1) {
        if (n == N)
            f(std::integral_constant<int, N>{});
        else
            dispatch_rows<N - 1>(n, f);
    }
}

__device__ __forceinline__ void top2_insert(unsigned &m1, unsigned &m2, unsigned key) {
    // m1 <= m2 invariant; median(m1, m2, key) is the new second best
    unsigned med;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));
    m2 = med;
    m1 = min(m1, key);
}

// minimum over every row of 16 lanes, returned in the row's last lane (15, 31, 47, 63): DPP row shifts, the moves ride
// on the v_min instructions, no LDS traffic
__device__ __forceinline__ unsigned row16_min_to_last_lane(unsigned v) {
#define SLAM_DPP_MIN(ctrl) v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, ctrl, 0xf, 0xf, false))
    SLAM_DPP_MIN(0x111);      // row_shr:1
    SLAM_DPP_MIN(0x112);      // row_shr:2
    SLAM_DPP_MIN(0x114);      // row_shr:4
    SLAM_DPP_MIN(0x118);      // row_shr:8
#undef SLAM_DPP_MIN
    return v;
}

// TOP2: keep the second best too (kNN mode); the cross-check rules need only the nearest row.
// COLMIN (mutual cross-check): besides every A row's nearest B row, every B row's nearest A row among this workgroup's
// A rows (first minimum: lowest A row on equal distance), so that one pass over the distance matrix yields both
// directions.  Per B row that is two key builds, one minimum and four DPP minima per lane and one LDS store from
// four lanes (~28 issue cycles against the 132 of the two distances), instead of a second all-pairs pass.
// RW = A rows per lane (ROWS = 2 in the shipped library; 1, 3, 4 exist in tuning builds, see plan_grid).
template <bool NORM2, bool TOP2, int RW, bool COLMIN = false>
__global__ __launch_bounds__(BLOCK) void allpairs_top2_kernel(MatchBatch mb) {
    __shared__ uint4 tile[2][TILE_B * 4];
    // COLMIN: per wavefront and B row of the tile in flight, the minima of the wavefront's four rows of 16 lanes (plain
    // stores: LDS atomics from a few lanes per instruction turned out several times dearer than the distances)
    constexpr int WAVES = BLOCK / 64;
    __shared__ __align__(16) unsigned colmin[COLMIN ? WAVES : 1][COLMIN ? TILE_B : 1][COLMIN ? 4 : 1];
    constexpr int ROWS = RW;                 // shadows the file-scope default inside the kernel
    constexpr int TILE_A = ROWS * BLOCK;

    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
    // range of logical blocks; the A tiles of one problem then share one L2 for their B rows.
    const unsigned nblk = gridDim.x;
    const unsigned logical = (blockIdx.x % 8u) * (nblk / 8u) + blockIdx.x / 8u;
    const unsigned total = (unsigned)mb.tiles_per_prob * (unsigned)mb.n_prob;
    if (logical >= total) return;
    // `order` lets the caller put problems that share an image next to each other in launch order (same XCD, same
    // time), so the image is fetched from HBM once; results do not depend on it
    const int slot = logical / mb.tiles_per_prob;
    const int p = mb.order ? mb.order[slot] : slot;
    const int tile_a = logical % mb.tiles_per_prob;
    const int split = blockIdx.y;

    const uint4 *A, *B;
    int na, nb;
    if (mb.prob_a) {
        const int ia = mb.prob_a[p], ib = mb.prob_b[p];
        A = mb.desc + (size_t)ia * mb.n_max * 4;
        B = mb.desc + (size_t)ib * mb.n_max * 4;
        // counts come from the caller's device array: clamp them to the image slot, so that a bad count can make a
        // result meaningless but never an access out of bounds
        na = min(max(mb.count[ia], 0), mb.n_max);
        nb = min(max(mb.count[ib], 0), mb.n_max);
    } else {
        A = mb.a_ptr;
        B = mb.b_ptr;
        na = mb.na;
        nb = mb.nb;
    }
    const int row0 = tile_a * TILE_A;
    if (row0 >= na) return;

    // B range of this split, in whole tiles
    const int tiles_b = (nb + TILE_B - 1) / TILE_B;
    // Front-loaded splits: the workgroups of the LAST split are dispatched last (blockIdx.y is the slow index), and a
    // launch ends with about half a "round" of its last workgroups running on a half-empty chip.  One more tile for
    // every earlier split leaves the last one short -- 16 tiles over 3 splits: 7, 7, 2 instead of 6, 6, 4 -- so the
    // ragged end of a small grid is made of short workgroups (when that leaves the last split a tile at all).
    int tiles_per_split = (tiles_b + mb.nsplit - 1) / mb.nsplit;
#ifndef SLAM_SPLIT_EVEN
    if (mb.nsplit > 1 && (mb.nsplit - 1) * (tiles_per_split + 1) < tiles_b) ++tiles_per_split;
#endif
    const int tb0 = split * tiles_per_split;
    const int tb1 = min(tiles_b, tb0 + tiles_per_split);

    const int tid = threadIdx.x;
    // A wavefront owns 64 * ROWS consecutive rows, slot r = rows [r, r + 1) * 64 of them.  Images hold a few rows less
    // than the tile grid (1970 of 2048 on the bench data), so the last wavefront of a problem has whole slots without
    // rows: `nact` (wave-uniform) is the number of slots that have any, and the scan below runs the variant compiled for
    // that many rows instead of computing distances of zero rows.
    const int wave_row0 = row0 + (tid >> 6) * (ROWS * 64);
    auto slot_row = [&](int r) { return wave_row0 + r * 64 + (tid & 63); };
    const int nact = __builtin_amdgcn_readfirstlane(max(0, min(ROWS, (na - wave_row0 + 63) / 64)));
    uint4 a[ROWS][4];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = slot_row(r);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < na) v = A[(size_t)row * 4 + c];
            a[r][c] = pack_cells<NORM2>(v);
        }
    }
    unsigned m1[ROWS], m2[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) m1[r] = m2[r] = NONE;
    unsigned gate[ROWS];          // distance a new row must beat: that of the worst entry kept
#pragma unroll
    for (int r = 0; r < ROWS; ++r) gate[r] = NONE >> SLAM_KEY_SHIFT;
    // COLMIN: the low key bits of this lane's rows; rows beyond the image get all ones, so their keys are SLAM_KEY_NONE
    unsigned rowbits[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rowbits[r] = slot_row(r) < na ? (unsigned)slot_row(r) : NONE;
    unsigned *col_out = COLMIN ? mb.col_out + ((size_t)p * mb.tiles_per_prob + tile_a) * mb.part_rows : nullptr;

    // prologue: first tile -> LDS.  A tile is TILE_B*4 uint4; each thread moves STAGE of them, coalesced.
    uint4 stage[STAGE];
    auto load_tile = [&](int tb) {
        const size_t base = (size_t)tb * TILE_B * 4;
        const size_t lim = (size_t)nb * 4;
#pragma unroll
        for (int k = 0; k < STAGE; ++k) {
            const size_t idx = base + k * BLOCK + tid;
            stage[k] = idx < lim ? B[idx] : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int k = 0; k < STAGE; ++k) tile[buf][k * BLOCK + tid] = pack_cells<NORM2>(stage[k]);
    };

    if (tb0 < tb1) {
        load_tile(tb0);
        store_tile(0);
    }
    __syncthreads();

    for (int tb = tb0; tb < tb1; ++tb) {
        const int buf = (tb - tb0) & 1;
        const bool more = tb + 1 < tb1;
        if (more) load_tile(tb + 1);                 // global loads in flight during the compute below

        const int jbase = tb * TILE_B;
        const int rows_here = min(TILE_B, nb - jbase);
        const uint4 *t = tile[buf];
        // NR = active row slots of this wavefront (compile-time in each variant)
        auto scan_tile = [&](auto nr_tag) {
            constexpr int NR = decltype(nr_tag)::value;
            // distances of B row j against this lane's NR A rows; true if any of them would enter its list
            auto dist_row = [&](int j, unsigned (&d)[ROWS]) {
                const uint4 b[4] = {t[j * 4 + 0], t[j * 4 + 1], t[j * 4 + 2], t[j * 4 + 3]};
                bool improves = false;
#if SLAM_INTERLEAVE
                // The chains (xor -> bitop3 -> bcnt) of NR rows x G words are issued side by side, one step of all of
                // them at a time: a VALU instruction that depends on one issued a few slots earlier waits, and the
                // scheduler, left to itself, may emit the rows one after the other.  Measured on the bench shard
                // (matcher ms): chains one after the other 17.9; 4 side by side 17.2; 8: 16.6; see DESIGN.md.
                constexpr int G = SLAM_INTERLEAVE;          // words per group (NORM2: a uint4 holds 2, else 4)
                constexpr int WORDS = NORM2 ? 8 : 16;
                static_assert(WORDS % G == 0, "group must divide the row");
                unsigned x[G][ROWS];
#define SLAM_STEP(expr)                                  \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) expr; \
    __builtin_amdgcn_sched_barrier(0);
                auto w_e = [&](const uint4 &q, int k) { return k ? q.y : q.x; };     // NORM2: E words of a quad
                auto w_o = [&](const uint4 &q, int k) { return k ? q.w : q.z; };     //        O words
                auto w_h = [&](const uint4 &q, int k) { return k == 0 ? q.x : k == 1 ? q.y : k == 2 ? q.z : q.w; };
                SLAM_STEP(d[r] = 0)
#pragma unroll
                for (int w0 = 0; w0 < WORDS; w0 += G) {
                    if (NORM2) {
#pragma unroll
                        for (int g = 0; g < G; ++g) SLAM_STEP(x[g][r] = w_o(a[r][(w0 + g) / 2], (w0 + g) & 1) ^ w_o(b[(w0 + g) / 2], (w0 + g) & 1))
#pragma unroll
                        for (int g = 0; g < G; ++g) SLAM_STEP(x[g][r] = __builtin_amdgcn_bitop3_b32(w_e(a[r][(w0 + g) / 2], (w0 + g) & 1), x[g][r], w_e(b[(w0 + g) / 2], (w0 + g) & 1), 0xde))
                    } else {
#pragma unroll
                        for (int g = 0; g < G; ++g) SLAM_STEP(x[g][r] = w_h(a[r][(w0 + g) / 4], (w0 + g) & 3) ^ w_h(b[(w0 + g) / 4], (w0 + g) & 3))
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) SLAM_STEP(d[r] = bcnt_acc(x[g][r], d[r]))
                }
#undef SLAM_STEP
#pragma unroll
                for (int r = 0; r < NR; ++r) improves |= d[r] < gate[r];
#else
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    unsigned acc = 0;
                    acc = dist_quad<NORM2>(a[r][0], b[0], acc);
                    acc = dist_quad<NORM2>(a[r][1], b[1], acc);
                    acc = dist_quad<NORM2>(a[r][2], b[2], acc);
                    acc = dist_quad<NORM2>(a[r][3], b[3], acc);
                    d[r] = acc;
                    improves |= acc < gate[r];
                }
#endif
                return improves;
            };
            auto insert_row = [&](int j, const unsigned (&d)[ROWS]) {
                const unsigned jg = (unsigned)(jbase + j);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const unsigned key = (d[r] << SLAM_KEY_SHIFT) | jg;
                    if (TOP2) {
                        top2_insert(m1[r], m2[r], key);
                        gate[r] = m2[r] >> SLAM_KEY_SHIFT;
                    } else {
                        m1[r] = min(m1[r], key);
                        gate[r] = m1[r] >> SLAM_KEY_SHIFT;
                    }
                }
            };
            // B rows are scanned in ascending order, so a row enters a lane's list iff its distance is strictly below
            // the list's worst distance.  The key / min / med3 updates (half-rate instructions) are skipped for the
            // whole wave when no lane improves, which is the common case after the first few hundred rows.  JU rows
            // share one such test (inserting a row that does not improve is a no-op, so testing them together only
            // makes the rare path slightly less rare); the rows of a group are inserted in ascending order.
            // COLMIN: this wavefront's nearest A row for B row j -> the tile's column slot in LDS.  Keys carry the A row
            // in their low bits, so a minimum over keys is the first minimum (lowest A row on equal distance).
            auto col_update = [&](int j, const unsigned (&d)[ROWS]) {
                unsigned m = (d[0] << SLAM_KEY_SHIFT) | rowbits[0];
#pragma unroll
                for (int r = 1; r < NR; ++r) m = min(m, (d[r] << SLAM_KEY_SHIFT) | rowbits[r]);
                m = row16_min_to_last_lane(m);
                if ((tid & 15) == 15) colmin[tid >> 6][j][(tid >> 4) & 3] = m;
            };
            constexpr int JU = SLAM_JUNROLL;
            int j = 0;
            for (; j + JU <= rows_here; j += JU) {
                unsigned d[JU][ROWS];
                bool improves = false;
#pragma unroll
                for (int u = 0; u < JU; ++u) improves |= dist_row(j + u, d[u]);
                if (COLMIN) {
#pragma unroll
                    for (int u = 0; u < JU; ++u) col_update(j + u, d[u]);
                }
                if (__any(improves)) {
#pragma unroll
                    for (int u = 0; u < JU; ++u) insert_row(j + u, d[u]);
                }
            }
            for (; j < rows_here; ++j) {
                unsigned d[ROWS];
                const bool improves = dist_row(j, d);
                if (COLMIN) col_update(j, d);
                if (__any(improves)) insert_row(j, d);
            }
        };
#if SLAM_ROW_VARIANTS
        dispatch_rows<ROWS>(nact, scan_tile);
#else
        scan_tile(std::integral_constant<int, ROWS>{});
#endif
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        if (COLMIN) {
            // the tile's column minima: 4 wavefronts x 4 lane rows per B row -> one key per B row of this A tile
            if (tid < rows_here) {
                unsigned m = NONE;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) {
                    const int wrow0 = row0 + w * (ROWS * 64);
                    if (wrow0 < na) {                                   // wavefronts without rows wrote nothing
                        const uint4 q = *(const uint4 *)colmin[w][tid];
                        m = min(min(m, min(q.x, q.y)), min(q.z, q.w));
                    }
                }
                col_out[jbase + tid] = m;
            }
            __syncthreads();                                            // before the next tile's scan overwrites the slots
        }
    }

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = slot_row(r);
        if (row >= na) continue;
        const size_t o = (size_t)p * mb.out_stride + (size_t)split * mb.part_rows + row;
        mb.out1[o] = m1[r];
        if (TOP2) mb.out2[o] = m2[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Second kernel: one workgroup per problem merges the B splits of the per-row keys and applies the matching rule.
struct ResolveArgs {
    const unsigned *part1, *part2;   // all-pairs output: [problem (x2 directions for MUTUAL)][split][part_rows]
    int nsplit, part_rows, part_stride;
    const int *count, *prob_q, *prob_t;   // batched: rows per image, query / train image per problem
    int nq, nt;                      // single problem
    int n_prob;
    unsigned *out1, *out2;           // final keys, out_stride per problem
    int out_stride, out_rows;        // out_rows keys of every problem are written (rows beyond nq: SLAM_KEY_NONE)
    double ratio;                    // KNN2: > 0 applies the ratio test to key1
    int slots_in_lds;                // CROSSCHECK: query slots fit the LDS array (else global atomics on a pre-set out1)
    // MUTUAL, fused form: the query direction comes from the same pass as per-A-tile column minima
    const unsigned *col;             // [problem][A tile][part_rows], or NULL: second direction in part1 behind the first
    int col_tiles, col_tile_rows;    // A tiles per problem slot, train rows per A tile
};

constexpr int RESOLVE_BLOCK = 256;
constexpr int LDS_SLOTS = 8192;      // 32 KB: query rows of one problem whose cross-check slots are reduced in LDS

__device__ __forceinline__ unsigned merged_top1(const unsigned *part, int nsplit, int part_rows, int row) {
    unsigned m = NONE;
    for (int s = 0; s < nsplit; ++s) m = min(m, part[(size_t)s * part_rows + row]);
    return m;
}

template <int MODE>
__global__ __launch_bounds__(RESOLVE_BLOCK) void resolve_kernel(ResolveArgs ra) {
    __shared__ unsigned slot[MODE == SLAM_MATCH_CROSSCHECK ? LDS_SLOTS : 1];
    const int p = blockIdx.x;
    int nq, nt;
    if (ra.count) {
        nq = min(max(ra.count[ra.prob_q[p]], 0), ra.part_rows);        // clamped as in the all-pairs kernel
        nt = min(max(ra.count[ra.prob_t[p]], 0), ra.part_rows);
    } else {
        nq = ra.nq;
        nt = ra.nt;
    }
    unsigned *out1 = ra.out1 + (size_t)p * ra.out_stride;
    const unsigned idx_mask = SLAM_MAX_ROWS - 1;
    if (MODE == SLAM_MATCH_CROSSCHECK) {
        // lanes of the all-pairs kernel owned TRAIN rows: part1[j] = (distance, nearest query) of train row j.
        // OpenCV's legacy rule: per query the train row of least distance among those whose nearest query it is
        // (ties: lowest train row) = min over (distance << 20 | train row) per query slot.
        const unsigned *part = ra.part1 + (size_t)p * ra.part_stride;
        if (ra.slots_in_lds) {
            for (int i = threadIdx.x; i < nq; i += RESOLVE_BLOCK) slot[i] = NONE;
            __syncthreads();
            for (int j = threadIdx.x; j < nt; j += RESOLVE_BLOCK) {
                const unsigned m = merged_top1(part, ra.nsplit, ra.part_rows, j);
                if (m != NONE) atomicMin(&slot[m & idx_mask], (m & ~idx_mask) | (unsigned)j);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < ra.out_rows; i += RESOLVE_BLOCK) out1[i] = i < nq ? slot[i] : NONE;
        } else {
            // more query rows than LDS slots: out1 was set to SLAM_KEY_NONE by the host; scatter with global atomics
            for (int j = threadIdx.x; j < nt; j += RESOLVE_BLOCK) {
                const unsigned m = merged_top1(part, ra.nsplit, ra.part_rows, j);
                if (m != NONE) atomicMin(out1 + (m & idx_mask), (m & ~idx_mask) | (unsigned)j);
            }
        }
    } else if (MODE == SLAM_MATCH_CROSSCHECK_MUTUAL) {
        // direction T (problem p): lanes owned train rows, part = nearest query of every train row;
        // direction Q (problem n_prob + p): lanes owned query rows, part = nearest train of every query row.
        // Query i keeps its nearest train j (first minimum) iff j's nearest query (first minimum) is i.
        const unsigned *part_t = ra.part1 + (size_t)p * ra.part_stride;
        // nearest train of every query row: column minima of the train tiles that hold rows (fused pass), or the
        // second all-pairs pass' lists
        const unsigned *part_q = ra.col ? ra.col + (size_t)p * ra.col_tiles * ra.part_rows
                                        : ra.part1 + (size_t)(ra.n_prob + p) * ra.part_stride;
        const int q_lists = ra.col ? (nt + ra.col_tile_rows - 1) / ra.col_tile_rows : ra.nsplit;
        for (int i = threadIdx.x; i < ra.out_rows; i += RESOLVE_BLOCK) {
            unsigned key = NONE;
            if (i < nq) {
                const unsigned mq = merged_top1(part_q, q_lists, ra.part_rows, i);
                if (mq != NONE) {
                    const unsigned mt = merged_top1(part_t, ra.nsplit, ra.part_rows, (int)(mq & idx_mask));
                    if (mt != NONE && (mt & idx_mask) == (unsigned)i) key = mq;
                }
            }
            out1[i] = key;
        }
    } else {
        // kNN (k = 2): lanes owned query rows; merge the per-split lists (lower train row first on equal distance)
        const unsigned *p1 = ra.part1 + (size_t)p * ra.part_stride, *p2 = ra.part2 + (size_t)p * ra.part_stride;
        unsigned *out2 = ra.out2 + (size_t)p * ra.out_stride;
        for (int i = threadIdx.x; i < ra.out_rows; i += RESOLVE_BLOCK) {
            unsigned m1 = NONE, m2 = NONE;
            if (i < nq)
                for (int s = 0; s < ra.nsplit; ++s) {
                    top2_insert(m1, m2, p1[(size_t)s * ra.part_rows + i]);
                    top2_insert(m1, m2, p2[(size_t)s * ra.part_rows + i]);
                }
            if (ra.ratio > 0.0 && m1 != NONE) {
                // significance_test (utils/utills.py:131-152): first.distance < ratio * second.distance in float64;
                // a query with a single neighbour has no `second` (the reference cannot unpack it): dropped
                const double d1 = (double)(m1 >> SLAM_KEY_SHIFT), d2 = (double)(m2 >> SLAM_KEY_SHIFT);
                if (m2 == NONE || !(d1 < ra.ratio * d2)) m1 = NONE;
            }
            out1[i] = m1;
            out2[i] = m2;
        }
    }
}

__global__ void unpack_keys_kernel(const unsigned *key, int n, int *index, int *distance) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = key[i];
    index[i] = k == NONE ? -1 : (int)(k & (SLAM_MAX_ROWS - 1));
    distance[i] = k == NONE ? INT32_MAX : (int)(k >> SLAM_KEY_SHIFT);
}

constexpr int MAX_SPLIT = 64;
// a plan that splits B launches at most this many workgroups (plan_grid: the target, rounded up by less than one
// split's worth of A tiles, i.e. less than twice the target)
constexpr long PLAN_MAX_WORKGROUPS = 2 * 7000;

template <int RW>
void launch_allpairs_rows(MatchBatch mb, int a_rows, int norm, hipStream_t st) {
    mb.tiles_per_prob = slam_div_up(a_rows, RW * BLOCK);
    const unsigned total = (unsigned)mb.tiles_per_prob * (unsigned)mb.n_prob;
    dim3 grid((total + 7u) / 8u * 8u, mb.nsplit);
    const bool top2 = mb.out2 != nullptr, colmin = mb.col_out != nullptr;
    if (norm == SLAM_NORM_HAMMING2) {
        if (top2)
            hipLaunchKernelGGL((allpairs_top2_kernel<true, true, RW>), grid, dim3(BLOCK), 0, st, mb);
        else if (colmin)
            hipLaunchKernelGGL((allpairs_top2_kernel<true, false, RW, true>), grid, dim3(BLOCK), 0, st, mb);
        else
            hipLaunchKernelGGL((allpairs_top2_kernel<true, false, RW>), grid, dim3(BLOCK), 0, st, mb);
    } else {
        if (top2)
            hipLaunchKernelGGL((allpairs_top2_kernel<false, true, RW>), grid, dim3(BLOCK), 0, st, mb);
        else if (colmin)
            hipLaunchKernelGGL((allpairs_top2_kernel<false, false, RW, true>), grid, dim3(BLOCK), 0, st, mb);
        else
            hipLaunchKernelGGL((allpairs_top2_kernel<false, false, RW>), grid, dim3(BLOCK), 0, st, mb);
    }
}

// How a batch of n_prob problems of a_rows x b_rows is laid over the chip: A rows per lane and B splits.
// Measured on the MI355X (tools/sweep_grid_plan.py, profiles/r2_matcher_grid_plan_sweep.txt):
//   - 2 rows per lane (79 VGPRs, 6 wavefronts per SIMD) beats 4 rows per lane (139 VGPRs, 3 per SIMD) at every batch
//     size now that the instruction order inside a B row is pinned: 14.0 against 15.5 ms for the 3450-frame shard,
//     1.95 against 2.19 ms for an eighth of it.  (Round 1 chose 4 rows on a sweep made before the order was pinned.)
//   - a grid of fewer than ~7000 workgroups leaves part of the chip idle at its ragged end: B is split until there
//     are about that many (at most 16 splits, each at least one B tile); the resolve kernel merges the splits.
struct GridPlan { int rows, nsplit; };
constexpr long PLAN_TARGET_WORKGROUPS = 7000;

GridPlan plan_grid(int n_prob, int a_rows, int b_rows) {
    const int tiles_b = std::max(1, slam_div_up(b_rows, TILE_B));
#ifdef SLAM_MATCH_ALL_ROW_VARIANTS
    // tuning builds only (tools/sweep_grid_plan.py builds with every row variant): "rows,nsplit" from the environment.
    // The shipped library has one row variant and reads no environment variable here.
    if (const char *e = std::getenv("SLAM_MATCH_PLAN")) {
        int r = 0, ns = 0;
        if (std::sscanf(e, "%d,%d", &r, &ns) == 2 && r >= 1 && r <= 4 && ns >= 1)
            return {r, std::min(std::min(ns, MAX_SPLIT), tiles_b)};
    }
#endif
    const long tiles_a = (long)slam_div_up(a_rows, TILE_A) * (n_prob > 0 ? n_prob : 1);
    long ns = (PLAN_TARGET_WORKGROUPS + tiles_a - 1) / tiles_a;
    ns = std::min<long>(ns, std::min(tiles_b, 16));
    return {ROWS, ns < 1 ? 1 : (int)ns};
}

void launch_allpairs(const MatchBatch &mb, int a_rows, int norm, int rows, hipStream_t st) {
#ifdef SLAM_MATCH_ALL_ROW_VARIANTS
    if (rows == 1) return launch_allpairs_rows<1>(mb, a_rows, norm, st);
    if (rows == 3) return launch_allpairs_rows<3>(mb, a_rows, norm, st);
    if (rows == 4) return launch_allpairs_rows<4>(mb, a_rows, norm, st);
    if (rows == 2) return launch_allpairs_rows<2>(mb, a_rows, norm, st);
#endif
    (void)rows;
    launch_allpairs_rows<ROWS>(mb, a_rows, norm, st);
}

// The mutual rule runs fused (column minima out of the train-side pass) while a problem slot has at most this many
// A tiles (8192 train rows at 2 rows per lane); beyond that, two all-pairs passes.
constexpr int FUSED_MUTUAL_MAX_TILES = 16;

// keys of scratch a match call needs: n_prob x nsplit x part_rows per direction / list, or for the fused mutual rule
// n_prob x (nsplit + A tiles per problem) x part_rows
size_t scratch_keys(int n_prob, int part_rows, int nsplit, int mode, int col_tiles) {
    size_t lists = mode == SLAM_MATCH_CROSSCHECK ? nsplit : 2 * (size_t)nsplit;     // MUTUAL: two directions; KNN2: two lists
    if (mode == SLAM_MATCH_CROSSCHECK_MUTUAL && col_tiles > 0) lists = (size_t)nsplit + col_tiles;
    return lists * (size_t)n_prob * (size_t)part_rows;
}

template <int MODE>
void launch_resolve(const ResolveArgs &ra, hipStream_t st) {
    hipLaunchKernelGGL((resolve_kernel<MODE>), dim3(ra.n_prob), dim3(RESOLVE_BLOCK), 0, st, ra);
}

// The whole match: all-pairs (both directions for MUTUAL) into `scratch`, then resolve into out1 / out2.
// `mb` arrives with the QUERY side in (prob_a | a_ptr, na) and the TRAIN side in (prob_b | b_ptr, nb).
int run_match(const char *who, MatchBatch mb, int q_rows, int t_rows, int norm, int mode, double ratio, uint32_t *d_key1,
              uint32_t *d_key2, int out_stride, int out_rows, unsigned *scratch, size_t scratch_bytes, hipStream_t st) {
    const bool knn = mode == SLAM_MATCH_KNN2, mutual = mode == SLAM_MATCH_CROSSCHECK_MUTUAL;
    const int n_prob = mb.n_prob;
    // one split count for both directions, so that they share the partial layout
    const int part_rows = q_rows > t_rows ? q_rows : t_rows;
    // lanes own the query rows (kNN, and the second pass of the mutual rule) or the train rows (cross-check)
    const GridPlan plan_q = plan_grid(n_prob, q_rows, t_rows), plan_t = plan_grid(n_prob, t_rows, q_rows);
    int nsplit = knn ? plan_q.nsplit : plan_t.nsplit;
    if (mutual) nsplit = std::min(nsplit, plan_q.nsplit);
    // mutual rule: one pass with column minima when the train side has few A tiles, else two passes
    const int t_tiles = slam_div_up(t_rows, plan_t.rows * BLOCK);
    const bool fused = mutual && t_tiles <= FUSED_MUTUAL_MAX_TILES;
    if (fused) nsplit = plan_t.nsplit;
    const size_t need = scratch_keys(n_prob, part_rows, nsplit, mode, fused ? t_tiles : 0) * sizeof(unsigned);
    if (scratch_bytes < need || !scratch) {
        slam_set_error("%s: workspace %zu B < %zu B", who, scratch_bytes, need);
        return SLAM_E_WORKSPACE;
    }
    const size_t dir_keys = (size_t)n_prob * nsplit * part_rows;
    MatchBatch fwd = mb;                       // lanes own QUERY rows, scan the train rows
    MatchBatch rev = mb;                       // lanes own TRAIN rows, scan the query rows
    std::swap(rev.prob_a, rev.prob_b);
    std::swap(rev.a_ptr, rev.b_ptr);
    std::swap(rev.na, rev.nb);
    for (MatchBatch *m : {&fwd, &rev}) {
        m->nsplit = nsplit;
        m->part_rows = part_rows;
        m->out_stride = nsplit * part_rows;
        m->out2 = nullptr;
        m->col_out = nullptr;
    }
    if (knn) {
        fwd.out1 = scratch;
        fwd.out2 = scratch + dir_keys;
        launch_allpairs(fwd, q_rows, norm, plan_q.rows, st);
    } else {
        rev.out1 = scratch;
        if (fused) rev.col_out = scratch + dir_keys;
        launch_allpairs(rev, t_rows, norm, plan_t.rows, st);
        if (mutual && !fused) {
            fwd.out1 = scratch + dir_keys;
            launch_allpairs(fwd, q_rows, norm, plan_q.rows, st);
        }
    }
    SLAM_HIP_CHECK(hipGetLastError());

    ResolveArgs ra{};
    ra.part1 = scratch;
    ra.part2 = knn ? scratch + dir_keys : nullptr;
    ra.nsplit = nsplit;
    ra.part_rows = part_rows;
    ra.part_stride = nsplit * part_rows;
    ra.count = mb.count;
    ra.prob_q = mb.prob_a;
    ra.prob_t = mb.prob_b;
    ra.nq = mb.na;
    ra.nt = mb.nb;
    ra.n_prob = n_prob;
    ra.out1 = d_key1;
    ra.out2 = d_key2;
    ra.out_stride = out_stride;
    ra.out_rows = out_rows;
    ra.ratio = ratio;
    ra.slots_in_lds = q_rows <= LDS_SLOTS;
    ra.col = fused ? scratch + dir_keys : nullptr;
    ra.col_tiles = t_tiles;
    ra.col_tile_rows = plan_t.rows * BLOCK;
    if (mode == SLAM_MATCH_CROSSCHECK) {
        if (!ra.slots_in_lds)
            SLAM_HIP_CHECK(hipMemsetAsync(d_key1, 0xFF, sizeof(uint32_t) * ((size_t)(n_prob - 1) * out_stride + out_rows), st));
        launch_resolve<SLAM_MATCH_CROSSCHECK>(ra, st);
    } else if (mutual) {
        launch_resolve<SLAM_MATCH_CROSSCHECK_MUTUAL>(ra, st);
    } else {
        launch_resolve<SLAM_MATCH_KNN2>(ra, st);
    }
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

int check_mode(const char *who, int norm, int mode) {
    SLAM_REQUIRE(norm == SLAM_NORM_HAMMING || norm == SLAM_NORM_HAMMING2, "%s: norm %d unsupported", who, norm);
    SLAM_REQUIRE(mode == SLAM_MATCH_CROSSCHECK || mode == SLAM_MATCH_KNN2 || mode == SLAM_MATCH_CROSSCHECK_MUTUAL,
                 "%s: mode %d unsupported", who, mode);
    return SLAM_OK;
}

// One problem.  Every key the resolve kernel reads is written by the all-pairs kernel first (rows < na of every split;
// empty splits write SLAM_KEY_NONE), so `scratch` needs no clearing between calls.
int match_single(const char *who, const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train, int norm,
                 int mode, uint32_t *d_key1, uint32_t *d_key2, unsigned *scratch, size_t scratch_bytes, hipStream_t st) {
    if (n_train == 0) {
        SLAM_HIP_CHECK(hipMemsetAsync(d_key1, 0xFF, sizeof(uint32_t) * (size_t)n_query, st));
        if (mode == SLAM_MATCH_KNN2) SLAM_HIP_CHECK(hipMemsetAsync(d_key2, 0xFF, sizeof(uint32_t) * (size_t)n_query, st));
        return SLAM_OK;
    }
    MatchBatch mb{};
    mb.a_ptr = (const uint4 *)d_query;
    mb.b_ptr = (const uint4 *)d_train;
    mb.na = n_query;
    mb.nb = n_train;
    mb.n_prob = 1;
    return run_match(who, mb, n_query, n_train, norm, mode, 0.0, d_key1, d_key2, 0, n_query, scratch, scratch_bytes, st);
}

int match_single_check_args(const char *who, const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train,
                            int norm, int mode, const uint32_t *d_key1, const uint32_t *d_key2, int max_rows) {
    const int rc = check_mode(who, norm, mode);
    if (rc) return rc;
    SLAM_REQUIRE(n_query >= 0 && n_train >= 0 && n_query < max_rows && n_train < max_rows,
                 "%s: row counts %d/%d out of range (each must be below %d)", who, n_query, n_train, max_rows);
    if (n_query == 0) return SLAM_OK;
    SLAM_REQUIRE(d_key1 && (mode != SLAM_MATCH_KNN2 || d_key2), "%s: null output", who);
    SLAM_REQUIRE(d_query && (d_train || n_train == 0), "%s: null input", who);
    SLAM_REQUIRE(((uintptr_t)d_query & 15) == 0 && ((uintptr_t)d_train & 15) == 0,
                 "%s: descriptor rows must be 16-byte aligned", who);
    return SLAM_OK;
}

}  // namespace

extern "C" size_t slam_match_workspace_bytes(int n_query, int n_train) {
    const size_t n = (size_t)(n_query > n_train ? n_query : n_train);
    return (2 * MAX_SPLIT + FUSED_MUTUAL_MAX_TILES) * n * sizeof(unsigned) + 256;   // two lists / directions x at most 64 splits
}

extern "C" int slam_match_hamming(const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train, int norm,
                                  int mode, uint32_t *d_key1, uint32_t *d_key2, void *d_workspace,
                                  size_t workspace_bytes, void *stream) {
    const char *who = "slam_match_hamming";
    const int rc = match_single_check_args(who, d_query, n_query, d_train, n_train, norm, mode, d_key1, d_key2, SLAM_MAX_ROWS);
    if (rc != SLAM_OK || n_query == 0) return rc;
    return match_single(who, d_query, n_query, d_train, n_train, norm, mode, d_key1, d_key2, (unsigned *)d_workspace,
                        workspace_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The same matcher for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that
// forbid hipMalloc).  The per-split partial lists -- the scratch buffer slam_match_hamming takes from its caller's
// workspace -- live in a file-scope device array instead: (2 lists x 64 splits + 16 column tiles) x
// SLAM_MATCH_NOALLOC_MAX_ROWS keys = 4.5 MiB, held by every GPU the library runs on until the process ends.  In host code the array's NAME is only a
// stand-in for the symbol; its address on a device comes from hipGetSymbolAddress, asked once per device with that
// device current, and only that pointer is handed to kernels.
constexpr size_t NOALLOC_WORDS = (2 * (size_t)MAX_SPLIT + FUSED_MUTUAL_MAX_TILES) * SLAM_MATCH_NOALLOC_MAX_ROWS;
static __device__ __align__(256) unsigned scratch[NOALLOC_WORDS];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned *g_scratch_ptr[MAX_DEVICES];        // per device: that device's copy of `scratch`

// the current device's copy of `scratch` (looked up on the first call on that device)
int scratch_of_current_device(const char *who, hipStream_t st, unsigned **out) {
    int dev = -1;
    SLAM_HIP_CHECK(hipGetDevice(&dev));
    SLAM_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "%s: device ordinal %d not supported (max %d)", who, dev, MAX_DEVICES);
    if (st) {
        int sdev = -1;
        SLAM_HIP_CHECK(hipStreamGetDevice(st, &sdev));
        SLAM_REQUIRE(sdev == dev, "%s: stream belongs to device %d but device %d is current", who, sdev, dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        SLAM_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        SLAM_REQUIRE(p != nullptr, "%s: hipGetSymbolAddress returned a null address", who);
        g_scratch_ptr[dev] = (unsigned *)p;
    }
    *out = g_scratch_ptr[dev];
    return SLAM_OK;
}

}  // namespace

extern "C" int slam_match_hamming_noalloc(const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train,
                                          int norm, int mode, uint32_t *d_key1, uint32_t *d_key2, void *stream) {
    const char *who = "slam_match_hamming_noalloc";
    // sizes beyond the static scratch are refused here, before anything is launched
    const int rc = match_single_check_args(who, d_query, n_query, d_train, n_train, norm, mode, d_key1, d_key2,
                                           SLAM_MATCH_NOALLOC_MAX_ROWS + 1);
    if (rc != SLAM_OK || n_query == 0) return rc;
    unsigned *part = nullptr;
    const int rs = scratch_of_current_device(who, (hipStream_t)stream, &part);
    if (rs != SLAM_OK) return rs;
    return match_single(who, d_query, n_query, d_train, n_train, norm, mode, d_key1, d_key2, part,
                        NOALLOC_WORDS * sizeof(unsigned), (hipStream_t)stream);
}

// Enough for ANY batch of up to n_prob problems (a caller sizes one workspace for its largest batch and reuses it for
// smaller ones, which split B more often): plans never launch more than PLAN_MAX_WORKGROUPS workgroups when they split,
// so n x nsplit(n) <= max(n, min(n x B tiles, PLAN_MAX_WORKGROUPS / A tiles per problem)).
extern "C" size_t slam_match_batched_workspace_bytes(int n_prob, int n_max, int mode) {
    if (n_prob <= 0 || n_max <= 0) return 256;
    const long tiles_b = slam_div_up(n_max, TILE_B), tpp = slam_div_up(n_max, TILE_A);
    const long split_bound = std::min<long>((long)n_prob * std::min<long>(tiles_b, MAX_SPLIT), PLAN_MAX_WORKGROUPS / tpp);
    const long units = std::max<long>(n_prob, split_bound);
    size_t keys = scratch_keys(1, n_max, 1, mode, 0) * (size_t)units;
    if (mode == SLAM_MATCH_CROSSCHECK_MUTUAL)       // fused form: units split lists + the A tiles' column minima per problem
        keys = std::max(keys, ((size_t)units + (size_t)n_prob * std::min<long>(tpp, FUSED_MUTUAL_MAX_TILES)) * (size_t)n_max);
    return keys * sizeof(unsigned) + 256;
}

extern "C" int slam_match_hamming_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max,
                                          const int32_t *d_prob_query, const int32_t *d_prob_train, int n_prob,
                                          int norm, int mode, double ratio, uint32_t *d_key1, uint32_t *d_key2,
                                          void *d_workspace, size_t workspace_bytes, void *stream) {
    return slam_launch_match_batched(d_desc, d_count, n_max, d_prob_query, d_prob_train, nullptr, n_prob, norm, mode,
                                     ratio, d_key1, d_key2, d_workspace, workspace_bytes, (hipStream_t)stream);
}

int slam_launch_match_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max, const int32_t *d_prob_query,
                              const int32_t *d_prob_train, const int32_t *d_order, int n_prob, int norm, int mode,
                              double ratio, uint32_t *d_key1, uint32_t *d_key2, void *d_workspace, size_t workspace_bytes,
                              hipStream_t st) {
    const char *who = "slam_match_hamming_batched";
    const int rc = check_mode(who, norm, mode);
    if (rc) return rc;
    SLAM_REQUIRE(n_max > 0 && n_max < SLAM_MAX_ROWS && n_prob >= 0, "%s: bad sizes", who);
    if (n_prob == 0) return SLAM_OK;
    SLAM_REQUIRE(d_desc && d_count && d_prob_query && d_prob_train && d_key1 && (mode != SLAM_MATCH_KNN2 || d_key2),
                 "%s: null pointer", who);
    SLAM_REQUIRE(((uintptr_t)d_desc & 15) == 0, "%s: descriptors must be 16-byte aligned", who);
    MatchBatch mb{};
    mb.desc = (const uint4 *)d_desc;
    mb.count = d_count;
    mb.prob_a = d_prob_query;
    mb.prob_b = d_prob_train;
    mb.order = d_order;
    mb.n_max = n_max;
    mb.n_prob = n_prob;
    return run_match(who, mb, n_max, n_max, norm, mode, mode == SLAM_MATCH_KNN2 ? ratio : 0.0, d_key1, d_key2, n_max, n_max,
                     (unsigned *)d_workspace, workspace_bytes, st);
}

extern "C" int slam_unpack_keys(const uint32_t *d_key, int n, int32_t *d_index, int32_t *d_distance, void *stream) {
    if (n <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_key && d_index && d_distance, "slam_unpack_keys: null pointer");
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(slam_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_key, n,
                       d_index, d_distance);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}
