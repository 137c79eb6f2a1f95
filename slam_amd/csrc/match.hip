// All-pairs binary-descriptor matching for gfx950.
//
// Replaces cv2.BFMatcher(NORM_HAMMING2, crossCheck=True).match and .knnMatch(k=2) as called from
// utils/utills.py:57-70, :92-102, :105-119 of the reference.
//
// Shape of the work: integer VALU (xor / or / popcount), not a GEMM and not HBM bound
// (2000 x 2000 rows x 64 B = 256 KB in, 35 lane-ops per row pair).  Mapping:
//   - every lane owns ROWS "A" rows in registers (64 VGPRs), pre-transformed so that a
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
//   - the 32 distance chains of a B row (4 rows x 8 words) are issued side by side, one step of all of them at a
//     time (32 x v_xor, 32 x v_bitop3, 32 x v_bcnt): the order matters by 15 % on this part.
//   - tile shape (4 rows/lane, 256 threads, 128-row B tiles) chosen by a sweep on the MI355X.
//   - cross-check (OpenCV batchDistance(crosscheck=true)): A = train rows; the nearest query of
//     every train row is scattered with atomicMin on key (distance << 20 | train row) into the
//     query's slot.  min is order independent, so the result is deterministic.
#include <mutex>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

#ifndef SLAM_ROWS
#define SLAM_ROWS 4
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
    unsigned *out1, *out2;    // per problem stride out_stride (top-1 / top-2 keys, or cross-check slots)
    int out_stride;
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

// FUSED: cross-check scatter straight from the all-pairs kernel (nsplit == 1).
// TOP2: keep the second best too (kNN mode); cross-check needs only the nearest row.
// RW = A rows per lane: ROWS (4) where the grid fills the chip, 2 for a batch of few problems (twice as many
// wavefronts for the same work, see launch_allpairs).
template <bool NORM2, bool FUSED, bool TOP2, int RW>
__global__ __launch_bounds__(BLOCK) void allpairs_top2_kernel(MatchBatch mb) {
    __shared__ uint4 tile[2][TILE_B * 4];
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
        na = mb.count[ia];
        nb = mb.count[ib];
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
    const int tiles_per_split = (tiles_b + mb.nsplit - 1) / mb.nsplit;
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
            constexpr int JU = SLAM_JUNROLL;
            int j = 0;
            for (; j + JU <= rows_here; j += JU) {
                unsigned d[JU][ROWS];
                bool improves = false;
#pragma unroll
                for (int u = 0; u < JU; ++u) improves |= dist_row(j + u, d[u]);
                if (__any(improves)) {
#pragma unroll
                    for (int u = 0; u < JU; ++u) insert_row(j + u, d[u]);
                }
            }
            for (; j < rows_here; ++j) {
                unsigned d[ROWS];
                if (__any(dist_row(j, d))) insert_row(j, d);
            }
        };
#if SLAM_ROW_VARIANTS
        dispatch_rows<ROWS>(nact, scan_tile);
#else
        scan_tile(std::integral_constant<int, ROWS>{});
#endif
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = slot_row(r);
        if (row >= na) continue;
        if (FUSED) {
            // A = train rows, B = query rows: scatter (distance, train row) into the nearest query's slot
            if (m1[r] != NONE) {
                const unsigned q = m1[r] & (SLAM_MAX_ROWS - 1);
                const unsigned key = (m1[r] & ~(unsigned)(SLAM_MAX_ROWS - 1)) | (unsigned)row;
                atomicMin(mb.out1 + (size_t)p * mb.out_stride + q, key);
            }
        } else {
            const size_t o = (size_t)p * mb.out_stride + (size_t)split * na + row;
            mb.out1[o] = m1[r];
            if (mb.out2) mb.out2[o] = m2[r];
        }
    }
}

// merges per-split partial top-2 lists of one problem; mode cross-check scatters, mode knn2 writes keys
__global__ void merge_splits_kernel(const unsigned *part1, const unsigned *part2, int na, int nsplit, int crosscheck,
                                    unsigned *out1, unsigned *out2) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= na) return;
    unsigned m1 = NONE, m2 = NONE;
    for (int s = 0; s < nsplit; ++s) {
        top2_insert(m1, m2, part1[(size_t)s * na + row]);
        if (part2) top2_insert(m1, m2, part2[(size_t)s * na + row]);
    }
    if (crosscheck) {
        if (m1 != NONE) {
            const unsigned q = m1 & (SLAM_MAX_ROWS - 1);
            atomicMin(out1 + q, (m1 & ~(unsigned)(SLAM_MAX_ROWS - 1)) | (unsigned)row);
        }
    } else {
        out1[row] = m1;
        if (out2) out2[row] = m2;
    }
}

__global__ void unpack_keys_kernel(const unsigned *key, int n, int *index, int *distance) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = key[i];
    index[i] = k == NONE ? -1 : (int)(k & (SLAM_MAX_ROWS - 1));
    distance[i] = k == NONE ? INT32_MAX : (int)(k >> SLAM_KEY_SHIFT);
}

// Workgroups the chip holds at once: 256 CUs x 3 (139 VGPRs = 3 wavefronts per SIMD, 4 wavefronts per workgroup).
constexpr unsigned CHIP_WORKGROUPS = 768;
constexpr int ROWS_SMALL = 2;

template <bool FUSED, int RW>
void launch_allpairs_rows(MatchBatch mb, int a_rows, int norm, hipStream_t st) {
    mb.tiles_per_prob = slam_div_up(a_rows, RW * BLOCK);
    const unsigned total = (unsigned)mb.tiles_per_prob * (unsigned)mb.n_prob;
    dim3 grid((total + 7u) / 8u * 8u, mb.nsplit);
    const bool top2 = mb.out2 != nullptr;
    if (norm == SLAM_NORM_HAMMING2) {
        if (top2)
            hipLaunchKernelGGL((allpairs_top2_kernel<true, false, true, RW>), grid, dim3(BLOCK), 0, st, mb);
        else
            hipLaunchKernelGGL((allpairs_top2_kernel<true, FUSED, false, RW>), grid, dim3(BLOCK), 0, st, mb);
    } else {
        if (top2)
            hipLaunchKernelGGL((allpairs_top2_kernel<false, false, true, RW>), grid, dim3(BLOCK), 0, st, mb);
        else
            hipLaunchKernelGGL((allpairs_top2_kernel<false, FUSED, false, RW>), grid, dim3(BLOCK), 0, st, mb);
    }
}

// a_rows = A rows per problem slot (the tile grid covers them).  A batch of few problems (the stress configuration:
// 59 problems of 8k x 8k rows = 472 workgroups at 4 rows per lane) leaves a third of the chip's wavefront slots empty;
// it runs with 2 rows per lane instead: twice the wavefronts for the same distances.  At full load 4 rows per lane is
// 9 % faster (DESIGN.md tile sweep), so that stays the choice whenever the grid fills the chip.
template <bool FUSED>
void launch_allpairs(const MatchBatch &mb, int a_rows, int norm, hipStream_t st) {
    const unsigned total4 = (unsigned)slam_div_up(a_rows, TILE_A) * (unsigned)mb.n_prob * (unsigned)mb.nsplit;
    if (ROWS > ROWS_SMALL && total4 < CHIP_WORKGROUPS)
        launch_allpairs_rows<FUSED, ROWS_SMALL>(mb, a_rows, norm, st);
    else
        launch_allpairs_rows<FUSED, ROWS>(mb, a_rows, norm, st);
}

int pick_nsplit(int na, int nb) {
    const int tiles_a = slam_div_up(na, TILE_A), tiles_b = slam_div_up(nb, TILE_B);
    int ns = 1024 / (tiles_a > 0 ? tiles_a : 1);
    if (ns > tiles_b) ns = tiles_b;
    if (ns > 64) ns = 64;
    return ns < 1 ? 1 : ns;
}

}  // namespace

extern "C" size_t slam_match_workspace_bytes(int n_query, int n_train) {
    const size_t n = (size_t)(n_query > n_train ? n_query : n_train);
    return 2 * 64 * n * sizeof(unsigned) + 256;
}

namespace {

// One problem: B split over blockIdx.y into per-split partial lists (`part`, 2 * nsplit * na keys), then merged.
// Every key of the partial lists the merge reads is written by the all-pairs kernel first (rows < na of every split,
// empty splits write SLAM_KEY_NONE), so `part` needs no clearing.
int match_single(const char *who, const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train, int norm,
                 int mode, uint32_t *d_key1, uint32_t *d_key2, unsigned *part, size_t part_bytes, hipStream_t st) {
    const bool cross = mode == SLAM_MATCH_CROSSCHECK;
    SLAM_HIP_CHECK(hipMemsetAsync(d_key1, 0xFF, sizeof(uint32_t) * (size_t)n_query, st));
    if (!cross) SLAM_HIP_CHECK(hipMemsetAsync(d_key2, 0xFF, sizeof(uint32_t) * (size_t)n_query, st));
    if (n_train == 0) return SLAM_OK;

    MatchBatch mb{};
    // cross-check: lanes own TRAIN rows and scan the queries; knn: lanes own QUERY rows and scan the trains
    mb.a_ptr = (const uint4 *)(cross ? d_train : d_query);
    mb.b_ptr = (const uint4 *)(cross ? d_query : d_train);
    mb.na = cross ? n_train : n_query;
    mb.nb = cross ? n_query : n_train;
    mb.n_prob = 1;
    mb.nsplit = pick_nsplit(mb.na, mb.nb);
    const size_t need = 2 * (size_t)mb.nsplit * mb.na * sizeof(unsigned);
    if (part_bytes < need || !part) {
        slam_set_error("%s: workspace %zu B < %zu B", who, part_bytes, need);
        return SLAM_E_WORKSPACE;
    }
    unsigned *part1 = part;
    unsigned *part2 = part1 + (size_t)mb.nsplit * mb.na;
    mb.out1 = part1;
    mb.out2 = cross ? nullptr : part2;
    mb.out_stride = 0;
    launch_allpairs_rows<false, ROWS>(mb, mb.na, norm, st);      // B is split until the grid fills the chip (pick_nsplit)
    SLAM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(merge_splits_kernel, dim3(slam_div_up(mb.na, 256)), dim3(256), 0, st, part1, mb.out2, mb.na,
                       mb.nsplit, cross ? 1 : 0, d_key1, d_key2);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

int match_single_check_args(const char *who, const uint8_t *d_query, int n_query, const uint8_t *d_train, int n_train,
                            int norm, int mode, const uint32_t *d_key1, const uint32_t *d_key2, int max_rows) {
    SLAM_REQUIRE(norm == SLAM_NORM_HAMMING || norm == SLAM_NORM_HAMMING2, "%s: norm %d unsupported", who, norm);
    SLAM_REQUIRE(mode == SLAM_MATCH_CROSSCHECK || mode == SLAM_MATCH_KNN2, "%s: mode %d unsupported", who, mode);
    SLAM_REQUIRE(n_query >= 0 && n_train >= 0 && n_query < max_rows && n_train < max_rows,
                 "%s: row counts %d/%d out of range (each must be below %d)", who, n_query, n_train, max_rows);
    if (n_query == 0) return SLAM_OK;
    SLAM_REQUIRE(d_key1 && (mode == SLAM_MATCH_CROSSCHECK || d_key2), "%s: null output", who);
    SLAM_REQUIRE(d_query && (d_train || n_train == 0), "%s: null input", who);
    SLAM_REQUIRE(((uintptr_t)d_query & 15) == 0 && ((uintptr_t)d_train & 15) == 0,
                 "%s: descriptor rows must be 16-byte aligned", who);
    return SLAM_OK;
}

}  // namespace

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
// workspace -- live in a file-scope device array instead: 2 lists x 64 splits x SLAM_MATCH_NOALLOC_MAX_ROWS keys
// = 4 MiB, held by every GPU the library runs on until the process ends.  In host code the array's NAME is only a
// stand-in for the symbol; its address on a device comes from hipGetSymbolAddress, asked once per device with that
// device current, and only that pointer is handed to kernels.
constexpr size_t NOALLOC_WORDS = 2 * (size_t)64 * SLAM_MATCH_NOALLOC_MAX_ROWS;
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

extern "C" int slam_match_hamming_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max,
                                          const int32_t *d_prob_query, const int32_t *d_prob_train, int n_prob,
                                          int norm, int mode, uint32_t *d_key1, uint32_t *d_key2, void *stream) {
    return slam_launch_match_batched(d_desc, d_count, n_max, d_prob_query, d_prob_train, nullptr, n_prob, norm, mode,
                                     d_key1, d_key2, (hipStream_t)stream);
}

int slam_launch_match_batched(const uint8_t *d_desc, const int32_t *d_count, int n_max, const int32_t *d_prob_query,
                              const int32_t *d_prob_train, const int32_t *d_order, int n_prob, int norm, int mode,
                              uint32_t *d_key1, uint32_t *d_key2, hipStream_t st) {
    SLAM_REQUIRE(norm == SLAM_NORM_HAMMING || norm == SLAM_NORM_HAMMING2, "slam_match_hamming_batched: norm %d", norm);
    SLAM_REQUIRE(mode == SLAM_MATCH_CROSSCHECK || mode == SLAM_MATCH_KNN2, "slam_match_hamming_batched: mode %d", mode);
    SLAM_REQUIRE(n_max > 0 && n_max < SLAM_MAX_ROWS && n_prob >= 0, "slam_match_hamming_batched: bad sizes");
    if (n_prob == 0) return SLAM_OK;
    SLAM_REQUIRE(d_desc && d_count && d_prob_query && d_prob_train && d_key1 && (mode == SLAM_MATCH_CROSSCHECK || d_key2),
                 "slam_match_hamming_batched: null pointer");
    SLAM_REQUIRE(((uintptr_t)d_desc & 15) == 0, "slam_match_hamming_batched: descriptors must be 16-byte aligned");
    const bool cross = mode == SLAM_MATCH_CROSSCHECK;
    SLAM_HIP_CHECK(hipMemsetAsync(d_key1, 0xFF, sizeof(uint32_t) * (size_t)n_prob * n_max, st));
    if (!cross) SLAM_HIP_CHECK(hipMemsetAsync(d_key2, 0xFF, sizeof(uint32_t) * (size_t)n_prob * n_max, st));
    MatchBatch mb{};
    mb.desc = (const uint4 *)d_desc;
    mb.count = d_count;
    mb.prob_a = cross ? d_prob_train : d_prob_query;
    mb.prob_b = cross ? d_prob_query : d_prob_train;
    mb.order = d_order;
    mb.n_max = n_max;
    mb.n_prob = n_prob;
    mb.nsplit = 1;
    mb.out1 = d_key1;
    mb.out2 = cross ? nullptr : d_key2;
    mb.out_stride = n_max;
    if (cross)
        launch_allpairs<true>(mb, n_max, norm, st);
    else
        launch_allpairs<false>(mb, n_max, norm, st);   // nsplit == 1: partial slot 0 is the final answer
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}

extern "C" int slam_unpack_keys(const uint32_t *d_key, int n, int32_t *d_index, int32_t *d_distance, void *stream) {
    if (n <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_key && d_index && d_distance, "slam_unpack_keys: null pointer");
    hipLaunchKernelGGL(unpack_keys_kernel, dim3(slam_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_key, n,
                       d_index, d_distance);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}
