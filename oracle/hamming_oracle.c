/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the brute-force descriptor matcher behind
 * utils/utills.py:57-70 (bf_matching -> cv2.BFMatcher(NORM_HAMMING2,
 * crossCheck=True).match) and :92-102 (knnMatch, k = 2).
 *
 * PARITY UNPINNED: the arithmetic lives in OpenCV (core/src/batch_distance.cpp,
 * core/src/norm.cpp), which is neither vendored nor version-pinned by the
 * reference and is not installed here.  The functions below restate its
 * published algorithm:
 *   - normHamming(cellSize = 2): per byte of a^b, the number of non-zero 2-bit
 *     cells (popCountTable2); cellSize = 1 is the plain bit count.
 *   - BatchDistInvoker, K > 0: per src1 row, scan src2 rows ascending, insert
 *     on strict `d < dist[K-1]`, shifting entries with `dist[k] > d`.
 *   - batchDistance(crosscheck = true): nearest src1 row for every src2 row,
 *     then a scan over src2 rows ascending that keeps, per src1 row, the
 *     smallest distance (strict `<`).
 * Build: see oracle/Makefile.
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORACLE_NORM_HAMMING = 6, ORACLE_NORM_HAMMING2 = 7 };

static uint8_t tab1[256], tab2[256];
static int tab_ready = 0;

static void init_tabs(void) {
    if (tab_ready) return;
    for (int v = 0; v < 256; ++v) {
        int c1 = 0, c2 = 0;
        for (int b = 0; b < 8; ++b) c1 += (v >> b) & 1;
        for (int c = 0; c < 4; ++c) c2 += ((v >> (2 * c)) & 3) != 0;
        tab1[v] = (uint8_t)c1;
        tab2[v] = (uint8_t)c2;
    }
    tab_ready = 1;
}

static inline int dist_bytes(const uint8_t *a, const uint8_t *b, int n, const uint8_t *tab) {
    int s = 0;
    for (int i = 0; i < n; ++i) s += tab[a[i] ^ b[i]];
    return s;
}

/* D[i*nb + j] = distance(a_i, b_j) */
void oracle_dist_matrix(const uint8_t *a, int na, int stride_a, const uint8_t *b, int nb, int stride_b,
                        int dbytes, int norm, int32_t *D) {
    init_tabs();
    const uint8_t *tab = norm == ORACLE_NORM_HAMMING2 ? tab2 : tab1;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j)
            D[(size_t)i * nb + j] = dist_bytes(a + (size_t)i * stride_a, b + (size_t)j * stride_b, dbytes, tab);
}

/* K nearest src2 rows for every src1 row (BatchDistInvoker).  idx/dist are [n1*K]. */
static void batch_knn(const uint8_t *s1, int n1, int st1, const uint8_t *s2, int n2, int st2, int dbytes,
                      const uint8_t *tab, int K, int32_t *idx, int32_t *dist) {
    for (int i = 0; i < n1; ++i) {
        int32_t *dp = dist + (size_t)i * K, *ip = idx + (size_t)i * K;
        for (int k = 0; k < K; ++k) { dp[k] = INT_MAX; ip[k] = -1; }
        const uint8_t *a = s1 + (size_t)i * st1;
        for (int j = 0; j < n2; ++j) {
            int d = dist_bytes(a, s2 + (size_t)j * st2, dbytes, tab);
            if (d < dp[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && dp[k] > d; --k) { ip[k + 1] = ip[k]; dp[k + 1] = dp[k]; }
                ip[k + 1] = j;
                dp[k + 1] = d;
            }
        }
    }
}

/* knnMatch(query, train, 2): idx/dist are [nq*2]; -1 / INT_MAX when nt < 2. */
void oracle_knn2(const uint8_t *q, int nq, int stq, const uint8_t *t, int nt, int stt, int dbytes, int norm,
                 int32_t *idx, int32_t *dist) {
    init_tabs();
    batch_knn(q, nq, stq, t, nt, stt, dbytes, norm == ORACLE_NORM_HAMMING2 ? tab2 : tab1, 2, idx, dist);
}

/* BFMatcher(crossCheck=True).match: out_train[i] = matched train row or -1, out_dist[i] its distance.
 * Returns the number of matches. */
int oracle_match_crosscheck(const uint8_t *q, int nq, int stq, const uint8_t *t, int nt, int stt, int dbytes,
                            int norm, int32_t *out_train, int32_t *out_dist) {
    init_tabs();
    const uint8_t *tab = norm == ORACLE_NORM_HAMMING2 ? tab2 : tab1;
    for (int i = 0; i < nq; ++i) { out_train[i] = -1; out_dist[i] = INT_MAX; }
    if (nq == 0 || nt == 0) return 0;
    int32_t *tidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
    int32_t *tdist = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
    batch_knn(t, nt, stt, q, nq, stq, dbytes, tab, 1, tidx, tdist);
    int count = 0;
    for (int j = 0; j < nt; ++j) {
        int i = tidx[j];
        if (tdist[j] < out_dist[i]) {
            if (out_train[i] < 0) ++count;
            out_dist[i] = tdist[j];
            out_train[i] = j;
        }
    }
    free(tidx);
    free(tdist);
    return count;
}

/* The other behaviour BFMatcher(crossCheck=True) has had (newer OpenCV computes both directions in
 * batchDistance and keeps train j for query tidx[j] only if sidx[tidx[j]] == j): mutual nearest rows,
 * first minimum on both sides.  UNPINNED like the rule above.  Same outputs as oracle_match_crosscheck. */
int oracle_match_crosscheck_mutual(const uint8_t *q, int nq, int stq, const uint8_t *t, int nt, int stt, int dbytes,
                                   int norm, int32_t *out_train, int32_t *out_dist) {
    init_tabs();
    const uint8_t *tab = norm == ORACLE_NORM_HAMMING2 ? tab2 : tab1;
    for (int i = 0; i < nq; ++i) { out_train[i] = -1; out_dist[i] = INT_MAX; }
    if (nq == 0 || nt == 0) return 0;
    int32_t *tidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
    int32_t *tdist = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
    int32_t *sidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nq);
    int32_t *sdist = (int32_t *)malloc(sizeof(int32_t) * (size_t)nq);
    batch_knn(t, nt, stt, q, nq, stq, dbytes, tab, 1, tidx, tdist);
    batch_knn(q, nq, stq, t, nt, stt, dbytes, tab, 1, sidx, sdist);
    int count = 0;
    for (int j = 0; j < nt; ++j) {
        int i = tidx[j];
        if (tdist[j] < out_dist[i] && sidx[i] == j) {
            if (out_train[i] < 0) ++count;
            out_dist[i] = tdist[j];
            out_train[i] = j;
        }
    }
    free(tidx); free(tdist); free(sidx); free(sdist);
    return count;
}

/* ---- word-parallel variant of the same matcher, used as the timed CPU baseline ("port") ----
 * Rows must be zero-padded to 64 bytes (16 dwords); identical results to the byte-table code above. */
static inline int dist_words(const uint64_t *a, const uint64_t *b, int norm2) {
    int s = 0;
    if (norm2) {
        for (int w = 0; w < 8; ++w) {
            uint64_t x = a[w] ^ b[w];
            s += __builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ULL);
        }
    } else {
        for (int w = 0; w < 8; ++w) s += __builtin_popcountll(a[w] ^ b[w]);
    }
    return s;
}

int oracle_match_crosscheck_w64(const uint8_t *q, int nq, const uint8_t *t, int nt, int norm,
                                int32_t *out_train, int32_t *out_dist) {
    const int norm2 = norm == ORACLE_NORM_HAMMING2;
    for (int i = 0; i < nq; ++i) { out_train[i] = -1; out_dist[i] = INT_MAX; }
    int count = 0;
    for (int j = 0; j < nt; ++j) {
        const uint64_t *b = (const uint64_t *)(t + (size_t)j * 64);
        int best = INT_MAX, bi = -1;
        for (int i = 0; i < nq; ++i) {
            int d = dist_words((const uint64_t *)(q + (size_t)i * 64), b, norm2);
            if (d < best) { best = d; bi = i; }
        }
        if (bi >= 0 && best < out_dist[bi]) {
            if (out_train[bi] < 0) ++count;
            out_dist[bi] = best;
            out_train[bi] = j;
        }
    }
    return count;
}
