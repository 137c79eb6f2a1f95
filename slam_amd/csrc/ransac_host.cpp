// Host side of the RANSAC loop: the reference's random stream and its adaptive stopping rule.
//
// utils/utills.py:502 draws np.random.randint(0, n, 4) from numpy's legacy global generator: MT19937
// seeded by init_genrand, 32-bit outputs, bounded by masked rejection (numpy/random/src/distributions:
// buffered_bounded_masked_uint32).  The host functions below serve the per-pair drop-in (online_est_pnp_ransac draws
// from the global np.random stream) and the tests; the shard pipeline generates the same stream on the device
// (p3p.hip, draw_samples_kernel), one wavefront per pair.
#include "ransac_host.h"

#include <cmath>

namespace {

struct MT19937 {
    uint32_t mt[624];
    int pos;
    explicit MT19937(uint32_t seed) {
        for (int i = 0; i < 624; ++i) {
            mt[i] = seed;
            seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
    }
    void twist() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
        int k = 0;
        for (; k < 624 - 397; ++k) {
            const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
            mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; k < 623; ++k) {
            const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
            mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        pos = 0;
    }
    inline uint32_t next() {
        if (pos >= 624) twist();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

}  // namespace

static void slam_mt_seed_words(uint32_t seed, uint32_t *out, int n_words) {
    MT19937 g(seed);
    for (int i = 0; i < n_words; ++i) out[i] = g.next();
}

extern "C" int slam_mt19937_words(uint32_t seed, uint32_t *h_words, int n_words) {
    if (n_words < 0 || (!h_words && n_words)) {
        slam_set_error("slam_mt19937_words: bad arguments");
        return SLAM_E_INVALID;
    }
    slam_mt_seed_words(seed, h_words, n_words);
    return SLAM_OK;
}

extern "C" long slam_randint_host(uint32_t seed, uint32_t bound, int32_t *h_out, long n_out) {
    if (bound == 0) {
        slam_set_error("slam_randint_host: low >= high");   // numpy raises ValueError
        return SLAM_E_EMPTY;
    }
    if (n_out < 0 || (!h_out && n_out)) {
        slam_set_error("slam_randint_host: bad arguments");
        return SLAM_E_INVALID;
    }
    const uint32_t rng = bound - 1u;
    if (rng == 0) {
        for (long i = 0; i < n_out; ++i) h_out[i] = 0;
        return 0;
    }
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    MT19937 g(seed);
    long used = 0;
    for (long i = 0; i < n_out; ++i) {
        uint32_t v;
        do {
            v = g.next() & mask;
            ++used;
        } while (v > rng);
        h_out[i] = (int32_t)v;
    }
    return used;
}

// ---- the adaptive loop of online_est_pnp_ransac (utils/utills.py:488-523), replayed over batched results ----
bool RansacReplay::wants_more(const RansacParams &p) const {
    if (p.fixed_iterations > 0) return it < p.fixed_iterations;
    if (outliers_perc == 0.0) return false;
    // first_loop_iter_est(prob, outliers_perc) = log(1 - prob) / log(1 - (1 - outliers_perc) ** model_parm_num)
    const double est = std::log(1.0 - p.prob) / std::log(1.0 - std::pow(1.0 - outliers_perc, 4.0));
    return (double)it < est && (double)it < p.max_iterations;
}

int RansacReplay::draws_hint(const RansacParams &p) const {
    double rem;
    if (p.fixed_iterations > 0) {
        rem = (double)p.fixed_iterations - it;
    } else {
        // a cached estimate at or above the cap for a ratio not above the current one (the estimate grows with the ratio)
        // gives the same minimum without log / pow: pairs of unrelated frames, thousands per loop-closure pass
        const bool capped = est_for > 0.0 && outliers_perc >= est_for && est * (1.0 - 1e-9) >= p.max_iterations;
        const double e = capped ? p.max_iterations : std::log(1.0 - p.prob) / std::log(1.0 - std::pow(1.0 - outliers_perc, 4.0));
        rem = (e < p.max_iterations ? e : p.max_iterations) - it;
    }
    if (!(rem > 1.0)) rem = 1.0;
    const double ok_rate = draws > 0 ? (double)it / (double)draws : 1.0;
    double d = rem / (ok_rate > 0.05 ? ok_rate : 0.05) * 1.1 + 32.0;
    const double left = (double)p.max_draws - draws;
    if (d > left) d = left;
    if (!(d >= 1.0)) d = 1.0;
    return d > 1e9 ? 1000000000 : (int)d;
}

void RansacReplay::consume(const RansacParams &p, int n_points, const uint8_t *status, const int32_t *count,
                           int draw_begin, int draw_end) {
    // status/count are indexed by (draw - draw_begin)
    while (!finished) {
        // wants_more(), with the log/pow of the estimate recomputed only when the outlier ratio changed (pairs stuck at
        // the 0.99 clip run hundreds of iterations on one value)
        bool more = false;
        if (p.fixed_iterations > 0) {
            more = it < p.fixed_iterations;
        } else if (outliers_perc == 0.0) {
            more = false;
        } else {
            // The estimate grows with the outlier ratio, so the value cached for a ratio r0 settles the test for every
            // ratio on the far side of r0 from the answer: r >= r0 and it < est(r0) -> more; r <= r0 and it >= est(r0)
            // -> no more (a margin of 1e-9 relative against the rounding of log / pow, which is ~1e-16).  A pair stuck
            // near one ratio -- unrelated frames run a thousand iterations around 0.93 -- then recomputes log and pow
            // only when its ratio reaches a new minimum, not at every iteration.  Same decisions, bit for bit.
            bool known = false;
            if (est_for > 0.0) {
                if (outliers_perc >= est_for && (double)it < est * (1.0 - 1e-9)) {
                    more = (double)it < p.max_iterations;
                    known = true;
                } else if (outliers_perc <= est_for && (double)it >= est * (1.0 + 1e-9)) {
                    more = false;
                    known = true;
                }
            }
            if (!known) {
                if (est_for != outliers_perc) {
                    est = std::log(1.0 - p.prob) / std::log(1.0 - std::pow(1.0 - outliers_perc, 4.0));
                    est_for = outliers_perc;
                }
                more = (double)it < est && (double)it < p.max_iterations;
            }
        }
        if (!more) {
            finished = true;
            break;
        }
        if (draws >= p.max_draws) {
            finished = true;
            hit_cap = true;
            break;
        }
        if (draws >= draw_end) break;            // need another round
        const int d = draws++;
        if (!status[d - draw_begin]) continue;   // utills.py:506-507: failed PnP is not an iteration
        const int num = count[d - draw_begin];
        if (num > max_supp) {                    // strict: the first best wins (utills.py:514)
            max_supp = num;
            best_draw = d;
        }
        ++it;
        outl += n_points - num;
        inl += num;
        const double perc = (double)outl / (double)(inl + outl);
        outliers_perc = perc < 0.99 ? perc : 0.99;
    }
}

// the same replay behind the C ABI, for callers that evaluate hypotheses in batches themselves (the per-pair drop-in)
extern "C" int slam_ransac_state_init(SlamRansacState *st) {
    if (!st) {
        slam_set_error("slam_ransac_state_init: null state");
        return SLAM_E_INVALID;
    }
    const RansacReplay r;
    st->it = r.it, st->draws = r.draws, st->max_supp = r.max_supp, st->best_draw = r.best_draw;
    st->finished = 0, st->hit_cap = 0;
    st->inliers = r.inl, st->outliers = r.outl;
    st->outliers_perc = r.outliers_perc, st->est_for = r.est_for, st->est = r.est;
    return SLAM_OK;
}

extern "C" int slam_ransac_replay(SlamRansacState *st, int n_points, const uint8_t *h_status, const int32_t *h_count,
                                  int draw_begin, int draw_end, double prob, double max_iterations, int fixed_iterations,
                                  int max_draws) {
    if (!st || draw_end < draw_begin || (draw_end > draw_begin && (!h_status || !h_count)) || st->draws < draw_begin) {
        slam_set_error("slam_ransac_replay: bad arguments (the batch must start at or before the draws consumed so far)");
        return SLAM_E_INVALID;
    }
    RansacReplay r;
    r.it = st->it, r.draws = st->draws, r.max_supp = st->max_supp, r.best_draw = st->best_draw;
    r.finished = st->finished != 0, r.hit_cap = st->hit_cap != 0;
    r.inl = st->inliers, r.outl = st->outliers;
    r.outliers_perc = st->outliers_perc, r.est_for = st->est_for, r.est = st->est;
    RansacParams p;
    p.prob = prob, p.max_iterations = max_iterations, p.fixed_iterations = fixed_iterations;
    p.max_draws = max_draws > 0 ? max_draws : 0x7fffffff;
    r.consume(p, n_points, h_status, h_count, draw_begin, draw_end);
    st->it = r.it, st->draws = r.draws, st->max_supp = r.max_supp, st->best_draw = r.best_draw;
    st->finished = r.finished, st->hit_cap = r.hit_cap;
    st->inliers = r.inl, st->outliers = r.outl;
    st->outliers_perc = r.outliers_perc, st->est_for = r.est_for, st->est = r.est;
    return SLAM_OK;
}

// ---- convert_trans_from_rel_to_global (utils/utills.py:749-757): running composition T_i @ [T_{i-1..0}; 0 0 0 1] ----
// numpy evaluates `second @ hom(first)` through BLAS dgemm: k ascending, fused multiply-adds from a zero accumulator.
extern "C" int slam_compose_global(const double *h_rel, int n, double *h_global) {
    if (n < 0 || (n && (!h_rel || !h_global))) {
        slam_set_error("slam_compose_global: bad arguments");
        return SLAM_E_INVALID;
    }
    double last[12];
    for (int k = 0; k < 12; ++k) last[k] = n ? h_rel[k] : 0.0;
    for (int i = 0; i < n; ++i) {
        const double *T = h_rel + (size_t)i * 12;
        double out[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                const double hom3 = c == 3 ? 1.0 : 0.0;
                out[r * 4 + c] = std::fma(T[r * 4 + 3], hom3,
                                          std::fma(T[r * 4 + 2], last[8 + c], std::fma(T[r * 4 + 1], last[4 + c], T[r * 4] * last[c])));
            }
        for (int k = 0; k < 12; ++k) h_global[(size_t)i * 12 + k] = last[k] = out[k];
    }
    return SLAM_OK;
}
