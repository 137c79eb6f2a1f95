// Host replay of the reference's adaptive RANSAC loop (utils/utills.py:474-530).
#pragma once
#include "common.h"

struct RansacParams {
    double prob = 0.9999;
    double max_iterations = 1e3;
    int fixed_iterations = 0;
    int max_draws = 64000;
};

struct RansacReplay {
    int it = 0;               // counted iterations (first_loop_iter)
    int draws = 0;            // draws consumed from the stream, failed ones included
    long long inl = 0, outl = 0;
    double outliers_perc = 0.99;
    int max_supp = -1;
    int best_draw = -1;
    bool finished = false;
    bool hit_cap = false;
    double est_for = -1.0, est = 0.0;   // the loop's iteration estimate, cached for the outlier ratio it was computed from

    bool wants_more(const RansacParams &p) const;
    // draws a later round should evaluate for this pair so that the loop very likely finishes inside it: the iterations
    // the loop still asks for (its current estimate, capped like the loop itself), scaled by the failed-solve rate seen
    // so far, plus slack.  Only a batch size: consume() stops where the reference's loop stops, whatever was evaluated.
    int draws_hint(const RansacParams &p) const;
    void consume(const RansacParams &p, int n_points, const uint8_t *status, const int32_t *count, int draw_begin,
                 int draw_end);
};
