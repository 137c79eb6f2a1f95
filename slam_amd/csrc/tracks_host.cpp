// Track chaining on index arrays (host): the immediate consumer of the hot path's output.
//
// Replaces DataBase.create_db / find_tracks_between_consecutive_frame (DataBaseDirectory/DataBase.py:56-153) of the
// reference: for every supporter of pair (i, i+1), in order, the frame-i LEFT key-point index is looked up in the
// dictionary left by pair (i-1, i); a hit extends that track, a miss opens track number len(tracks).  The reference
// does that with a Python dict of Feature objects per pair; here the dictionary is a dense array indexed by key-point
// with a generation stamp (no clearing between pairs), and the result is a CSR table of observations, which is what
// slam_amd/types.py:ArrayDataBase serves the reference's accessors from.  One pass to number the tracks, one to place
// the observations (a track's observations are created in ascending frame order, so a stable bucket fill by track id
// is the (track, frame) order).
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.h"
#include "host_pool.h"

extern "C" int slam_chain_tracks(const int32_t *h_kp, const int64_t *h_pair_ptr, int32_t n_pairs, int32_t n_kp,
                                 const float *h_xy, int32_t n_max, int64_t *h_track_ptr, int32_t *h_obs_frame,
                                 int32_t *h_obs_kp, double *h_obs_xy, int64_t *h_frame_ptr, int64_t *h_frame_tracks,
                                 int64_t *h_n_tracks, int64_t *h_n_obs) try {
    if (n_pairs < 0 || n_kp < 0 || !h_pair_ptr || !h_track_ptr || !h_frame_ptr || !h_n_tracks || !h_n_obs) {
        slam_set_error("slam_chain_tracks: bad arguments");
        return SLAM_E_INVALID;
    }
    const int64_t M = h_pair_ptr[n_pairs];
    if (h_pair_ptr[0] != 0 || M < 0 || M > INT32_MAX || (M && (!h_kp || !h_obs_frame || !h_obs_kp || !h_frame_tracks))) {
        slam_set_error("slam_chain_tracks: bad arguments");
        return SLAM_E_INVALID;
    }
    if (h_xy && (!h_obs_xy || n_max < n_kp)) {
        slam_set_error("slam_chain_tracks: coordinates need an output array and n_max >= n_kp");
        return SLAM_E_INVALID;
    }
    for (int i = 0; i < n_pairs; ++i)
        if (h_pair_ptr[i + 1] < h_pair_ptr[i]) {
            slam_set_error("slam_chain_tracks: pair_ptr is not ascending at pair %d", i);
            return SLAM_E_INVALID;
        }
    // carried[k]: the track that frame i's left key-point k carries (left there by pair i-1), valid if its stamp is i
    std::vector<int32_t> stamp_a(n_kp, -1), stamp_b(n_kp, -1), seen0(n_kp, -1);
    std::vector<int64_t> tid_a(n_kp), tid_b(n_kp);
    std::vector<int64_t> tid(M);                 // track of every supporter; bit 62 marks "opened here"
    std::vector<int64_t> n_new(n_pairs + 1, 0);
    const int64_t NEW = (int64_t)1 << 62;
    int64_t n_tracks = 0;
    h_track_ptr[0] = 0;
    int32_t *sa = stamp_a.data(), *sb = stamp_b.data();
    int64_t *ta = tid_a.data(), *tb = tid_b.data();
    for (int i = 0; i < n_pairs; ++i) {
        for (int64_t s = h_pair_ptr[i]; s < h_pair_ptr[i + 1]; ++s) {
            const int32_t *k = h_kp + s * 4;
            if (k[0] < 0 || k[0] >= n_kp || k[2] < 0 || k[2] >= n_kp || k[1] < 0 || k[3] < 0 ||
                (h_xy && (k[1] >= n_max || k[3] >= n_max))) {
                slam_set_error("slam_chain_tracks: key-point index out of range in pair %d", i);
                return SLAM_E_INVALID;
            }
            if (seen0[k[0]] == i || sb[k[2]] == i + 1) {
                // the reference's dict would overwrite; the array rule does not reproduce that (tracks.pairs_are_chainable)
                slam_set_error("slam_chain_tracks: duplicate key-points inside pair %d", i);
                return SLAM_E_DUPLICATE;
            }
            seen0[k[0]] = i;
            int64_t t;
            if (sa[k[0]] == i) {
                t = ta[k[0]];
                h_track_ptr[t + 1] += 1;
                tid[s] = t;
            } else {
                t = n_tracks++;
                h_track_ptr[t + 1] = 2;          // frame i and frame i+1
                tid[s] = t | NEW;
                n_new[i] += 1;
            }
            sb[k[2]] = i + 1;
            tb[k[2]] = t;
        }
        int32_t *ts = sa; sa = sb; sb = ts;
        int64_t *tt = ta; ta = tb; tb = tt;
    }
    for (int64_t t = 0; t < n_tracks; ++t) h_track_ptr[t + 1] += h_track_ptr[t];
    const int64_t n_obs = n_tracks ? h_track_ptr[n_tracks] : 0;
    // frame f lists the tracks of pair f-1's supporters (in order), then the tracks pair f opens (DataBase.py:120-150)
    h_frame_ptr[0] = 0;
    for (int f = 0; f <= n_pairs; ++f) {
        const int64_t from_prev = f ? h_pair_ptr[f] - h_pair_ptr[f - 1] : 0;
        h_frame_ptr[f + 1] = h_frame_ptr[f] + from_prev + (f < n_pairs ? n_new[f] : 0);
    }
    for (int i = 0; i < n_pairs; ++i) {
        int64_t *ft_next = h_frame_tracks + h_frame_ptr[i + 1];                                            // frame i+1, first part
        int64_t *ft_new = h_frame_tracks + h_frame_ptr[i] + (i ? h_pair_ptr[i] - h_pair_ptr[i - 1] : 0);   // frame i, second part
        for (int64_t s = h_pair_ptr[i]; s < h_pair_ptr[i + 1]; ++s) {
            if (tid[s] & NEW) *ft_new++ = tid[s] & ~NEW;
            *ft_next++ = tid[s] & ~NEW;
        }
    }
    // Placement is a scatter (one cache miss per observation and output array), so it is split over the host pool by
    // track range: every worker walks all supporters in creation order and places those of its own tracks.
    std::vector<int64_t> fill(h_track_ptr, h_track_ptr + n_tracks);
    const size_t fstride = (size_t)2 * n_max * 2;      // floats per frame of h_xy: [2][n_max][2]
    auto place = [&](int64_t t, int frame, int32_t kl, int32_t kr) {
        const int64_t row = fill[t]++;
        h_obs_frame[row] = frame;
        h_obs_kp[row * 2] = kl;
        h_obs_kp[row * 2 + 1] = kr;
        if (h_xy) {
            const float *l = h_xy + frame * fstride + (size_t)kl * 2;
            const float *r = h_xy + frame * fstride + (size_t)n_max * 2 + (size_t)kr * 2;
            double *o = h_obs_xy + row * 4;          // Feature's constructor order: x_left, x_right, y_left, y_right
            o[0] = l[0]; o[1] = r[0]; o[2] = l[1]; o[3] = r[1];
        }
    };
    parallel_for(HostPool::get(), (int)n_tracks, 1 << 16, [&](int t_lo, int t_hi) {
        for (int i = 0; i < n_pairs; ++i)
            for (int64_t s = h_pair_ptr[i]; s < h_pair_ptr[i + 1]; ++s) {
                const int64_t t = tid[s] & ~NEW;
                if (t < t_lo || t >= t_hi) continue;
                const int32_t *k = h_kp + s * 4;
                if (tid[s] & NEW) place(t, i, k[0], k[1]);
                place(t, i + 1, k[2], k[3]);
            }
    });
    *h_n_tracks = n_tracks;
    *h_n_obs = n_obs;
    return SLAM_OK;
}
SLAM_CATCH_ALL("slam_chain_tracks")
