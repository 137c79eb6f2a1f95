// AddressSanitizer + UBSan run of the HOST-side entry points of the C ABI (built and run by
// tests/test_host_sanitizers_cpu.py with `hipcc -Xarch_host -fsanitize=...`; no GPU call is made): every output buffer
// is a heap block of exactly the documented size, so a write past it stops the run, and the results are compared with
// plain restatements written here (std::map for the reference's per-pair dictionaries, DataBase.py:56-153; the replay
// of utils/utills.py:488-523 in one go against round-by-round consumption).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../include/slam_amd.h"
#include "../slam_amd/csrc/ransac_host.h"

static long bad = 0;
#define EXPECT(cond)                                                      \
    do {                                                                  \
        if (!(cond)) {                                                    \
            ++bad;                                                        \
            std::fprintf(stderr, "line %d: %s\n", __LINE__, #cond);       \
        }                                                                 \
    } while (0)

// ---- slam_chain_tracks against the dictionary formulation ----
static void chain_case(std::mt19937 &rng, int n_pairs, int n_kp, int n_max, bool with_xy, int empty_every) {
    std::vector<int64_t> ptr(n_pairs + 1, 0);
    std::vector<int32_t> kp;
    std::vector<int> perm0(n_kp), perm2(n_kp);
    for (int i = 0; i < n_pairs; ++i) {
        int n = (empty_every && i % empty_every == empty_every - 1) ? 0 : (int)(rng() % (n_kp + 1));
        for (int k = 0; k < n_kp; ++k) perm0[k] = perm2[k] = k;
        std::shuffle(perm0.begin(), perm0.end(), rng);
        std::shuffle(perm2.begin(), perm2.end(), rng);
        for (int s = 0; s < n; ++s) {
            kp.push_back(perm0[s]);
            kp.push_back((int32_t)(rng() % n_max));
            kp.push_back(perm2[s]);
            kp.push_back((int32_t)(rng() % n_max));
        }
        ptr[i + 1] = ptr[i] + n;
    }
    const int64_t M = ptr[n_pairs];
    std::vector<float> xy;
    if (with_xy) {
        xy.resize((size_t)(n_pairs + 1) * 2 * n_max * 2);
        for (float &v : xy) v = (float)(rng() % 100000) / 64.0f;
    }
    // the reference's dictionaries: key-point of frame i -> track, rebuilt per pair
    std::map<int, int64_t> carried;
    std::vector<std::vector<std::array<int32_t, 3>>> want;          // per track: (frame, kl, kr)
    std::vector<std::vector<int64_t>> want_frame(n_pairs + 1);
    for (int i = 0; i < n_pairs; ++i) {
        std::map<int, int64_t> next;
        std::vector<int64_t> opened;
        for (int64_t s = ptr[i]; s < ptr[i + 1]; ++s) {
            const int32_t *k = kp.data() + s * 4;
            int64_t t;
            auto hit = carried.find(k[0]);
            if (hit != carried.end()) {
                t = hit->second;
            } else {
                t = (int64_t)want.size();
                want.push_back({{i, k[0], k[1]}});
                opened.push_back(t);
            }
            want[t].push_back({i + 1, k[2], k[3]});
            next[k[2]] = t;
            want_frame[i + 1].push_back(t);
        }
        want_frame[i].insert(want_frame[i].end(), opened.begin(), opened.end());
        carried.swap(next);
    }
    int64_t want_obs = 0;
    for (auto &t : want) want_obs += (int64_t)t.size();

    // outputs at their documented sizes (upper bounds known before the call: M tracks, 2 M observations)
    int64_t *track_ptr = (int64_t *)std::malloc(sizeof(int64_t) * (M + 1));
    int32_t *obs_frame = (int32_t *)std::malloc(sizeof(int32_t) * std::max<int64_t>(1, 2 * M));
    int32_t *obs_kp = (int32_t *)std::malloc(sizeof(int32_t) * std::max<int64_t>(1, 4 * M));
    double *obs_xy = with_xy ? (double *)std::malloc(sizeof(double) * std::max<int64_t>(1, 8 * M)) : nullptr;
    int64_t *frame_ptr = (int64_t *)std::malloc(sizeof(int64_t) * (n_pairs + 2));
    int64_t *frame_tracks = (int64_t *)std::malloc(sizeof(int64_t) * std::max<int64_t>(1, 2 * M));
    int64_t n_tracks = -1, n_obs = -1;
    const int rc = slam_chain_tracks(kp.data(), ptr.data(), n_pairs, n_kp, with_xy ? xy.data() : nullptr, n_max, track_ptr,
                                     obs_frame, obs_kp, obs_xy, frame_ptr, frame_tracks, &n_tracks, &n_obs);
    EXPECT(rc == SLAM_OK);
    if (rc == SLAM_OK) {
        EXPECT(n_tracks == (int64_t)want.size() && n_obs == want_obs);
        for (int64_t t = 0; t < n_tracks && bad == 0; ++t) {
            EXPECT(track_ptr[t + 1] - track_ptr[t] == (int64_t)want[t].size());
            for (size_t o = 0; o < want[t].size(); ++o) {
                const int64_t row = track_ptr[t] + (int64_t)o;
                EXPECT(obs_frame[row] == want[t][o][0] && obs_kp[row * 2] == want[t][o][1] && obs_kp[row * 2 + 1] == want[t][o][2]);
                if (with_xy) {
                    const size_t fs = (size_t)2 * n_max * 2;
                    const float *l = xy.data() + want[t][o][0] * fs + (size_t)want[t][o][1] * 2;
                    const float *r = xy.data() + want[t][o][0] * fs + (size_t)n_max * 2 + (size_t)want[t][o][2] * 2;
                    const double *g = obs_xy + row * 4;
                    EXPECT(g[0] == l[0] && g[1] == r[0] && g[2] == l[1] && g[3] == r[1]);
                }
            }
        }
        for (int f = 0; f <= n_pairs && bad == 0; ++f) {
            EXPECT(frame_ptr[f + 1] - frame_ptr[f] == (int64_t)want_frame[f].size());
            for (size_t k = 0; k < want_frame[f].size(); ++k) EXPECT(frame_tracks[frame_ptr[f] + (int64_t)k] == want_frame[f][k]);
        }
    }
    std::free(track_ptr); std::free(obs_frame); std::free(obs_kp); std::free(obs_xy); std::free(frame_ptr); std::free(frame_tracks);
}

static void chain_rejections() {
    int64_t ptr[2] = {0, 2};
    int64_t tp[3], fp[3], ft[4], nt, no;
    int32_t of[4], ok[8];
    int32_t dup0[8] = {3, 0, 1, 0, 3, 0, 2, 0}, dup2[8] = {3, 0, 1, 0, 4, 0, 1, 0}, oob[8] = {3, 0, 1, 0, 9, 0, 2, 0};
    EXPECT(slam_chain_tracks(dup0, ptr, 1, 8, nullptr, 0, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_DUPLICATE);
    EXPECT(slam_chain_tracks(dup2, ptr, 1, 8, nullptr, 0, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_DUPLICATE);
    EXPECT(slam_chain_tracks(oob, ptr, 1, 8, nullptr, 0, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_INVALID);
    int64_t desc[2] = {0, -1};
    EXPECT(slam_chain_tracks(dup0, desc, 1, 8, nullptr, 0, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_INVALID);
    EXPECT(slam_chain_tracks(nullptr, ptr, 1, 8, nullptr, 0, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_INVALID);
    float xy1[1] = {0};
    EXPECT(slam_chain_tracks(oob, ptr, 1, 8, xy1, 4, tp, of, ok, nullptr, fp, ft, &nt, &no) == SLAM_E_INVALID);   // no h_obs_xy, n_max < n_kp
}

// ---- the random stream: exact-size outputs, chunk independence, the rejection count ----
static void stream_cases(std::mt19937 &rng) {
    for (int it = 0; it < 200; ++it) {
        const uint32_t seed = rng(), bound = 1 + rng() % (it % 3 == 0 ? 5u : 70000u);
        const long n = (long)(rng() % 700);
        int32_t *out = (int32_t *)std::malloc(sizeof(int32_t) * std::max<long>(1, n));
        const long used = slam_randint_host(seed, bound, out, n);
        EXPECT(used >= (bound == 1 ? 0 : n));
        for (long i = 0; i < n; ++i) EXPECT(out[i] >= 0 && (uint32_t)out[i] < bound);
        if (bound > 1 && used > 0) {
            // the same numbers from the raw words by the masked rejection rule
            uint32_t *w = (uint32_t *)std::malloc(sizeof(uint32_t) * used);
            EXPECT(slam_mt19937_words(seed, w, (int)used) == SLAM_OK);
            uint32_t mask = bound - 1;
            for (int s = 1; s < 32; s <<= 1) mask |= mask >> s;
            long k = 0;
            for (long i = 0; i < used && k < n; ++i)
                if ((w[i] & mask) <= bound - 1) EXPECT(out[k++] == (int32_t)(w[i] & mask));
            EXPECT(k == n);
            std::free(w);
        }
        std::free(out);
    }
    EXPECT(slam_randint_host(1, 0, nullptr, 0) == SLAM_E_EMPTY);
    EXPECT(slam_randint_host(1, 5, nullptr, 3) == SLAM_E_INVALID);
    EXPECT(slam_mt19937_words(1, nullptr, 0) == SLAM_OK && slam_mt19937_words(1, nullptr, 2) == SLAM_E_INVALID);
}

// ---- the adaptive loop: one consume() over all draws == the same draws handed over in ragged rounds ----
static void replay_cases(std::mt19937 &rng) {
    for (int it = 0; it < 300; ++it) {
        RansacParams p;
        p.max_iterations = it % 5 == 0 ? 40 : 1000;
        p.fixed_iterations = it % 7 == 0 ? 33 : 0;
        p.max_draws = it % 11 == 0 ? 90 : 64000;
        const int n_points = 5 + (int)(rng() % 900);
        const int n_draws = 1 + (int)(rng() % 3000);
        const double inlier_rate = (rng() % 1000) / 1000.0;
        uint8_t *status = (uint8_t *)std::malloc(n_draws);
        int32_t *count = (int32_t *)std::malloc(sizeof(int32_t) * n_draws);
        for (int d = 0; d < n_draws; ++d) {
            status[d] = rng() % 10 != 0;
            count[d] = (int32_t)(n_points * inlier_rate * ((rng() % 1000) / 1000.0));
        }
        RansacReplay whole;
        whole.consume(p, n_points, status, count, 0, n_draws);
        RansacReplay parts;
        int begin = 0;
        while (!parts.finished && begin < n_draws) {
            const int end = std::min(n_draws, begin + 1 + (int)(rng() % 200));
            // a round's arrays start at its first draw: hand over a copy of exactly that slice
            uint8_t *st = (uint8_t *)std::malloc(end - begin);
            int32_t *ct = (int32_t *)std::malloc(sizeof(int32_t) * (end - begin));
            std::memcpy(st, status + begin, end - begin);
            std::memcpy(ct, count + begin, sizeof(int32_t) * (end - begin));
            parts.consume(p, n_points, st, ct, begin, end);
            std::free(st); std::free(ct);
            EXPECT(parts.draws <= end);
            if (!parts.finished) { EXPECT(parts.draws == end); EXPECT(parts.draws_hint(p) >= 1); }
            begin = end;
        }
        EXPECT(parts.finished == whole.finished && parts.it == whole.it && parts.draws == whole.draws &&
               parts.best_draw == whole.best_draw && parts.max_supp == whole.max_supp && parts.hit_cap == whole.hit_cap);
        EXPECT(whole.wants_more(p) == !whole.finished || whole.hit_cap);
        std::free(status); std::free(count);
    }
}

static void compose_cases(std::mt19937 &rng) {
    for (int n : {0, 1, 2, 57}) {
        double *rel = (double *)std::malloc(sizeof(double) * std::max(1, 12 * n));
        double *glob = (double *)std::malloc(sizeof(double) * std::max(1, 12 * n));
        for (int i = 0; i < 12 * n; ++i) rel[i] = (double)(rng() % 2000) / 1000.0 - 1.0;
        if (n) { const double eye[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; std::memcpy(rel, eye, sizeof eye); }
        EXPECT(slam_compose_global(rel, n, glob) == SLAM_OK);
        for (int i = 1; i < n; ++i)                            // T_i @ hom(G_{i-1}), row 0 column 3 by hand
            EXPECT(std::fabs(glob[12 * i + 3] - (rel[12 * i] * glob[12 * (i - 1) + 3] + rel[12 * i + 1] * glob[12 * (i - 1) + 7] +
                                                 rel[12 * i + 2] * glob[12 * (i - 1) + 11] + rel[12 * i + 3])) < 1e-9 * (1 + std::fabs(glob[12 * i + 3])));
        std::free(rel); std::free(glob);
    }
    EXPECT(slam_compose_global(nullptr, 3, nullptr) == SLAM_E_INVALID);
}

int main() {
    std::mt19937 rng(20240607);
    chain_case(rng, 0, 8, 8, false, 0);
    chain_case(rng, 1, 1, 64, true, 0);
    for (int it = 0; it < 60; ++it)
        chain_case(rng, 1 + (int)(rng() % 40), 1 + (int)(rng() % 90), 128, it % 2 == 0, it % 3 == 0 ? 4 : 0);
    chain_case(rng, 600, 900, 960, true, 7);                 // > 2^16 tracks: the pool splits the placement by track range
    chain_rejections();
    stream_cases(rng);
    replay_cases(rng);
    compose_cases(rng);
    std::printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
