// slam_track_pairs under HOST AddressSanitizer + UBSan (device code is not instrumented: GPU sanitizers are not
// available on this pool).  Built by tools/asan_track.sh from all library sources; reads the case files of
// tools/asan_track_case.py.  The host outputs are pageable heap blocks of exactly the documented sizes, the host
// workspace is pinned as the header asks, and every output is compared bit for bit with the ordinary library's.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/slam_amd.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

template <typename T>
static bool read_vec(std::FILE *f, std::vector<T> &v, size_t n) {
    v.resize(n);
    return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

static int run_case(const char *path) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); return 2; }
    int32_t head[4];
    if (std::fread(head, sizeof(int32_t), 4, f) != 4 || head[3] != (int32_t)sizeof(SlamTrackConfig)) {
        std::fprintf(stderr, "%s: bad header (config struct %d bytes, library %zu)\n", path, head[3], sizeof(SlamTrackConfig));
        return 2;
    }
    const int F = head[0], P = head[1], N = head[2];
    SlamTrackConfig cfg;
    if (std::fread(&cfg, sizeof cfg, 1, f) != 1) return 2;
    std::vector<uint8_t> desc;
    std::vector<float> xy;
    std::vector<int32_t> count, f0, f1, want_stats, want_kp;
    std::vector<uint32_t> seeds;
    std::vector<double> want_pose;
    std::vector<uint64_t> want_mask;
    bool ok = read_vec(f, desc, (size_t)2 * F * N * 64) && read_vec(f, xy, (size_t)2 * F * N * 2) && read_vec(f, count, 2 * F) &&
              read_vec(f, f0, P) && read_vec(f, f1, P) && read_vec(f, seeds, P) && read_vec(f, want_pose, (size_t)P * 12) &&
              read_vec(f, want_stats, (size_t)P * 8) && read_vec(f, want_mask, (size_t)P * (N / 64)) &&
              read_vec(f, want_kp, (size_t)P * N * 4);
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "%s: truncated\n", path); return 2; }

    uint8_t *d_desc; float *d_xy; int32_t *d_count, *d_kp; void *d_ws, *h_ws;
    HIP_OK(hipMalloc(&d_desc, desc.size()));
    HIP_OK(hipMalloc(&d_xy, xy.size() * sizeof(float)));
    HIP_OK(hipMalloc(&d_count, count.size() * sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_kp, (size_t)P * N * 4 * sizeof(int32_t)));
    HIP_OK(hipMemcpy(d_desc, desc.data(), desc.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_xy, xy.data(), xy.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_count, count.data(), count.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_kp, 0, (size_t)P * N * 4 * sizeof(int32_t)));
    const size_t dwb = slam_track_device_workspace_bytes(F, P, N, &cfg), hwb = slam_track_host_workspace_bytes(F, P, N, &cfg);
    HIP_OK(hipMalloc(&d_ws, dwb));
    HIP_OK(hipHostMalloc(&h_ws, hwb, hipHostMallocDefault));
    // host inputs and outputs as exact heap blocks (copies, so that a read past the inputs is seen too)
    int32_t *h_f0 = (int32_t *)std::malloc(sizeof(int32_t) * P), *h_f1 = (int32_t *)std::malloc(sizeof(int32_t) * P);
    uint32_t *h_seed = (uint32_t *)std::malloc(sizeof(uint32_t) * P);
    std::memcpy(h_f0, f0.data(), sizeof(int32_t) * P);
    std::memcpy(h_f1, f1.data(), sizeof(int32_t) * P);
    std::memcpy(h_seed, seeds.data(), sizeof(uint32_t) * P);
    double *h_pose = (double *)std::malloc(sizeof(double) * 12 * P);
    int32_t *h_stats = (int32_t *)std::malloc(sizeof(int32_t) * 8 * P);
    uint64_t *h_mask = (uint64_t *)std::malloc(sizeof(uint64_t) * (N / 64) * P);

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    SlamTrackIO io;
    std::memset(&io, 0, sizeof io);
    io.d_desc = d_desc; io.d_xy = d_xy; io.d_count = d_count;
    io.n_frames = F; io.n_max = N;
    io.h_pair_f0 = h_f0; io.h_pair_f1 = h_f1; io.h_pair_seed = h_seed; io.n_pairs = P;
    io.d_tracked_kp = d_kp;
    io.h_pose = h_pose; io.h_stats = h_stats; io.h_mask = h_mask;
    io.d_workspace = d_ws; io.d_workspace_bytes = dwb;
    io.h_workspace = h_ws; io.h_workspace_bytes = hwb;
    long bad = 0;
    for (int rep = 0; rep < 2; ++rep) {          // twice: the second call reuses the thread's streams / events / pool
        std::memset(h_pose, 0xAB, sizeof(double) * 12 * P);
        std::memset(h_stats, 0xAB, sizeof(int32_t) * 8 * P);
        std::memset(h_mask, 0xAB, sizeof(uint64_t) * (N / 64) * P);
        const int rc = slam_track_pairs(&cfg, &io, st);
        if (rc != SLAM_OK) { std::fprintf(stderr, "%s: rc %d: %s\n", path, rc, slam_last_error_string()); return 1; }
        HIP_OK(hipStreamSynchronize(st));
        bad += std::memcmp(h_pose, want_pose.data(), sizeof(double) * 12 * P) != 0;
        bad += std::memcmp(h_stats, want_stats.data(), sizeof(int32_t) * 8 * P) != 0;
        bad += std::memcmp(h_mask, want_mask.data(), sizeof(uint64_t) * (N / 64) * P) != 0;
        std::vector<int32_t> kp((size_t)P * N * 4);
        HIP_OK(hipMemcpy(kp.data(), d_kp, kp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int p = 0; p < P; ++p)              // rows past a pair's tracked count are not part of the result
            bad += std::memcmp(kp.data() + (size_t)p * N * 4, want_kp.data() + (size_t)p * N * 4, sizeof(int32_t) * 4 * want_stats[p * 8]) != 0;
    }
    // the error paths leave through the same frames: a workspace one byte short, a frame index out of range
    io.d_workspace_bytes = dwb / 2;
    bad += slam_track_pairs(&cfg, &io, st) != SLAM_E_WORKSPACE;
    io.d_workspace_bytes = dwb;
    h_f1[P - 1] = F;
    bad += slam_track_pairs(&cfg, &io, st) != SLAM_E_INVALID;
    std::printf("%s: F %d P %d N %d chunks %d mode %d: %s\n", path, F, P, N, cfg.pipeline_chunks, cfg.match_mode, bad ? "MISMATCH" : "equal");
    HIP_OK(hipStreamDestroy(st));
    std::free(h_f0); std::free(h_f1); std::free(h_seed); std::free(h_pose); std::free(h_stats); std::free(h_mask);
    HIP_OK(hipHostFree(h_ws));
    HIP_OK(hipFree(d_ws)); HIP_OK(hipFree(d_kp)); HIP_OK(hipFree(d_count)); HIP_OK(hipFree(d_xy)); HIP_OK(hipFree(d_desc));
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    int rc = 0;
    for (int i = 1; i < argc && rc == 0; ++i) rc = run_case(argv[i]);
    std::printf("asan_track_driver: %s\n", rc ? "FAILED" : "all cases equal, no sanitizer report");
    return rc;
}
