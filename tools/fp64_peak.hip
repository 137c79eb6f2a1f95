// Micro-benchmark: sustained issue rate of the FP64 instructions the triangulation / AP3P / scoring / refit kernels are
// made of (gfx950).  Same harness as tools/valu_peak.hip: every SIMD holds WAVES waves of 16 independent chains.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2048;
constexpr int UNROLL = 16;
int g_reps = 1;

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    double v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = seed * (threadIdx.x + 1) + i;
    const double b = 1.0000001, c = 0.25;
    unsigned long long vc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (KIND == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            if (KIND == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i]));
            if (KIND == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[i]));
            if (KIND == 6) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(vc) : "v"(v[i]), "v"(b));
            if (KIND == 7) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
        }
    }
    double acc = (double)vc;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += v[i];
    if (acc == 0.123) out[0] = acc;
}

template <int KIND>
void run(const char *name, int waves_per_simd, double *d) {
    const int blocks = 256 * waves_per_simd;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        for (int g = 0; g < g_reps; ++g) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.5);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double ops = (double)blocks * 256 * ITER * UNROLL * g_reps;
    const double rate = ops / (best * 1e-3) / 1e12;
    printf("%-14s waves/SIMD=%d  %8.3f ms  %6.2f T lane-ops/s  (%.2f lanes/clk/SIMD at 2.4 GHz)\n", name, waves_per_simd, best, rate,
           rate * 1e12 / (1024 * 2.4e9));
}

int main() {
    double *d;
    CHECK(hipMalloc(&d, 256));
    for (int reps : {20, 1}) {
        g_reps = reps;
        printf("== %d back-to-back launches ==\n", reps);
        for (int w : {2, 4}) {
            run<0>("v_fma_f64", w, d);
            run<7>("v_fmac_f64", w, d);
            run<1>("v_mul_f64", w, d);
            run<2>("v_add_f64", w, d);
            run<3>("v_max_f64", w, d);
            run<6>("v_cmp_lt_f64", w, d);
            run<4>("v_rcp_f64", w, d);
            run<5>("v_rsq_f64", w, d);
        }
    }
    return 0;
}
