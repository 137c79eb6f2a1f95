// Micro-benchmark: sustained VALU issue rate of the instructions the matcher is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/valu_peak ; run on the GPU box.
// Prints T lane-ops/s per instruction kind with every SIMD holding WAVES_PER_SIMD waves.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 4096;
int g_reps = 1;   // launches per timed region (sustained-clock runs use many)
constexpr int UNROLL = 16;   // independent chains per lane

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    unsigned v[UNROLL];
    float f[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        v[i] = seed * (threadIdx.x + 1) + i;
        f[i] = (float)v[i];
    }
    const unsigned b = seed ^ 0x9e3779b9u, c = seed * 31u;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (KIND == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[i]) : "v"(b));
            if (KIND == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(b));
            if (KIND == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xde" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(b));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(1.0001f), "v"(0.5f));
            if (KIND == 5) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 6) asm volatile("v_lshl_or_b32 %0, %0, 20, %1" : "+v"(v[i]) : "v"(b));
            if (KIND == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 12" : "+v"(v[i]) : "v"(b));
            if (KIND == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 9) asm volatile("v_min_u32 %0, %1, %0" : "+v"(v[i]) : "v"(b));
            if (KIND == 10) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 12) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 13) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(b));
            if (KIND == 14) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 15) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 16) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(c));
            if (KIND == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
        }
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= v[i] ^ (unsigned)f[i];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int KIND>
double run(const char *name, int waves_per_simd, unsigned *d) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        for (int g = 0; g < g_reps; ++g) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double ops = (double)blocks * 256 * ITER * UNROLL * g_reps;
    const double rate = ops / (best * 1e-3) / 1e12;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f T lane-ops/s  (%.2f lanes/clk/SIMD at 2.4 GHz)\n", name, waves_per_simd, best,
           rate, rate * 1e12 / (1024 * 2.4e9));
    return rate;
}

int main() {
    unsigned *d;
    CHECK(hipMalloc(&d, 256));
    printf("== sustained (40 back-to-back launches, 4 waves/SIMD) ==\n");
    g_reps = 40;
    run<0>("v_xor_b32", 4, d);
    run<2>("v_bitop3_b32", 4, d);
    run<1>("v_bcnt_u32_b32", 4, d);
    run<5>("v_med3_u32", 4, d);
    run<6>("v_lshl_or_b32", 4, d);
    run<7>("v_alignbit_b32", 4, d);
    run<8>("v_mad_u32_u24", 4, d);
    run<9>("v_min_u32", 4, d);
    run<10>("v_bfi_b32", 4, d);
    run<11>("v_and_or_b32", 4, d);
    run<12>("v_perm_b32", 4, d);
    run<13>("v_lshl_add_u32", 4, d);
    run<14>("v_add3_u32", 4, d);
    run<15>("v_or3_b32", 4, d);
    run<16>("v_sad_u8", 4, d);
    run<17>("v_cndmask_b32", 4, d);
    g_reps = 1;
    printf("== single launches ==\n");
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32", w, d);
        run<1>("v_bcnt_u32_b32", w, d);
        run<2>("v_bitop3_b32", w, d);
        run<3>("v_add_u32", w, d);
        run<4>("v_fma_f32", w, d);
        run<5>("v_med3_u32", w, d);
        run<6>("v_lshl_or_b32", w, d);
    }
    return 0;
}
