// Ablation of the matcher's inner loop (gfx950): same instruction mix as allpairs_top2_kernel, with the B rows
//   MODE 0: held in registers (no memory traffic at all)  -> pure VALU ceiling of this mix
//   MODE 1: read through LDS broadcasts (as the product kernel)
//   MODE 2: read through scalar loads (uniform address -> s_load_dwordx16, no LDS, no VGPRs for B)
// Reports descriptor pairs/s and issue slots/s; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROWS = 4, NB = 2048;

__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ unsigned dq(const uint4 &a, const uint4 &b, unsigned acc) {
    acc = bcnt_acc(__builtin_amdgcn_bitop3_b32(a.x, a.z ^ b.z, b.x, 0xde), acc);
    acc = bcnt_acc(__builtin_amdgcn_bitop3_b32(a.y, a.w ^ b.w, b.y, 0xde), acc);
    return acc;
}
__device__ __forceinline__ void ins(unsigned &m1, unsigned &m2, unsigned key, bool top2) {
    if (top2) {
        unsigned med;
        asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));
        m2 = med;
    }
    m1 = min(m1, key);
}

template <int MODE, bool TOP2>
__global__ __launch_bounds__(128) void k(const uint4 *__restrict__ A, const uint4 *__restrict__ B, unsigned *out, int reps) {
    __shared__ uint4 tile[NB * 4 / 8];   // 256 rows
    uint4 a[ROWS][4];
    for (int r = 0; r < ROWS; ++r)
        for (int c = 0; c < 4; ++c) a[r][c] = A[((blockIdx.x * 128 + threadIdx.x) * ROWS + r) % 4096 * 4 + c];
    unsigned m1[ROWS], m2[ROWS];
    for (int r = 0; r < ROWS; ++r) m1[r] = m2[r] = 0xFFFFFFFFu;
    if (MODE == 1) {
        for (int i = threadIdx.x; i < 256 * 4; i += 128) tile[i] = B[i];
        __syncthreads();
    }
    uint4 rb0 = B[0], rb1 = B[1], rb2 = B[2], rb3 = B[3];
    for (int rep = 0; rep < reps; ++rep) {
        for (int j = 0; j < NB; ++j) {
            uint4 b0, b1, b2, b3;
            if (MODE == 0) {
                b0 = rb0; b1 = rb1; b2 = rb2; b3 = rb3;
                // make every dword opaque per iteration so nothing is hoisted out of the loop
                asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w));
                asm volatile("" : "+v"(b2.x), "+v"(b2.y), "+v"(b2.z), "+v"(b2.w), "+v"(b3.x), "+v"(b3.y), "+v"(b3.z), "+v"(b3.w));
            } else if (MODE == 1) {
                const uint4 *t = tile + (j & 255) * 4;
                b0 = t[0]; b1 = t[1]; b2 = t[2]; b3 = t[3];
            } else {
                const uint4 *t = B + (size_t)j * 4;     // uniform address: scalar loads
                b0 = t[0]; b1 = t[1]; b2 = t[2]; b3 = t[3];
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                unsigned d = 0;
                d = dq(a[r][0], b0, d);
                d = dq(a[r][1], b1, d);
                d = dq(a[r][2], b2, d);
                d = dq(a[r][3], b3, d);
                ins(m1[r], m2[r], (d << 20) | (unsigned)j, TOP2);
            }
        }
    }
    unsigned acc = 0;
    for (int r = 0; r < ROWS; ++r) acc ^= m1[r] ^ m2[r];
    out[blockIdx.x * 128 + threadIdx.x] = acc;
}

template <int MODE, bool TOP2>
void run(const char *name, const uint4 *A, const uint4 *B, unsigned *out, int waves_per_simd) {
    const int blocks = 256 * 2 * waves_per_simd;   // 128-thread blocks: 2 per CU per wave/SIMD
    const int reps = 4;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, TOP2>), dim3(blocks), dim3(128), 0, 0, A, B, out, reps);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int g = 0; g < 3; ++g) hipLaunchKernelGGL((k<MODE, TOP2>), dim3(blocks), dim3(128), 0, 0, A, B, out, reps);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double pairs = 3.0 * blocks * 128 * ROWS * (double)NB * reps;
    printf("%-28s waves/SIMD=%d  %.2f ms  %.2f G descriptor pairs/s  (%.1f T instr-lanes/s at %d instr/pair)\n", name,
           waves_per_simd, ms, pairs / (ms * 1e-3) / 1e9, pairs * (TOP2 ? 27 : 26) / (ms * 1e-3) / 1e12, TOP2 ? 27 : 26);
}

int main() {
    uint4 *A, *B;
    unsigned *out;
    CHECK(hipMalloc(&A, 4096 * 64));
    CHECK(hipMalloc(&B, NB * 64));
    CHECK(hipMalloc(&out, 256 * 2 * 8 * 128 * 4));
    unsigned *h = (unsigned *)malloc(4096 * 64);
    for (int i = 0; i < 4096 * 16; ++i) h[i] = (unsigned)rand() * 2654435761u;
    CHECK(hipMemcpy(A, h, 4096 * 64, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, h + 1000, NB * 64, hipMemcpyHostToDevice));
    for (int w : {2, 4, 5}) {
        run<0, true>("registers (VALU ceiling)", A, B, out, w);
        run<1, true>("LDS broadcast", A, B, out, w);
        run<2, true>("scalar loads", A, B, out, w);
        run<0, false>("registers, top-1 only", A, B, out, w);
        run<1, false>("LDS broadcast, top-1 only", A, B, out, w);
        run<2, false>("scalar loads, top-1 only", A, B, out, w);
    }
    return 0;
}
