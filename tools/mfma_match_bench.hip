// Ceiling measurement: NORM_HAMMING2 all-pairs distances on the matrix cores instead of the VALU.
//
// Exact +-1 encoding: a 2-bit cell (b1, b0) becomes three values x = 1 - 2 b0, y = 1 - 2 b1, z = x y in {+1, -1}.  For two
// cells the sum x x' + y y' + z z' is 3 when the cells are equal and -1 otherwise, so over the 244 cells of a 61-byte
// descriptor S = sum of products = 4 (#equal cells) - 244 and the distance is d = (732 - S) / 4, exactly, in an i32 or
// f32 accumulator (|S| <= 732).  K = 732 values per row, padded with zeros to 768.
//   FMT 0: int8 operands,  v_mfma_i32_32x32x32_i8              -> 24 MFMAs per 32 x 32 tile of descriptor pairs
//   FMT 1: fp4 (E2M1: +1 = 0x2, -1 = 0xA) operands, v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales -> 12 MFMAs per tile
// A wave owns 32 "A" rows whose fragments stay in registers; "B" rows pass through LDS in 32-row tiles stored
// fragment-major (one conflict-free ds_read_b128 per MFMA).  Epilogue per tile: every lane holds, for B row
// (tile, lane & 31), the sums against 16 of the A rows; it folds them into 16 running maxima of the key
// (S << 12 | 4095 - tile) -- largest S = nearest row, earliest tile on ties -- and the 32 lanes of an A row are
// reduced once at the end (lowest B row on ties: the matcher's first-minimum rule).
// The product keeps the VALU matcher (north_star: XOR / popcount for binary descriptors); this file only measures what
// the other formulation could reach.  Prints descriptor pairs/s; checks the result of a small case against the
// bit-by-bit definition on the host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int KPAD = 768;                      // values per encoded row
template <int FMT> struct Fmt;
template <> struct Fmt<0> { static constexpr int NCH = KPAD / 32; static constexpr int ROWB = KPAD; };        // int8
template <> struct Fmt<1> { static constexpr int NCH = KPAD / 64; static constexpr int ROWB = KPAD / 2; };    // fp4
constexpr int TILE = 32;                       // rows per MFMA tile

// encoded image layout (A and B alike): [tile][chunk][lane][16 bytes], lane = (row & 31) + 32 * half; lane (r, h) of
// chunk c holds the h-th 16 bytes of the chunk's bytes of row r.  A and B use the same k order, so any k order inside
// the hardware's fragment gives the same dot products.
template <int FMT>
__host__ __device__ inline size_t frag_offset(int row, int byte_in_row) {
    constexpr int CB = FMT == 0 ? 32 : 32;     // bytes of one row per chunk (32 int8, or 64 fp4 = 32 bytes)
    const int tile = row / TILE, r = row % TILE, c = byte_in_row / CB, h = (byte_in_row % CB) / 16, b = byte_in_row % 16;
    return (((size_t)tile * Fmt<FMT>::NCH + c) * 64 + (r + 32 * h)) * 16 + b;
}

template <int FMT, int BT, bool EPI>
__global__ __launch_bounds__(256) void mfma_match_kernel(const uint4 *__restrict__ A, const uint4 *__restrict__ B, int n_btiles,
                                                         int reps, unsigned *__restrict__ out_key, int *__restrict__ sink) {
    constexpr int NCH = Fmt<FMT>::NCH;
    extern __shared__ uint4 lds[];             // BT tiles of B, fragment-major: BT * NCH * 64 uint4
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a_tile = blockIdx.x * 4 + wave;
    uint4 a[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) a[c] = A[((size_t)a_tile * NCH + c) * 64 + lane];
    int best[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = INT32_MIN;
    int chk = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int t0 = 0; t0 < n_btiles; t0 += BT) {
            __syncthreads();
            for (int i = threadIdx.x; i < BT * NCH * 64; i += 256) lds[i] = B[(size_t)t0 * NCH * 64 + i];
            __syncthreads();
#pragma unroll 1
            for (int t = 0; t < BT; ++t) {
                const uint4 *bt = lds + (size_t)t * NCH * 64 + lane;
                int s[16];
                if (FMT == 0) {
                    v16i acc = {0};
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const uint4 b = bt[c * 64];
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const v4i *)&a[c], *(const v4i *)&b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = acc[r];
                } else {
                    v16f acc = {0};
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const uint4 b = bt[c * 64];
                        v8i av = {(int)a[c].x, (int)a[c].y, (int)a[c].z, (int)a[c].w, 0, 0, 0, 0};
                        v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
                        // cbsz = blgp = 4: fp4 operands; scales 127 = 2^0 (E8M0)
                        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 127, 0, 127);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (int)acc[r];
                }
                if (EPI) {
                    const int tag = 4095 - (t0 + t);
#pragma unroll
                    for (int r = 0; r < 16; ++r) best[r] = max(best[r], (s[r] << 12) | tag);     // s >= -732: arithmetic shift keeps order
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) chk ^= s[r];
                }
            }
        }
    }
    if (EPI) {
        // reduce the 32 lanes that share an A row (same half): max S, then the lowest B row = (tile, lane & 31)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // full key: S (signed) | tile tag | 31 - (lane & 31)  -> one max over the lanes of a half
            long long k = ((long long)best[r] << 5) | (31 - (lane & 31));
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
                const long long o = __shfl_xor(k, m);
                k = o > k ? o : k;
            }
            if ((lane & 31) == 0) {
                const int a_row = a_tile * TILE + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int S = (int)(k >> 17), tile = 4095 - (int)((k >> 5) & 4095), col = 31 - (int)(k & 31);
                out_key[a_row] = ((unsigned)((732 - S) / 4) << 20) | (unsigned)(tile * TILE + col);
            }
        }
    } else if (chk == 0x7fffffff) {
        sink[0] = chk;
    }
}

static int hamming2(const uint8_t *a, const uint8_t *b) {
    int d = 0;
    for (int i = 0; i < 61; ++i) {
        const unsigned x = a[i] ^ b[i];
        d += __builtin_popcount((x | (x >> 1)) & 0x55);
    }
    return d;
}

template <int FMT>
static void encode(const std::vector<uint8_t> &desc, int n_rows, std::vector<uint8_t> &enc) {
    const int n_tiles = (n_rows + TILE - 1) / TILE;
    enc.assign((size_t)n_tiles * Fmt<FMT>::NCH * 64 * 16, 0);
    for (int row = 0; row < n_rows; ++row)
        for (int cell = 0; cell < 244; ++cell) {
            const int b0 = (desc[(size_t)row * 61 + cell / 4] >> (2 * (cell % 4))) & 1, b1 = (desc[(size_t)row * 61 + cell / 4] >> (2 * (cell % 4) + 1)) & 1;
            const int v[3] = {1 - 2 * b0, 1 - 2 * b1, (1 - 2 * b0) * (1 - 2 * b1)};
            for (int j = 0; j < 3; ++j) {
                const int k = cell * 3 + j;
                if (FMT == 0) {
                    enc[frag_offset<0>(row, k)] = (uint8_t)(int8_t)v[j];
                } else {
                    const uint8_t nib = v[j] > 0 ? 0x2 : 0xA;                 // E2M1: +1.0 / -1.0
                    uint8_t &byte = enc[frag_offset<1>(row, k / 2)];
                    byte |= (k & 1) ? (uint8_t)(nib << 4) : nib;
                }
            }
        }
}

template <int FMT, int BT>
static void run(const char *label, const std::vector<uint8_t> &dA, const std::vector<uint8_t> &dB, int na, int nb) {
    std::vector<uint8_t> eA, eB;
    encode<FMT>(dA, na, eA);
    encode<FMT>(dB, nb, eB);
    uint4 *A, *B;
    unsigned *out;
    int *sink;
    CHECK(hipMalloc(&A, eA.size()));
    CHECK(hipMalloc(&B, eB.size()));
    CHECK(hipMalloc(&out, sizeof(unsigned) * na));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(A, eA.data(), eA.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, eB.data(), eB.size(), hipMemcpyHostToDevice));
    const int n_btiles = nb / TILE, grid = na / (4 * TILE);
    const size_t lds = (size_t)BT * Fmt<FMT>::NCH * 64 * 16;
    CHECK(hipFuncSetAttribute((const void *)mfma_match_kernel<FMT, BT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute((const void *)mfma_match_kernel<FMT, BT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // ---- correctness of the formulation and of the epilogue: nearest B row of every A row, first minimum ----
    hipLaunchKernelGGL((mfma_match_kernel<FMT, BT, true>), dim3(grid), dim3(256), lds, 0, A, B, n_btiles, 1, out, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> got(na);
    CHECK(hipMemcpy(got.data(), out, sizeof(unsigned) * na, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < na && i < 512; ++i) {
        int bd = 1 << 30, bj = -1;
        for (int j = 0; j < nb; ++j) {
            const int d = hamming2(&dA[(size_t)i * 61], &dB[(size_t)j * 61]);
            if (d < bd) bd = d, bj = j;
        }
        if (got[i] != (((unsigned)bd << 20) | (unsigned)bj)) {
            if (bad++ < 4) printf("  row %d: got d=%u j=%u want d=%d j=%d\n", i, got[i] >> 20, got[i] & 0xFFFFF, bd, bj);
        }
    }
    printf("%-36s exact nearest row on the first 512 A rows: %s\n", label, bad ? "MISMATCH" : "ok");
    // ---- throughput: 3 warm launches, then 20 timed back to back ----
    for (int epi = 1; epi >= 0; --epi) {
        const int reps = 24;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        auto launch = [&] {
            if (epi) hipLaunchKernelGGL((mfma_match_kernel<FMT, BT, true>), dim3(grid), dim3(256), lds, 0, A, B, n_btiles, reps, out, sink);
            else hipLaunchKernelGGL((mfma_match_kernel<FMT, BT, false>), dim3(grid), dim3(256), lds, 0, A, B, n_btiles, reps, out, sink);
        };
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e0));
        const int L = 20;
        for (int i = 0; i < L; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double pairs = (double)na * nb * reps * L;
        printf("%-36s %-22s %8.3f ms/launch  %6.3f T descriptor pairs/s\n", label, epi ? "with min/argmin epilogue" : "MFMA + LDS only", ms / L,
               pairs / (ms * 1e-3) / 1e12);
    }
    CHECK(hipFree(A));
    CHECK(hipFree(B));
    CHECK(hipFree(out));
    CHECK(hipFree(sink));
}

int main() {
    const int na = 256 * 12 * 4 * TILE, nb = 2048;       // 12 workgroups of 4 waves per CU; one image of B rows
    std::vector<uint8_t> dA((size_t)na * 61), dB((size_t)nb * 61);
    srand(7);
    for (auto &v : dB) v = (uint8_t)(rand() >> 7);
    for (size_t i = 0; i < dB.size(); i += 61) dB[i + 60] &= 0x3F;
    for (int i = 0; i < na; ++i) {
        // two thirds of the A rows are noisy copies of a B row (a true match, with duplicates -> ties), the rest clutter
        if (i % 3 != 2) {
            memcpy(&dA[(size_t)i * 61], &dB[(size_t)((i * 7919) % nb) * 61], 61);
            for (int f = 0; f < 20; ++f) dA[(size_t)i * 61 + rand() % 61] ^= (uint8_t)(1 << (rand() % 8));
        } else {
            for (int b = 0; b < 61; ++b) dA[(size_t)i * 61 + b] = (uint8_t)(rand() >> 7);
        }
        dA[(size_t)i * 61 + 60] &= 0x3F;
    }
    printf("A rows %d (12 four-wave workgroups per CU), B rows %d, K = %d values per row\n", na, nb, KPAD);
    // B tiles per LDS stage set the workgroups per CU (LDS) next to the registers (int8: 160 VGPRs, fp4: 112)
    run<0, 1>("int8  i32_32x32x32, 1 tile/stage", dA, dB, na, nb);
    run<0, 2>("int8  i32_32x32x32, 2 tiles/stage", dA, dB, na, nb);
    run<1, 2>("fp4   f8f6f4_32x32x64, 2 tiles/stage", dA, dB, na, nb);
    run<1, 4>("fp4   f8f6f4_32x32x64, 4 tiles/stage", dA, dB, na, nb);
    run<1, 8>("fp4   f8f6f4_32x32x64, 8 tiles/stage", dA, dB, na, nb);
    return 0;
}
