// Dense wire format -> the padded device layout the path works on (the step in front of it).
//
// The reference hands every image's detections over as Python objects (utils/utills.py:46-54: key-point list +
// descriptor ndarray of cv2's detectAndCompute).  A shard of them crosses PCIe as three dense arrays -- descriptor rows
// at their own width (61 bytes for AKAZE MLDB), key-point coordinates, and a row pointer per image -- which is 8 % less
// than the padded layout (64-byte rows, n_max rows per image), and PCIe is what bounds a stream of shards once the
// kernels are where they are (17.8 ms of upload against a 15.5 ms step).  These two kernels rebuild the padded layout
// in HBM, writing every byte of it (rows past an image's count and bytes past the descriptor width are zero), so the
// result equals a padded upload bit for bit.  HBM-bound: 0.94 GB read + 1.02 GB written per 3450-frame shard.
#include "common.h"
#include "kernels.h"

namespace {

// rows of image `img` that may be unpacked: the count, unless the image's row range does not lie inside the dense arrays
__device__ __forceinline__ int usable_rows(const int64_t *row_ptr, const int *count, int img, int n_max, long n_rows_total) {
    const long a = row_ptr[img];
    int n = count[img];
    n = n < 0 ? 0 : (n > n_max ? n_max : n);
    if (a < 0 || a > n_rows_total - n) n = 0;      // not `a + n > total`: a hostile row pointer near INT64_MAX would wrap
    return n;
}

// One workgroup per 64 rows of an image's slot.  The rows' bytes are one contiguous run of the dense array (64 x width
// bytes, starting anywhere): it is brought into LDS with aligned, coalesced dword loads and the padded rows are put
// together from LDS bytes -- a first version that fetched every output dword with four byte loads from global memory
// took 3.8 ms per 3450-frame shard instead of < 1.
constexpr int UR = 64;    // rows per workgroup

__global__ __launch_bounds__(256) void unpack_desc_kernel(const uint8_t *dense, const int64_t *row_ptr, const int *count,
                                                          int n_max, int width, long n_rows_total, uint32_t *out,
                                                          int *count_out) {
    __shared__ uint32_t buf[UR * 64 / 4 + 2];
    const int img = blockIdx.y, tid = threadIdx.x;
    const int n = usable_rows(row_ptr, count, img, n_max, n_rows_total);
    const int r0 = blockIdx.x * UR;
    if (blockIdx.x == 0 && tid == 0) count_out[img] = n;
    const int rows = n - r0 < 0 ? 0 : (n - r0 > UR ? UR : n - r0);
    int off = 0;
    if (rows > 0) {
        const size_t total = (size_t)n_rows_total * width;
        const size_t begin = (size_t)(row_ptr[img] + r0) * width, end = begin + (size_t)rows * width;
        const size_t abegin = begin & ~(size_t)3;
        off = (int)(begin - abegin);
        const int ndw = (int)((end - abegin + 3) / 4);
        for (int i = tid; i < ndw; i += 256) {
            const size_t at = abegin + 4 * (size_t)i;
            uint32_t v = 0;
            if (at + 4 <= total) {
                v = *(const uint32_t *)(dense + at);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (at + b < total) v |= (uint32_t)dense[at + b] << (8 * b);      // the array's last, partial dword
            }
            buf[i] = v;
        }
    }
    __syncthreads();
    const uint8_t *bytes = (const uint8_t *)buf;
    uint32_t *dst = out + ((size_t)img * n_max + r0) * 16;
    for (int o = tid; o < UR * 16; o += 256) {
        const int r = o >> 4, w = o & 15;
        uint32_t v = 0;
        if (r < rows) {
            const int at = off + r * width + 4 * w;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (4 * w + b < width) v |= (uint32_t)bytes[at + b] << (8 * b);
        }
        dst[o] = v;
    }
}

__global__ __launch_bounds__(256) void unpack_xy_kernel(const float2 *dense, const int64_t *row_ptr, const int *count, int n_max,
                                                        long n_rows_total, float2 *out) {
    const int img = blockIdx.y;
    const int n = usable_rows(row_ptr, count, img, n_max, n_rows_total);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_max) return;
    out[(size_t)img * n_max + r] = r < n ? dense[row_ptr[img] + r] : make_float2(0.f, 0.f);
}

}  // namespace

extern "C" int slam_unpack_detections(const uint8_t *d_dense_desc, const float *d_dense_xy, const int64_t *d_row_ptr,
                                      const int32_t *d_count, int n_img, int n_max, int desc_width, int64_t n_rows_total,
                                      uint8_t *d_desc, float *d_xy, int32_t *d_count_out, void *stream) {
    if (n_img <= 0) return SLAM_OK;
    SLAM_REQUIRE(d_dense_desc && d_dense_xy && d_row_ptr && d_count && d_desc && d_xy && d_count_out,
                 "slam_unpack_detections: null pointer");
    SLAM_REQUIRE(n_max > 0 && n_max % 64 == 0 && n_max < SLAM_MAX_ROWS, "slam_unpack_detections: n_max must be a positive multiple of 64");
    SLAM_REQUIRE(desc_width >= 1 && desc_width <= 64, "slam_unpack_detections: descriptor width must be 1 .. 64 bytes");
    SLAM_REQUIRE(n_img <= 65535 && n_rows_total >= 0, "slam_unpack_detections: at most 65535 images per call");
    SLAM_REQUIRE(((uintptr_t)d_dense_desc & 3) == 0 && ((uintptr_t)d_dense_xy & 7) == 0 && ((uintptr_t)d_xy & 7) == 0 &&
                     ((uintptr_t)d_desc & 3) == 0 && ((uintptr_t)d_row_ptr & 7) == 0,
                 "slam_unpack_detections: d_dense_desc / d_desc must be 4-byte aligned, d_dense_xy / d_xy / d_row_ptr 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(unpack_desc_kernel, dim3(n_max / UR, n_img), dim3(256), 0, st, d_dense_desc,
                       d_row_ptr, d_count, n_max, desc_width, (long)n_rows_total, (uint32_t *)d_desc, d_count_out);
    hipLaunchKernelGGL(unpack_xy_kernel, dim3(slam_div_up(n_max, 256), n_img), dim3(256), 0, st, (const float2 *)d_dense_xy,
                       d_row_ptr, d_count, n_max, (long)n_rows_total, (float2 *)d_xy);
    SLAM_HIP_CHECK(hipGetLastError());
    return SLAM_OK;
}
