// row_blocks.hip -- the small weight-sized rearrangements around the message-passing layers as library launches.
//
// The edge-function layers split the first Dense layer of phi by ROW BLOCKS of its weight matrix and recombine the blocks with
// signs: ExplicitEdgeConv [wa; -wc] / [wb; wc] (/root/reference/src/layers.jl:106), VMHConv [wa - wb; -wc] / [wb; wc] (:316),
// MPPDEConv [wa; wc; we] / [wb; -wc] and the edge block wd (:409-410), GNOConv [wa] / [wb] / [wd] (:523); GNOConv's reassociated
// form needs W2 of phi's last layer transposed (:527-530).  Written with torch these are slices, cat, neg, sub and permuted copies
// -- and, in the pullback, zero-fills and adds of slice gradients: a dozen tiny kernels per layer call.  Here: ONE launch builds
// all recombined matrices of a layer, ONE launch scatters their gradients back into the gradient of the weight, and a tiled
// transpose.  No atomics: every output element is owned by one thread which walks the (<= 16) segments.
#include "common.h"

namespace ngpde {
namespace {

constexpr int kMaxSeg = 16;
struct SegK {
  int n;
  int out[kMaxSeg];       // which output matrix
  int dst_row0[kMaxSeg];  // first row in that output
  int src_row0[kMaxSeg];  // first row of the source
  int n_rows[kMaxSeg];
  float sign[kMaxSeg];
  float *dst[4];          // forward: the outputs; backward: their gradients
  int dst_rows[4];
};

// dst_o[r][c] = sum over the segments of output o that cover row r of sign * src[src_row0 + r - dst_row0][c]
__global__ void row_blocks_gather_kernel(int width, const float *__restrict__ src, const SegK k, int total_rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_rows * width) return;
  int row = idx / width;
  const int c = idx % width;
  int o = 0;
  while (o < 3 && row >= k.dst_rows[o]) row -= k.dst_rows[o], ++o;
  float v = 0.f;
  for (int s = 0; s < k.n; ++s)
    if (k.out[s] == o && row >= k.dst_row0[s] && row < k.dst_row0[s] + k.n_rows[s])
      v += k.sign[s] * src[(size_t)(k.src_row0[s] + row - k.dst_row0[s]) * width + c];
  k.dst[o][(size_t)row * width + c] = v;
}

// dsrc[r][c] = sum over the segments that read source row r of sign * ddst_o[dst_row0 + r - src_row0][c]  (0 for untouched rows)
__global__ void row_blocks_scatter_kernel(int width, int src_rows, float *__restrict__ dsrc, const SegK k) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= src_rows * width) return;
  const int row = idx / width, c = idx % width;
  float v = 0.f;
  for (int s = 0; s < k.n; ++s)
    if (k.dst[k.out[s]] && row >= k.src_row0[s] && row < k.src_row0[s] + k.n_rows[s])
      v += k.sign[s] * k.dst[k.out[s]][(size_t)(k.dst_row0[s] + row - k.src_row0[s]) * width + c];
  dsrc[idx] = v;
}

// dst[c][r] = src[r][c], 32 x 32 tiles through LDS (+1 padding)
__global__ void transpose_kernel(int rows, int cols, const float *__restrict__ src, float *__restrict__ dst) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = src[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[threadIdx.x][j];
  }
}

// out[i][:] = x[i][:] * scale[i]
__global__ void rows_scale_kernel(int64_t n, int d, const float *__restrict__ x, const float *__restrict__ scale, float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * d) return;
  out[idx] = x[idx] * scale[idx / d];
}

int32_t fill_segments(const char *fn, int32_t n_seg, const int32_t *out_index, const int32_t *dst_row0, const int32_t *src_row0,
                      const int32_t *n_rows, const float *sign, int32_t n_out, float *const *mats, const int32_t *mat_rows, SegK *k) {
  NGPDE_REQUIRE(n_seg >= 1 && n_seg <= kMaxSeg, NGPDE_ERR_INVALID_ARGUMENT, "%s: 1..%d segments (got %d)", fn, kMaxSeg, n_seg);
  NGPDE_REQUIRE(n_out >= 1 && n_out <= 4, NGPDE_ERR_INVALID_ARGUMENT, "%s: 1..4 output matrices (got %d)", fn, n_out);
  NGPDE_REQUIRE(out_index && dst_row0 && src_row0 && n_rows && sign && mats && mat_rows, NGPDE_ERR_INVALID_ARGUMENT, "%s: NULL argument", fn);
  k->n = n_seg;
  for (int s = 0; s < n_seg; ++s) {
    NGPDE_REQUIRE(out_index[s] >= 0 && out_index[s] < n_out && n_rows[s] >= 0 && dst_row0[s] >= 0 && src_row0[s] >= 0 &&
                      dst_row0[s] + n_rows[s] <= mat_rows[out_index[s]],
                  NGPDE_ERR_DIMENSION_MISMATCH, "%s: DimensionMismatch: segment %d does not fit its output", fn, s);
    k->out[s] = out_index[s]; k->dst_row0[s] = dst_row0[s]; k->src_row0[s] = src_row0[s]; k->n_rows[s] = n_rows[s]; k->sign[s] = sign[s];
  }
  for (int o = 0; o < 4; ++o) {
    k->dst[o] = o < n_out ? mats[o] : nullptr;
    k->dst_rows[o] = o < n_out ? mat_rows[o] : 0;
  }
  return NGPDE_OK;
}

}  // namespace
}  // namespace ngpde

using namespace ngpde;

namespace ngpde {
namespace {
// gather: dst[o][i][:] = src[o][index[i]][:]; scatter: dst[o][index[i]][:] = src[o][i][:] (dst zeroed beforehand)
__global__ void rows_index_kernel(int64_t outer, int64_t n_src, int64_t n_idx, int d, const int64_t *__restrict__ index,
                                  const float *__restrict__ src, float *__restrict__ dst, int scatter) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, total = outer * n_idx * d;
  if (k >= total) return;
  const int c = (int)(k % d);
  const int64_t i = (k / d) % n_idx, o = k / ((int64_t)d * n_idx), j = index[i];
  // an entry outside [0, n_rows) names no row: the gather leaves a zero row, the scatter writes nothing (never an out-of-bounds access)
  const bool in_range = (uint64_t)j < (uint64_t)n_src;
  if (scatter) {
    if (in_range) dst[(o * n_src + j) * d + c] = src[(o * n_idx + i) * d + c];
  } else {
    dst[(o * n_idx + i) * d + c] = in_range ? src[(o * n_src + j) * d + c] : 0.f;
  }
}
}  // namespace
}  // namespace ngpde

extern "C" {

int32_t ngpde_row_blocks_gather(int32_t width, int32_t src_rows, const float *src, int32_t n_seg, const int32_t *out_index,
                                const int32_t *dst_row0, const int32_t *src_row0, const int32_t *n_rows, const float *sign, int32_t n_out,
                                float *const *outs, const int32_t *out_rows, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegK k;
  int32_t st = fill_segments("ngpde_row_blocks_gather", n_seg, out_index, dst_row0, src_row0, n_rows, sign, n_out, outs, out_rows, &k);
  if (st) return st;
  NGPDE_REQUIRE(width > 0 && src != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_row_blocks_gather: bad arguments");
  int total = 0;
  for (int o = 0; o < n_out; ++o) {
    NGPDE_REQUIRE(outs[o] != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_row_blocks_gather: output %d is NULL", o);
    total += out_rows[o];
  }
  for (int s = 0; s < n_seg; ++s)
    NGPDE_REQUIRE(src_row0[s] + n_rows[s] <= src_rows, NGPDE_ERR_DIMENSION_MISMATCH,
                  "ngpde_row_blocks_gather: DimensionMismatch: segment %d reads beyond the %d source rows", s, src_rows);
  if (total == 0) return NGPDE_OK;
  const int64_t count = (int64_t)total * width;
  hipLaunchKernelGGL(row_blocks_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, width, src, k, total);
  NGPDE_LAUNCH_CHECK("row_blocks_gather_kernel");
  return NGPDE_OK;
}

int32_t ngpde_row_blocks_scatter(int32_t width, int32_t src_rows, float *dsrc, int32_t n_seg, const int32_t *out_index,
                                 const int32_t *dst_row0, const int32_t *src_row0, const int32_t *n_rows, const float *sign, int32_t n_out,
                                 float *const *douts, const int32_t *out_rows, ngpde_stream_t stream) {
  NGPDE_RANGE();
  SegK k;
  int32_t st = fill_segments("ngpde_row_blocks_scatter", n_seg, out_index, dst_row0, src_row0, n_rows, sign, n_out, douts, out_rows, &k);
  if (st) return st;
  NGPDE_REQUIRE(width > 0 && src_rows >= 0 && dsrc != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_row_blocks_scatter: bad arguments");
  for (int s = 0; s < n_seg; ++s)
    NGPDE_REQUIRE(src_row0[s] + n_rows[s] <= src_rows, NGPDE_ERR_DIMENSION_MISMATCH,
                  "ngpde_row_blocks_scatter: DimensionMismatch: segment %d writes beyond the %d source rows", s, src_rows);
  if (src_rows == 0) return NGPDE_OK;
  const int64_t count = (int64_t)src_rows * width;
  hipLaunchKernelGGL(row_blocks_scatter_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, width, src_rows, dsrc, k);
  NGPDE_LAUNCH_CHECK("row_blocks_scatter_kernel");
  return NGPDE_OK;
}

int32_t ngpde_transpose(int32_t rows, int32_t cols, const float *src, float *dst, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(rows >= 0 && cols >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_transpose: negative size");
  if (rows == 0 || cols == 0) return NGPDE_OK;
  NGPDE_REQUIRE(src && dst && src != dst, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_transpose: src / dst NULL or aliased");
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, rows, cols, src, dst);
  NGPDE_LAUNCH_CHECK("transpose_kernel");
  return NGPDE_OK;
}

int32_t ngpde_rows_scale(int64_t n, int32_t d, const float *x, const float *scale, float *out, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(n >= 0 && d > 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rows_scale: bad sizes");
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(x && scale && out, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rows_scale: NULL argument");
  hipLaunchKernelGGL(rows_scale_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d, x, scale, out);
  NGPDE_LAUNCH_CHECK("rows_scale_kernel");
  return NGPDE_OK;
}

int32_t ngpde_rows_index(int64_t outer, int64_t n_rows, int64_t n_index, int32_t d, const int64_t *index, const float *src, float *dst,
                         int32_t scatter, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(outer >= 0 && n_rows >= 0 && n_index >= 0 && d > 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rows_index: bad sizes");
  if (scatter && outer * n_rows > 0) {
    NGPDE_REQUIRE(dst != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rows_index: dst is NULL");
    int32_t st = launch_zero(dst, (size_t)(outer * n_rows * d) * sizeof(float), (hipStream_t)stream);   // rows no index names are zero
    if (st) return st;
  }
  const int64_t total = outer * n_index * d;
  if (total == 0) return NGPDE_OK;
  NGPDE_REQUIRE(index && src && dst, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_rows_index: NULL argument");
  hipLaunchKernelGGL(ngpde::rows_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, outer, n_rows, n_index, d, index,
                     src, dst, scatter);
  NGPDE_LAUNCH_CHECK("rows_index_kernel");
  return NGPDE_OK;
}

}  // extern "C"
