// Small fused "glue" kernels of the training iteration (gfx950).
//
// After the hot kernels were written, a CRNN iteration still contained ~110 framework launches of 2-10 us each --
// clone / fill / add / cast chains around the GRU biases, fill + reduce + copy triples behind every split-K weight
// gradient, slice copies that assemble dW_hh, flip + copy pairs that build the transposed convolution weights
// (profiles/r02_iteration_timeline_before.txt).  A dependent kernel boundary costs ~1.5-5 us on this chip whatever the
// kernel does (MI355X_MICROARCH.md, "boundary"), so each chain is replaced by ONE launch here.  All of them move a few
// hundred KB at most: latency-sized, not bandwidth-sized; they exist to remove boundaries.
//
// Reference lines they serve: nn.GRU's bias handling (model_crnn.py:65-72), the Linear / GRU weight gradients autograd
// computes for trainer.py:178, the data gradient of the encoder's 3x3 convolutions (model_crnn.py:5-17).
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

namespace {

constexpr int kGlueThreads = 256;

__device__ __forceinline__ float load_as_float(const void* p, long i, bool bf16) {
  return bf16 ? __uint_as_float(static_cast<unsigned>(static_cast<const unsigned short*>(p)[i]) << 16)
              : static_cast<const float*>(p)[i];
}
__device__ __forceinline__ void store_from_float(void* p, long i, bool bf16, float v) {
  if (bf16) static_cast<unsigned short*>(p)[i] = __bfloat16_as_ushort(__float2bfloat16(v));
  else static_cast<float*>(p)[i] = v;
}

// gi_bias[d][g][u] = b_ih[d][g][u] + (g < 2 ? b_hh[d][g][u] : 0)   (the r / z recurrent biases commute with the
// sigmoid argument: they ride on the input projection's bias);  b_hn[d][u] = b_hh[d][2][u]
__global__ __launch_bounds__(kGlueThreads) void gru_fold_bias_kernel(const float* __restrict__ b_ih,
                                                                     const float* __restrict__ b_hh, int h,
                                                                     void* __restrict__ gi_bias, int out_bf16,
                                                                     float* __restrict__ b_hn) {
  const int i = blockIdx.x * kGlueThreads + threadIdx.x;            // over [2][3][h]
  if (i >= 2 * 3 * h) return;
  const int u = i % h, g = (i / h) % 3, d = i / (3 * h);
  const float hh = b_hh[i];
  store_from_float(gi_bias, i, out_bf16 != 0, b_ih[i] + (g < 2 ? hh : 0.0f));
  if (g == 2) b_hn[d * h + u] = hh;
}

// The backward recurrence leaves per-tile sums of (da_r, da_z, da_n, da_n r) over (sequence, time):
// partial[tile][d][4][h].  db_ih[d] = (r, z, n) ; db_hh[d] = (r, z, n r): nn.GRU's bias gradients, in a fixed order.
__global__ __launch_bounds__(kGlueThreads) void gru_bias_grads_kernel(const float* __restrict__ partial, int tiles, int h,
                                                                      float* __restrict__ db_ih,
                                                                      float* __restrict__ db_hh) {
  const int i = blockIdx.x * kGlueThreads + threadIdx.x;            // over [2][4][h]
  if (i >= 2 * 4 * h) return;
  const int u = i % h, slot = (i / h) % 4, d = i / (4 * h);
  float s = 0.0f;
  for (int t = 0; t < tiles; ++t) s += partial[static_cast<long>(t) * 2 * 4 * h + i];
  if (slot < 2) {
    db_ih[(d * 3 + slot) * h + u] = s;
    db_hh[(d * 3 + slot) * h + u] = s;
  } else if (slot == 2) {
    db_ih[(d * 3 + 2) * h + u] = s;
  } else {
    db_hh[(d * 3 + 2) * h + u] = s;
  }
}

// out[i] = sum_c partial[c][i]  (fp32 accumulation, fixed order): the reduction behind a split-K product
template <int kVec>
__global__ __launch_bounds__(kGlueThreads) void sum_chunks_kernel(const void* __restrict__ partial, int in_bf16,
                                                                  int chunks, long count, void* __restrict__ out,
                                                                  int out_bf16) {
  const long i0 = (static_cast<long>(blockIdx.x) * kGlueThreads + threadIdx.x) * kVec;
  if (i0 >= count) return;
  float acc[kVec];
#pragma unroll
  for (int j = 0; j < kVec; ++j) acc[j] = 0.0f;
  for (int c = 0; c < chunks; ++c) {
    const long base = static_cast<long>(c) * count + i0;
    if (kVec == 8 && in_bf16) {
      const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(partial) + base);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __uint_as_float(w[j] << 16);
        acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < kVec; ++j)
        if (i0 + j < count) acc[j] += load_as_float(partial, base + j, in_bf16 != 0);
    }
  }
#pragma unroll
  for (int j = 0; j < kVec; ++j)
    if (i0 + j < count) store_from_float(out, i0 + j, out_bf16 != 0, acc[j]);
}

// partial[y][n] = sum over the rows of row block y of g[r][n]: stage 1 of a column sum of a tall matrix (the bias
// gradient of every nn.Linear: 8000 rows x 256..9072 columns).  The framework's reduce kernel takes 16 - 36 us for these
// shapes whatever the width (few, long threads); here the rows are cut into blocks so that ~1000 workgroups stream the
// matrix with 16-byte loads, and column_finish_kernel adds the row blocks in a fixed order (deterministic, fp32).
// 256 threads = 32 column groups of 8 columns x 8 row slots; grid (ceil(N / 256), row blocks).
__global__ __launch_bounds__(kGlueThreads) void column_partials_kernel(const void* __restrict__ g, int in_bf16, long rows,
                                                                       int n_cols, int rows_per_block,
                                                                       float* __restrict__ partial) {
  __shared__ float red[8][257];
  const int tid = threadIdx.x;
  const int cg = tid & 31, slot = tid >> 5;
  const int col0 = blockIdx.x * 256 + cg * 8;
  const long r0 = static_cast<long>(blockIdx.y) * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  if (col0 < n_cols) {
    auto add_row = [&](long r) {
      const long base = r * n_cols + col0;
      if (in_bf16) {
        const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(g) + base);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += __uint_as_float(w[j] << 16);
          acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
      } else {
        const float4* q = reinterpret_cast<const float4*>(static_cast<const float*>(g) + base);
        const float4 a = q[0], b = q[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
    };
    long r = r0 + slot;
    for (; r + 24 < r1; r += 32) {                          // four independent 16-byte loads in flight
      add_row(r);
      add_row(r + 8);
      add_row(r + 16);
      add_row(r + 24);
    }
    for (; r < r1; r += 8) add_row(r);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[slot][cg * 8 + j] = acc[j];
  __syncthreads();
  const int col = blockIdx.x * 256 + tid;
  if (col < n_cols) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += red[k][tid];
    partial[static_cast<long>(blockIdx.y) * n_cols + col] = sum;
  }
}

// out[n] = sum over the row blocks of partial[y][n] (fixed order): stage 2.  256 threads = 16 columns x 16 row slots
// (narrow blocks: a 256-column matrix still gives 16 workgroups, and a thread adds at most 8 partial rows).
__global__ __launch_bounds__(kGlueThreads) void column_finish_kernel(const float* __restrict__ partial, int blocks, int n_cols,
                                                                     void* __restrict__ out, int out_bf16) {
  __shared__ float red[16][17];
  const int tid = threadIdx.x;
  const int c = tid & 15, slot = tid >> 4;
  const int col = blockIdx.x * 16 + c;
  float sum = 0.0f;
  if (col < n_cols)
    for (int y = slot; y < blocks; y += 16) sum += partial[static_cast<long>(y) * n_cols + col];
  red[slot][c] = sum;
  __syncthreads();
  if (slot == 0 && col < n_cols) {
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += red[k][c];
    store_from_float(out, col, out_bf16 != 0, total);
  }
}

// ---- multi-tensor forms of the two reductions above -------------------------------------------------------------------
// A Conformer iteration has 23 Linear layers (model_conformer.py:19-41, 98-127), a ResNet50-Conformer one 43: their
// backward passes issued 3 latency-sized launches each (chunk sum of the split-K weight gradient, column partials +
// finish of the bias gradient): 69 launches, 455 us of a 5.1 ms iteration, for ~40 MB of reads
// (profiles/r02_conformer_timed_region_stats.csv).  Nothing consumes a weight or bias gradient before the optimiser, so
// the captured step queues the reductions during the backward pass and runs them here: ONE launch for every chunk sum,
// TWO (partials, finish) for every column sum, descriptors by value in the kernel arguments (the seld_multi_cast pattern).

constexpr int kMultiSum = 64;        // descriptors per launch: 64 x 40 B
constexpr int kMultiCol = 40;        // 40 x 56 B

struct SumDesc {
  const void* partial;
  void* out;
  long count;
  int chunks;
  int flags;                         // bit 0: partial is bf16, bit 1: out is bf16, bit 2: 8-wide (count % 8 == 0, aligned)
  int first_block;                   // index of this tensor's first workgroup (work list: no empty workgroups)
  int pad;
};
struct SumBatch { SumDesc d[kMultiSum]; int n; };

__global__ __launch_bounds__(kGlueThreads) void multi_sum_chunks_kernel(SumBatch b) {
  int t = 0;
  while (t + 1 < b.n && static_cast<int>(blockIdx.x) >= b.d[t + 1].first_block) ++t;      // uniform
  const SumDesc& d = b.d[t];
  const bool in_bf16 = d.flags & 1, out_bf16 = d.flags & 2, wide = d.flags & 4;
  const long i0 = (static_cast<long>(static_cast<int>(blockIdx.x) - d.first_block) * kGlueThreads + threadIdx.x) * 8;
  if (i0 >= d.count) return;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  for (int c = 0; c < d.chunks; ++c) {
    const long base = static_cast<long>(c) * d.count + i0;
    if (wide && in_bf16) {
      const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(d.partial) + base);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __uint_as_float(w[j] << 16);
        acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i0 + j < d.count) acc[j] += load_as_float(d.partial, base + j, in_bf16);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (i0 + j < d.count) store_from_float(d.out, i0 + j, out_bf16, acc[j]);
}

struct ColDesc {
  const void* g;                     // [rows][n_cols]
  void* out;                         // [n_cols]
  float* partial;                    // [row_blocks][n_cols] scratch
  int rows, n_cols;
  int row_blocks, rows_per_block;
  int first_item;                    // index of this matrix's first (column block, row block) work item (partials kernel)
  int first_finish;                  // index of its first 16-column workgroup (finish kernel)
  int flags;                         // bit 0: g is bf16, bit 1: out is bf16
  int pad;
};
struct ColBatch { ColDesc d[kMultiCol]; int n; };

// Stage 1, one workgroup per (matrix, 256-column block, row block): partial column sums of its slab exactly as
// column_partials_kernel forms them.  Stage 2 (multi_column_finish_kernel, the kernel boundary makes the partial rows
// visible: a device-scope fence inside one kernel costs an L2 write-back per workgroup on this multi-XCD chip -- the
// single-kernel "last workgroup finishes" form was measured at 375 us for a Conformer's 23 matrices) adds the row blocks
// in a fixed order, like column_finish_kernel.
__global__ __launch_bounds__(kGlueThreads) void multi_column_partials_kernel(ColBatch b) {
  __shared__ float red[8][257];
  const int tid = threadIdx.x;
  int t = 0;
  while (t + 1 < b.n && static_cast<int>(blockIdx.x) >= b.d[t + 1].first_item) ++t;      // uniform
  const ColDesc& d = b.d[t];
  const int item = static_cast<int>(blockIdx.x) - d.first_item;
  const int col_block = item / d.row_blocks, row_block = item - col_block * d.row_blocks;
  const bool in_bf16 = d.flags & 1;
  const int cg = tid & 31, slot = tid >> 5;
  const int col0 = col_block * 256 + cg * 8;
  const long r0 = static_cast<long>(row_block) * d.rows_per_block;
  const long r1 = r0 + d.rows_per_block < d.rows ? r0 + d.rows_per_block : d.rows;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  if (col0 < d.n_cols) {
    auto add_row = [&](long r) {
      const long base = r * d.n_cols + col0;
      if (in_bf16) {
        const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(d.g) + base);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += __uint_as_float(w[j] << 16);
          acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
        }
      } else {
        const float4* q = reinterpret_cast<const float4*>(static_cast<const float*>(d.g) + base);
        const float4 a = q[0], c = q[1];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += c.x; acc[5] += c.y; acc[6] += c.z; acc[7] += c.w;
      }
    };
    long r = r0 + slot;
    for (; r + 24 < r1; r += 32) {
      add_row(r);
      add_row(r + 8);
      add_row(r + 16);
      add_row(r + 24);
    }
    for (; r < r1; r += 8) add_row(r);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[slot][cg * 8 + j] = acc[j];
  __syncthreads();
  const int col = col_block * 256 + tid;
  if (col < d.n_cols) {
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += red[k][tid];
    d.partial[static_cast<long>(row_block) * d.n_cols + col] = sum;
  }
}

__global__ __launch_bounds__(kGlueThreads) void multi_column_finish_kernel(ColBatch b) {
  __shared__ float red[16][17];
  const int tid = threadIdx.x;
  int t = 0;
  while (t + 1 < b.n && static_cast<int>(blockIdx.x) >= b.d[t + 1].first_finish) ++t;    // uniform
  const ColDesc& d = b.d[t];
  const int c = tid & 15, slot = tid >> 4;
  const int col = (static_cast<int>(blockIdx.x) - d.first_finish) * 16 + c;
  float sum = 0.0f;
  if (col < d.n_cols)
    for (int y = slot; y < d.row_blocks; y += 16) sum += d.partial[static_cast<long>(y) * d.n_cols + col];
  red[slot][c] = sum;
  __syncthreads();
  if (slot == 0 && col < d.n_cols) {
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) total += red[k][c];
    store_from_float(d.out, col, (d.flags & 2) != 0, total);
  }
}

// dW_hh[d][g][u][k] from the two wasteful-but-well-shaped products the host forms (seld_gru._BiGRULayer.backward):
//   p_gi[c][d][g][u][d'][k] = chunk c of  dgi^T h_prev      (g = 0..2; only d' == d and g < 2 are wanted)
//   p_n [c][d][u][d'][k]    = chunk c of  (da_n r)^T h_prev (only d' == d)
// summed over the chunks in fp32 and written straight into the parameter-shaped gradient.
__global__ __launch_bounds__(kGlueThreads) void gru_dwhh_finish_kernel(const void* __restrict__ p_gi,
                                                                       const void* __restrict__ p_n, int in_bf16,
                                                                       int chunks, int h, void* __restrict__ dw_hh,
                                                                       int out_bf16) {
  const long i = static_cast<long>(blockIdx.x) * kGlueThreads + threadIdx.x;      // over [2][3][h][h]
  const long total = 2L * 3 * h * h;
  if (i >= total) return;
  const int k = static_cast<int>(i % h), u = static_cast<int>((i / h) % h);
  const int g = static_cast<int>((i / (static_cast<long>(h) * h)) % 3), d = static_cast<int>(i / (3L * h * h));
  float s = 0.0f;
  if (g < 2) {
    const long per = 2L * 3 * h * 2 * h;
    const long off = ((((static_cast<long>(d) * 3 + g) * h + u) * 2 + d) * h) + k;
    for (int c = 0; c < chunks; ++c) s += load_as_float(p_gi, c * per + off, in_bf16 != 0);
  } else {
    const long per = 2L * h * 2 * h;
    const long off = (((static_cast<long>(d) * h + u) * 2 + d) * h) + k;
    for (int c = 0; c < chunks; ++c) s += load_as_float(p_n, c * per + off, in_bf16 != 0);
  }
  store_from_float(dw_hh, i, out_bf16 != 0, s);
}

// Transposed, flipped 3x3 weights for "data gradient as a forward convolution" (model_crnn._Conv3x3):
//   wt[i][o][2-r][2-s] = w[o][i][r][s],  both tensors in channels-last memory: w is [O][3][3][I], wt is [I][3][3][O].
template <typename T>
__global__ __launch_bounds__(kGlueThreads) void conv_weight_flip_transpose_kernel(const T* __restrict__ w, int O, int I,
                                                                                  T* __restrict__ wt) {
  const long idx = static_cast<long>(blockIdx.x) * kGlueThreads + threadIdx.x;    // over wt memory [I][3][3][O]
  const long total = static_cast<long>(O) * I * 9;
  if (idx >= total) return;
  const int o = static_cast<int>(idx % O);
  const int rs = static_cast<int>((idx / O) % 9);
  const int i = static_cast<int>(idx / (9L * O));
  wt[idx] = w[(static_cast<long>(o) * 9 + (8 - rs)) * I + i];
}

}  // namespace

}  // namespace seld

extern "C" {

int seld_gru_fold_bias(const float* b_ih, const float* b_hh, int64_t H, void* gi_bias, int out_is_bf16, float* b_hn,
                       void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (H <= 0 || !b_ih || !b_hh || !gi_bias || !b_hn) return fail(kErrInvalidArgument, "seld_gru_fold_bias: bad argument");
  const int n = static_cast<int>(6 * H);
  hipLaunchKernelGGL(gru_fold_bias_kernel, dim3((n + kGlueThreads - 1) / kGlueThreads), dim3(kGlueThreads), 0,
                     static_cast<hipStream_t>(stream_), b_ih, b_hh, static_cast<int>(H), gi_bias, out_is_bf16, b_hn);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gru_bias_grads(const float* partial, int64_t tiles, int64_t H, float* db_ih, float* db_hh, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (H <= 0 || tiles <= 0 || !partial || !db_ih || !db_hh)
    return fail(kErrInvalidArgument, "seld_gru_bias_grads: bad argument");
  const int n = static_cast<int>(8 * H);
  hipLaunchKernelGGL(gru_bias_grads_kernel, dim3((n + kGlueThreads - 1) / kGlueThreads), dim3(kGlueThreads), 0,
                     static_cast<hipStream_t>(stream_), partial, static_cast<int>(tiles), static_cast<int>(H), db_ih,
                     db_hh);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_sum_chunks(const void* partial, int in_is_bf16, int64_t chunks, int64_t count, void* out, int out_is_bf16,
                    void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (chunks <= 0 || count <= 0 || !partial || !out) return fail(kErrInvalidArgument, "seld_sum_chunks: bad argument");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool wide = in_is_bf16 && count % 8 == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
  if (wide) {
    const long threads = count / 8;
    hipLaunchKernelGGL(sum_chunks_kernel<8>, dim3(static_cast<unsigned>((threads + kGlueThreads - 1) / kGlueThreads)),
                       dim3(kGlueThreads), 0, stream, partial, in_is_bf16, static_cast<int>(chunks),
                       static_cast<long>(count), out, out_is_bf16);
  } else {
    hipLaunchKernelGGL(sum_chunks_kernel<1>, dim3(static_cast<unsigned>((count + kGlueThreads - 1) / kGlueThreads)),
                       dim3(kGlueThreads), 0, stream, partial, in_is_bf16, static_cast<int>(chunks),
                       static_cast<long>(count), out, out_is_bf16);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int64_t seld_column_sums_blocks(int64_t rows, int64_t n_cols) {
  const long col_blocks = (n_cols + 255) / 256;
  long rb = 2048 / col_blocks;
  if (rb < 8) rb = 8;
  if (rb > 128) rb = 128;
  if (rb > rows) rb = rows;
  return rb;
}

int seld_column_sums(const void* g, int in_is_bf16, int64_t rows, int64_t n_cols, float* partial, void* out, int out_is_bf16,
                     void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (rows <= 0 || n_cols <= 0 || !g || !partial || !out) return fail(kErrInvalidArgument, "seld_column_sums: bad argument");
  if (n_cols % 8 != 0 || (reinterpret_cast<uintptr_t>(g) & 15) != 0)
    return fail(kErrUnsupported, "seld_column_sums: the column count must be a multiple of 8 and the matrix 16-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long rb = seld_column_sums_blocks(rows, n_cols);
  const long rows_per_block = (rows + rb - 1) / rb;
  hipLaunchKernelGGL(column_partials_kernel, dim3(static_cast<unsigned>((n_cols + 255) / 256), static_cast<unsigned>(rb)),
                     dim3(kGlueThreads), 0, stream, g, in_is_bf16, static_cast<long>(rows), static_cast<int>(n_cols),
                     static_cast<int>(rows_per_block), partial);
  hipLaunchKernelGGL(column_finish_kernel, dim3(static_cast<unsigned>((n_cols + 15) / 16)), dim3(kGlueThreads), 0, stream,
                     partial, static_cast<int>(rb), static_cast<int>(n_cols), out, out_is_bf16);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

// Row blocks of one matrix inside a multi-tensor launch: half of the stand-alone kernel's (the launch as a whole fills
// the chip).
static long multi_column_row_blocks(int64_t rows, int64_t n_cols) {
  long rb = seld_column_sums_blocks(rows, n_cols) / 2;
  if (rb < 8) rb = 8;
  if (rb > rows) rb = rows;
  return rb;
}

int seld_multi_sum_chunks(const void* const* partial, void* const* out, const int64_t* counts, const int32_t* chunks,
                          const int32_t* flags, int count, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (count < 0) return fail(kErrInvalidArgument, "seld_multi_sum_chunks: negative count");
  if (count == 0) return kOk;
  if (!partial || !out || !counts || !chunks || !flags) return fail(kErrInvalidArgument, "seld_multi_sum_chunks: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  for (int first = 0; first < count; first += kMultiSum) {
    SumBatch b;
    const int here = count - first < kMultiSum ? count - first : kMultiSum;
    long blocks = 0;
    for (int i = 0; i < here; ++i) {
      const int k = first + i;
      if (counts[k] <= 0 || chunks[k] <= 0 || !partial[k] || !out[k])
        return fail(kErrInvalidArgument, "seld_multi_sum_chunks: bad descriptor");
      int f = flags[k] & 3;
      if ((f & 1) && counts[k] % 8 == 0 && (reinterpret_cast<uintptr_t>(partial[k]) & 15) == 0) f |= 4;
      b.d[i] = SumDesc{partial[k], out[k], static_cast<long>(counts[k]), chunks[k], f, static_cast<int>(blocks), 0};
      blocks += ((counts[k] + 7) / 8 + kGlueThreads - 1) / kGlueThreads;
      if (blocks >= (1L << 31)) return fail(kErrUnsupported, "seld_multi_sum_chunks: too many elements for one launch");
    }
    b.n = here;
    hipLaunchKernelGGL(multi_sum_chunks_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kGlueThreads), 0, stream, b);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_multi_column_sums(const void* const* g, void* const* out, const int64_t* rows, const int64_t* n_cols,
                           const int32_t* flags, int count, float* partial, int64_t partial_floats, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (count < 0) return fail(kErrInvalidArgument, "seld_multi_column_sums: negative count");
  if (count == 0) return kOk;
  if (!g || !out || !rows || !n_cols || !flags || !partial)
    return fail(kErrInvalidArgument, "seld_multi_column_sums: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  long used_floats = 0;
  for (int first = 0; first < count; first += kMultiCol) {
    ColBatch b;
    const int here = count - first < kMultiCol ? count - first : kMultiCol;
    long items = 0, finish = 0;
    for (int i = 0; i < here; ++i) {
      const int k = first + i;
      if (rows[k] <= 0 || n_cols[k] <= 0 || rows[k] >= (1L << 31) || n_cols[k] >= (1L << 31) || !g[k] || !out[k])
        return fail(kErrInvalidArgument, "seld_multi_column_sums: bad descriptor");
      if (n_cols[k] % 8 != 0 || (reinterpret_cast<uintptr_t>(g[k]) & 15) != 0)
        return fail(kErrUnsupported, "seld_multi_column_sums: the column count must be a multiple of 8 and the matrix 16-byte aligned");
      const long rb = multi_column_row_blocks(rows[k], n_cols[k]);
      const long col_blocks = (n_cols[k] + 255) / 256;
      if (used_floats + rb * n_cols[k] > partial_floats)
        return fail(kErrInvalidArgument, "seld_multi_column_sums: scratch too small (seld_multi_column_sums_scratch)");
      b.d[i] = ColDesc{g[k], out[k], partial + used_floats, static_cast<int>(rows[k]), static_cast<int>(n_cols[k]),
                       static_cast<int>(rb), static_cast<int>((rows[k] + rb - 1) / rb), static_cast<int>(items),
                       static_cast<int>(finish), flags[k] & 3, 0};
      used_floats += rb * n_cols[k];
      items += col_blocks * rb;
      finish += (n_cols[k] + 15) / 16;
      if (items >= (1L << 31) || finish >= (1L << 31)) return fail(kErrUnsupported, "seld_multi_column_sums: too much work for one launch");
    }
    b.n = here;
    hipLaunchKernelGGL(multi_column_partials_kernel, dim3(static_cast<unsigned>(items)), dim3(kGlueThreads), 0, stream, b);
    hipLaunchKernelGGL(multi_column_finish_kernel, dim3(static_cast<unsigned>(finish)), dim3(kGlueThreads), 0, stream, b);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_multi_column_sums_scratch(const int64_t* rows, const int64_t* n_cols, int count, int64_t* partial_floats) {
  using namespace seld;
  if (count < 0 || (count > 0 && (!rows || !n_cols)) || !partial_floats)
    return fail(kErrInvalidArgument, "seld_multi_column_sums_scratch: bad argument");
  long f = 0;
  for (int k = 0; k < count; ++k) {
    if (rows[k] <= 0 || n_cols[k] <= 0) return fail(kErrInvalidArgument, "seld_multi_column_sums_scratch: bad extent");
    f += multi_column_row_blocks(rows[k], n_cols[k]) * n_cols[k];
  }
  *partial_floats = f;
  return kOk;
}

int seld_gru_dwhh_finish(const void* p_gi, const void* p_n, int in_is_bf16, int64_t chunks, int64_t H, void* dw_hh,
                         int out_is_bf16, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (chunks <= 0 || H <= 0 || !p_gi || !p_n || !dw_hh) return fail(kErrInvalidArgument, "seld_gru_dwhh_finish: bad argument");
  const long total = 6L * H * H;
  hipLaunchKernelGGL(gru_dwhh_finish_kernel, dim3(static_cast<unsigned>((total + kGlueThreads - 1) / kGlueThreads)),
                     dim3(kGlueThreads), 0, static_cast<hipStream_t>(stream_), p_gi, p_n, in_is_bf16,
                     static_cast<int>(chunks), static_cast<int>(H), dw_hh, out_is_bf16);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_conv_weight_flip_transpose(const void* w, int elem_bytes, int64_t O, int64_t I, void* wt, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (O <= 0 || I <= 0 || !w || !wt) return fail(kErrInvalidArgument, "seld_conv_weight_flip_transpose: bad argument");
  if (elem_bytes != 2 && elem_bytes != 4) return fail(kErrUnsupported, "seld_conv_weight_flip_transpose: 2- or 4-byte elements");
  const long total = O * I * 9;
  const dim3 grid(static_cast<unsigned>((total + kGlueThreads - 1) / kGlueThreads));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (elem_bytes == 2)
    hipLaunchKernelGGL(conv_weight_flip_transpose_kernel<unsigned short>, grid, dim3(kGlueThreads), 0, stream,
                       static_cast<const unsigned short*>(w), static_cast<int>(O), static_cast<int>(I),
                       static_cast<unsigned short*>(wt));
  else
    hipLaunchKernelGGL(conv_weight_flip_transpose_kernel<float>, grid, dim3(kGlueThreads), 0, stream,
                       static_cast<const float*>(w), static_cast<int>(O), static_cast<int>(I), static_cast<float*>(wt));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
