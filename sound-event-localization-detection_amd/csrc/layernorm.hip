// gfx950 LayerNorm [-> ReLU] over the last dimension, forward and backward, activations in their own dtype.
//
// Replaces the head's  LayerNorm(512) -> ReLU  of model_crnn.py:77-83 (and the LayerNorm(1024) heads of
// model_conformer.py / resnet50_model.py).  Under bf16 autocast the stock path casts the activation to fp32, runs
// LayerNorm, ReLU and Dropout on fp32 tensors and casts back for the next Linear: six framework kernels forward and
// seven backward that each move 4-byte elements.  Here: one kernel forward (read x, write y, in the activation's
// dtype; statistics and arithmetic in fp32, two-pass variance as in the framework) and one kernel + one small
// reduction backward.  HBM-bound: 2 elements moved per element forward (x, y), 3 backward (x, dy, dx).
//
// One wavefront per row: D = 64 lanes x kChunks chunks x V elements, the row lives in registers between the
// statistics and the apply phase; kChunks x V = D / 64 in {4, 8, 16, 32} (D = 256, 512, 1024, 2048).
// The backward kernel walks rows with a grid-sized stride so that each wavefront keeps running column sums of
// d(weight), d(bias) in registers; the 4 wavefronts of a block add theirs through LDS and write one partial row per
// block, a second kernel adds the partial rows in a fixed order (deterministic, no atomics; all loads of a thread in
// flight at once: a serial walk over 512 partial rows cost 42 us, more than everything this file replaces).
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

namespace {

constexpr int kLnWaves = 4;           // wavefronts (rows in flight) per block
constexpr int kLnMaxBlocks = 512;     // backward: partial rows of the column sums

template <typename T, int V> struct RowIo;

template <int V> struct RowIo<__hip_bfloat16, V> {
  static_assert(V == 4 || V == 8, "8- or 16-byte pieces");
  __device__ static void load(const void* base, long index, float (&v)[V]) {
    if (V == 8) {
      const uint4 r = static_cast<const uint4*>(base)[index];
      const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      }
    } else {
      const uint2 r = static_cast<const uint2*>(base)[index];
      v[0] = __uint_as_float(r.x << 16);
      v[1] = __uint_as_float(r.x & 0xffff0000u);
      v[2] = __uint_as_float(r.y << 16);
      v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
  }
  __device__ static unsigned pack(float lo, float hi) {
    const __hip_bfloat16 a = __float2bfloat16(lo), b = __float2bfloat16(hi);      // round to nearest even
    return static_cast<unsigned>(*reinterpret_cast<const unsigned short*>(&a)) |
           (static_cast<unsigned>(*reinterpret_cast<const unsigned short*>(&b)) << 16);
  }
  __device__ static void store(void* base, long index, const float (&v)[V]) {
    if (V == 8) {
      static_cast<uint4*>(base)[index] = make_uint4(pack(v[0], v[1]), pack(v[2], v[3]), pack(v[4], v[5]), pack(v[6], v[7]));
    } else {
      static_cast<uint2*>(base)[index] = make_uint2(pack(v[0], v[1]), pack(v[2], v[3]));
    }
  }
};

template <int V> struct RowIo<float, V> {
  __device__ static void load(const void* base, long index, float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < V / 4; ++i) {
      const float4 r = static_cast<const float4*>(base)[index * (V / 4) + i];
      v[4 * i] = r.x;
      v[4 * i + 1] = r.y;
      v[4 * i + 2] = r.z;
      v[4 * i + 3] = r.w;
    }
  }
  __device__ static void store(void* base, long index, const float (&v)[V]) {
#pragma unroll
    for (int i = 0; i < V / 4; ++i)
      static_cast<float4*>(base)[index * (V / 4) + i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// element (chunk c, lane l, j) of a row is column (c * 64 + l) * V + j: every chunk is one contiguous run per wavefront
template <typename T, int V, int kChunks, bool kRelu>
__global__ __launch_bounds__(64 * kLnWaves) void layernorm_forward_kernel(const void* __restrict__ x, long rows,
                                                                          const float* __restrict__ weight,
                                                                          const float* __restrict__ bias, float eps,
                                                                          void* __restrict__ y,
                                                                          float2* __restrict__ mean_rstd) {
  constexpr int kD = 64 * V * kChunks;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = static_cast<long>(blockIdx.x) * kLnWaves + wave;
  if (row >= rows) return;
  float v[kChunks][V];
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    RowIo<T, V>::load(x, row * (kD / V) + c * 64 + lane, v[c]);
#pragma unroll
    for (int j = 0; j < V; ++j) s += v[c][j];
  }
  const float mean = wave_sum(s) * (1.0f / kD);
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < kChunks; ++c)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float d = v[c][j] - mean;
      q = fmaf(d, d, q);
    }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / kD) + eps);
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    float w[V], b[V], o[V];
    RowIo<float, V>::load(weight, c * 64 + lane, w);
    RowIo<float, V>::load(bias, c * 64 + lane, b);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float z = fmaf((v[c][j] - mean) * rstd, w[j], b[j]);
      o[j] = kRelu ? fmaxf(z, 0.0f) : z;
    }
    RowIo<T, V>::store(y, row * (kD / V) + c * 64 + lane, o);
  }
  if (lane == 0) mean_rstd[row] = make_float2(mean, rstd);
}

// dx = rstd * (g w - mean(g w) - xhat * mean(g w xhat)),  g = dy (masked where the forward ReLU clipped: the
// pre-activation xhat * w + b is recomputed, nothing but x and the two statistics was kept)
template <typename T, int V, int kChunks, bool kRelu>
__global__ __launch_bounds__(64 * kLnWaves) void layernorm_backward_kernel(const void* __restrict__ x,
                                                                           const void* __restrict__ dy, long rows,
                                                                           const float* __restrict__ weight,
                                                                           const float* __restrict__ bias,
                                                                           const float2* __restrict__ mean_rstd,
                                                                           void* __restrict__ dx,
                                                                           float* __restrict__ partial) {
  constexpr int kD = 64 * V * kChunks;
  __shared__ __attribute__((aligned(16))) float fold[kLnWaves - 1][2][kD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float w[kChunks][V], b[kChunks][V], dw[kChunks][V], db[kChunks][V];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    RowIo<float, V>::load(weight, c * 64 + lane, w[c]);
    if (kRelu) RowIo<float, V>::load(bias, c * 64 + lane, b[c]);
#pragma unroll
    for (int j = 0; j < V; ++j) dw[c][j] = db[c][j] = 0.0f;
  }
  for (long row = static_cast<long>(blockIdx.x) * kLnWaves + wave; row < rows;
       row += static_cast<long>(gridDim.x) * kLnWaves) {
    const float2 st = mean_rstd[row];
    float xh[kChunks][V], g[kChunks][V];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      RowIo<T, V>::load(x, row * (kD / V) + c * 64 + lane, xh[c]);
      RowIo<T, V>::load(dy, row * (kD / V) + c * 64 + lane, g[c]);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        xh[c][j] = (xh[c][j] - st.x) * st.y;
        if (kRelu && !(fmaf(xh[c][j], w[c][j], b[c][j]) > 0.0f)) g[c][j] = 0.0f;
        dw[c][j] = fmaf(g[c][j], xh[c][j], dw[c][j]);
        db[c][j] += g[c][j];
        const float gw = g[c][j] * w[c][j];
        s1 += gw;
        s2 = fmaf(gw, xh[c][j], s2);
      }
    }
    const float c1 = wave_sum(s1) * (1.0f / kD), c2 = wave_sum(s2) * (1.0f / kD);
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = st.y * (g[c][j] * w[c][j] - c1 - xh[c][j] * c2);
      RowIo<T, V>::store(dx, row * (kD / V) + c * 64 + lane, o);
    }
  }
  // the block's column sums: wavefronts 1..3 hand theirs to wavefront 0 through LDS
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      RowIo<float, V>::store(fold[wave - 1][0], c * 64 + lane, dw[c]);
      RowIo<float, V>::store(fold[wave - 1][1], c * 64 + lane, db[c]);
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
#pragma unroll
      for (int k = 0; k < kLnWaves - 1; ++k) {
        float t[V], u[V];
        RowIo<float, V>::load(fold[k][0], c * 64 + lane, t);
        RowIo<float, V>::load(fold[k][1], c * 64 + lane, u);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          dw[c][j] += t[j];
          db[c][j] += u[j];
        }
      }
      float* out = partial + static_cast<long>(blockIdx.x) * 2 * kD;
      RowIo<float, V>::store(out, c * 64 + lane, dw[c]);
      RowIo<float, V>::store(out + kD, c * 64 + lane, db[c]);
    }
  }
}

// out[0][col] = sum over blocks of partial[blk][0][col] (d weight), out[1][col] likewise (d bias).  One block per 64
// columns: 16 row groups x 64 columns, every thread issues all of its (<= 32) loads before the first add -- one memory
// round trip instead of a dependent chain over the partial rows -- and the row groups are added through LDS in a fixed
// order (deterministic).
constexpr int kFinalGroups = 16;
constexpr int kFinalPerThread = kLnMaxBlocks / kFinalGroups;

__global__ __launch_bounds__(64 * kFinalGroups) void layernorm_backward_final_kernel(const float* __restrict__ partial,
                                                                                    int blocks, int d,
                                                                                    float* __restrict__ dweight,
                                                                                    float* __restrict__ dbias) {
  __shared__ float fold[kFinalGroups][64];
  const int lane = threadIdx.x & 63, group = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;                       // < 2 * d (d is a multiple of 64)
  float v[kFinalPerThread];
#pragma unroll
  for (int k = 0; k < kFinalPerThread; ++k) {
    const int row = group + kFinalGroups * k;
    v[k] = row < blocks ? partial[static_cast<long>(row) * 2 * d + col] : 0.0f;
  }
#pragma unroll
  for (int width = kFinalPerThread / 2; width >= 1; width >>= 1)
#pragma unroll
    for (int k = 0; k < width; ++k) v[k] += v[k + width];
  fold[group][lane] = v[0];
  __syncthreads();
  if (group == 0) {
    float total = 0.0f;
#pragma unroll
    for (int g = 0; g < kFinalGroups; ++g) total += fold[g][lane];
    if (col < d) dweight[col] = total;
    else dbias[col - d] = total;
  }
}

int backward_blocks(long rows) {
  const long want = (rows + kLnWaves - 1) / kLnWaves;
  return static_cast<int>(want < kLnMaxBlocks ? want : kLnMaxBlocks);
}

template <typename T, int V, int kChunks>
void launch_forward(const void* x, long rows, const float* weight, const float* bias, float eps, int relu, void* y,
                    float2* mean_rstd, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>((rows + kLnWaves - 1) / kLnWaves)), block(64 * kLnWaves);
  if (relu) hipLaunchKernelGGL((layernorm_forward_kernel<T, V, kChunks, true>), grid, block, 0, stream, x, rows, weight,
                               bias, eps, y, mean_rstd);
  else hipLaunchKernelGGL((layernorm_forward_kernel<T, V, kChunks, false>), grid, block, 0, stream, x, rows, weight,
                          bias, eps, y, mean_rstd);
}

template <typename T, int V, int kChunks>
void launch_backward(const void* x, const void* dy, long rows, const float* weight, const float* bias,
                     const float2* mean_rstd, int relu, void* dx, float* partial, hipStream_t stream) {
  const dim3 grid(static_cast<unsigned>(backward_blocks(rows))), block(64 * kLnWaves);
  if (relu) hipLaunchKernelGGL((layernorm_backward_kernel<T, V, kChunks, true>), grid, block, 0, stream, x, dy, rows,
                               weight, bias, mean_rstd, dx, partial);
  else hipLaunchKernelGGL((layernorm_backward_kernel<T, V, kChunks, false>), grid, block, 0, stream, x, dy, rows,
                          weight, bias, mean_rstd, dx, partial);
}

bool supported(int64_t d) { return d == 256 || d == 512 || d == 1024 || d == 2048; }

}  // namespace

}  // namespace seld

extern "C" {

int seld_layernorm_supported(int64_t D) { return seld::supported(D) ? 1 : 0; }

int64_t seld_layernorm_workspace_floats(int64_t rows, int64_t D) {
  if (rows <= 0 || !seld::supported(D)) return 0;
  return static_cast<int64_t>(seld::backward_blocks(rows)) * 2 * D;
}

int seld_layernorm_forward(const void* x, int is_bf16, int64_t rows, int64_t D, const float* weight, const float* bias,
                           float eps, int relu, void* y, float* mean_rstd, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (rows < 0) return fail(kErrInvalidArgument, "seld_layernorm_forward: negative row count");
  if (!supported(D)) return fail(kErrUnsupported, "seld_layernorm_forward: D must be 256, 512, 1024 or 2048");
  if (rows == 0) return kOk;
  if (!x || !weight || !bias || !y || !mean_rstd) return fail(kErrInvalidArgument, "seld_layernorm_forward: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float2* st = reinterpret_cast<float2*>(mean_rstd);
#define SELD_LN_FWD(T)                                                                               \
  switch (D) {                                                                                       \
    case 256: launch_forward<T, 4, 1>(x, rows, weight, bias, eps, relu, y, st, stream); break;       \
    case 512: launch_forward<T, 8, 1>(x, rows, weight, bias, eps, relu, y, st, stream); break;       \
    case 1024: launch_forward<T, 8, 2>(x, rows, weight, bias, eps, relu, y, st, stream); break;      \
    default: launch_forward<T, 8, 4>(x, rows, weight, bias, eps, relu, y, st, stream); break;        \
  }
  if (is_bf16) { SELD_LN_FWD(__hip_bfloat16) } else { SELD_LN_FWD(float) }
#undef SELD_LN_FWD
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_layernorm_backward(const void* x, const void* dy, int is_bf16, int64_t rows, int64_t D, const float* weight,
                            const float* bias, const float* mean_rstd, int relu, void* dx, float* dweight,
                            float* dbias, float* workspace, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (rows <= 0) return fail(kErrInvalidArgument, "seld_layernorm_backward: rows must be positive");
  if (!supported(D)) return fail(kErrUnsupported, "seld_layernorm_backward: D must be 256, 512, 1024 or 2048");
  if (!x || !dy || !weight || !bias || !mean_rstd || !dx || !dweight || !dbias || !workspace)
    return fail(kErrInvalidArgument, "seld_layernorm_backward: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const float2* st = reinterpret_cast<const float2*>(mean_rstd);
#define SELD_LN_BWD(T)                                                                                         \
  switch (D) {                                                                                                 \
    case 256: launch_backward<T, 4, 1>(x, dy, rows, weight, bias, st, relu, dx, workspace, stream); break;     \
    case 512: launch_backward<T, 8, 1>(x, dy, rows, weight, bias, st, relu, dx, workspace, stream); break;     \
    case 1024: launch_backward<T, 8, 2>(x, dy, rows, weight, bias, st, relu, dx, workspace, stream); break;    \
    default: launch_backward<T, 8, 4>(x, dy, rows, weight, bias, st, relu, dx, workspace, stream); break;      \
  }
  if (is_bf16) { SELD_LN_BWD(__hip_bfloat16) } else { SELD_LN_BWD(float) }
#undef SELD_LN_BWD
  const int d = static_cast<int>(D);
  hipLaunchKernelGGL(layernorm_backward_final_kernel, dim3(2 * d / 64), dim3(64 * kFinalGroups), 0, stream, workspace,
                     backward_blocks(rows), d, dweight, dbias);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
