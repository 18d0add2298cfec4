// Fused tail of a CNN encoder block for gfx950:  BatchNorm2d -> ReLU -> MaxPool2d((1, 2))  (forward and backward).
//
// Replaces, for channels-last activations, the three stock modules after every 3x3 convolution of the shared
// CRNN / Conformer encoder (model_crnn.py:5-17 ConvBlock.forward: ``self.pool(self.relu(self.bn(self.conv(x))))``).
// Unfused on ROCm that is 5 launches forward (MIOpen mean/var, final mean/var, normalise; clamp; max_pool) and 5
// backward (max_pool_backward with int64 indices, threshold_backward, 3 MIOpen BN kernels) which read / write the
// 65.5 MB pre-pool activation of a block about 9 times per direction.  This is pure HBM-bound elementwise and
// reduction work; here
//
//   forward :  stats (read x once)  ->  finalise (tiny)  ->  apply: y = max_pair(relu(a x + b))   (read x, write y/2)
//   backward:  reduce (read x, dy)  ->  finalise (tiny)  ->  apply: dx = a dz + p + q x            (read x, dy; write dx)
//
// and nothing but x itself is kept for the backward pass: the ReLU mask and the pooling argmax are recomputed from
// x with the SAME roundings the forward used (z is rounded to the activation dtype before the comparisons, exactly
// what the unfused bf16 modules compare), ties go to the first element like max_pool2d.
//
// Layout: x is the conv output in channels-last memory order, i.e. a row-major [rows = B*T*F][C] matrix; the two
// frequency bins of a pooling pair are ADJACENT ROWS (2o, 2o+1), so an output row's operands are one contiguous
// 2*C run.  A thread owns 8 consecutive channels (one 16-byte bf16 access); C/8 threads cover a row and C/8
// divides the 256-thread block, so a thread's channel group -- and its per-channel coefficients, kept in
// registers -- never changes along the grid-stride loop.
//
// Algorithmic bytes per pre-pool element (bf16): forward 2 (stats) + 2 + 1 (apply) = 5 B; backward 2 + 1 (reduce)
// + 2 + 1 + 2 (apply) = 8 B.  Statistics: fp32 accumulation of shifted data (shift = the channel's first element,
// which removes the E[x^2] - E[x]^2 cancellation), fixed-order two-level reduction, final combination in double:
// deterministic, no float atomics.
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kTailThreads = 256;
constexpr int kTailStatBlocks = 1024;    // partial-sum rows per statistic (4 blocks per CU)
constexpr int kFinalChannels = 16;       // channels per finalise block
constexpr int kFinalThreads = 1024;      // 32 (statistic, channel) items x 32 partial rows in flight

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16_pair(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));     // v_cvt_pk_bf16_f32 (RNE)
}

// ---- 8 consecutive channels of one row
template <typename T> struct Row8;
template <> struct Row8<__hip_bfloat16> {
  static __device__ __forceinline__ void load(const void* base, long elem, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(base) + elem);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(void* base, long elem, const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16_pair(f[0], f[1]);
    v.y = pack_bf16_pair(f[2], f[3]);
    v.z = pack_bf16_pair(f[4], f[5]);
    v.w = pack_bf16_pair(f[6], f[7]);
    *reinterpret_cast<uint4*>(static_cast<unsigned short*>(base) + elem) = v;
  }
  // the value the unfused modules would hold after BatchNorm wrote its bf16 output
  static __device__ __forceinline__ float round(float z) {
    return __uint_as_float(pack_bf16_pair(z, 0.0f) << 16);
  }
};
template <> struct Row8<float> {
  static __device__ __forceinline__ void load(const void* base, long elem, float (&f)[8]) {
    const float4* p = reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
    const float4 a = p[0], b = p[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  static __device__ __forceinline__ void store(void* base, long elem, const float (&f)[8]) {
    float4* p = reinterpret_cast<float4*>(static_cast<float*>(base) + elem);
    p[0] = make_float4(f[0], f[1], f[2], f[3]);
    p[1] = make_float4(f[4], f[5], f[6], f[7]);
  }
  static __device__ __forceinline__ float round(float z) { return z; }
};

// Sum `acc[kN]` (per thread: 8 channels x kN/8 statistics) over the row slots of the block and write the block's
// partial sums to partials[stat][block][C].  Threads are laid out tid = slot * groups + cg.
template <int kStats>
__device__ __forceinline__ void block_partials(const float (&acc)[kStats][8], int C, float* __restrict__ partials,
                                               float* __restrict__ lds /* [kTailThreads][kStats*8 + 1] */) {
  const int tid = threadIdx.x;
  const int groups = C >> 3;
  const int slots = kTailThreads / groups;
  constexpr int kPitch = kStats * 8 + 1;
#pragma unroll
  for (int s = 0; s < kStats; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[tid * kPitch + s * 8 + i] = acc[s][i];
  __syncthreads();
  // one (statistic, channel) per thread pass
  for (int item = tid; item < kStats * C; item += kTailThreads) {
    const int s = item / C, ch = item - s * C;
    const int cg = ch >> 3, i = ch & 7;
    float sum = 0.0f;
    for (int slot = 0; slot < slots; ++slot) sum += lds[(slot * groups + cg) * kPitch + s * 8 + i];
    partials[(static_cast<long>(s) * gridDim.x + blockIdx.x) * C + ch] = sum;
  }
}

// ------------------------------------------------------------------------------------------- forward

template <typename T>
__global__ __launch_bounds__(kTailThreads) void tail_stats_kernel(const void* __restrict__ x, long rows, int C,
                                                                  float* __restrict__ partials) {
  __shared__ float lds[kTailThreads * 17];
  const int groups = C >> 3;
  const int cg = threadIdx.x % groups;
  const long slot = (static_cast<long>(blockIdx.x) * kTailThreads + threadIdx.x) / groups;
  const long slots_total = static_cast<long>(gridDim.x) * kTailThreads / groups;
  float shift[8];
  Row8<T>::load(x, 8 * cg, shift);                    // row 0 of this channel group (same for every block)
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.0f;
  long r = slot;
  for (; r + 3 * slots_total < rows; r += 4 * slots_total) {       // 4 independent 16-byte loads in flight
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) Row8<T>::load(x, (r + u * slots_total) * C + 8 * cg, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[u][i] - shift[i];
        acc[0][i] += d;
        acc[1][i] = fmaf(d, d, acc[1][i]);
      }
  }
  for (; r < rows; r += slots_total) {
    float v[8];
    Row8<T>::load(x, r * C + 8 * cg, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float d = v[i] - shift[i];
      acc[0][i] += d;
      acc[1][i] = fmaf(d, d, acc[1][i]);
    }
  }
  block_partials<2>(acc, C, partials, lds);
}

// One block per 16 channels: reduce the partial rows in double, produce mean / invstd / affine coefficients and
// update the running statistics exactly like nn.BatchNorm2d in training mode (biased variance for the
// normalisation, unbiased for running_var, momentum blend).  training == 0: coefficients from the running stats.
template <typename T>
__global__ __launch_bounds__(kFinalThreads) void tail_stats_final_kernel(const void* __restrict__ x, long rows, int C,
                                                                        const float* __restrict__ partials, int nblocks,
                                                                        const float* __restrict__ weight,
                                                                        const float* __restrict__ bias,
                                                                        float* __restrict__ running_mean,
                                                                        float* __restrict__ running_var,
                                                                        float momentum, float eps, int training,
                                                                        float* __restrict__ mean_invstd /* [2][C] */,
                                                                        float* __restrict__ scale_shift /* [2][C] */) {
  __shared__ double red[kFinalThreads];
  const int tid = threadIdx.x;
  const int item = tid & 31;                       // (statistic, channel-in-block)
  const int s = item >> 4, ch = blockIdx.x * kFinalChannels + (item & 15);
  const int lane_row = tid >> 5;                   // 32 partial rows in parallel, 8 loads in flight each
  double sum = 0.0;
  if (training && ch < C) {
#pragma unroll 8
    for (int b = lane_row; b < nblocks; b += kFinalThreads / 32)
      sum += static_cast<double>(partials[(static_cast<long>(s) * nblocks + b) * C + ch]);
  }
  red[tid] = sum;
  __syncthreads();
  if (tid < 32) {
    for (int k = 1; k < kFinalThreads / 32; ++k) sum += red[tid + 32 * k];
    red[tid] = sum;
  }
  __syncthreads();
  if (tid < kFinalChannels && ch < C) {
    double mean, var;
    if (training) {
      const double n = static_cast<double>(rows);
      float first[8];
      Row8<T>::load(x, ch & ~7, first);
      const double shift = static_cast<double>(first[ch & 7]);
      const double m1 = red[tid] / n, m2 = red[tid + 16] / n;
      mean = shift + m1;
      var = m2 - m1 * m1;
      if (var < 0.0) var = 0.0;
      if (running_mean) {
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_mean[ch] = static_cast<float>((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = static_cast<float>((1.0 - momentum) * running_var[ch] + momentum * unbiased);
      }
    } else {
      mean = running_mean[ch];
      var = running_var[ch];
    }
    const float meanf = static_cast<float>(mean);
    const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
    const float a = (weight ? weight[ch] : 1.0f) * invstd;
    mean_invstd[ch] = meanf;
    mean_invstd[C + ch] = invstd;
    scale_shift[ch] = a;
    scale_shift[C + ch] = (bias ? bias[ch] : 0.0f) - meanf * a;
  }
}

// SiLU (Swish) and its derivative at z, for mode 4
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  return s * fmaf(z, 1.0f - s, 1.0f);
}

// kPool: 2 = MaxPool2d((1,2)) over adjacent rows; 1 = no pooling; 3 = no pooling + residual add before the ReLU;
// 4 = no pooling, SiLU instead of ReLU (BatchNorm1d -> Swish of the Conformer convolution module, model_conformer.py:71-96)
// (the tail of a ResNet bottleneck, resnet50_model.py:30-52: relu(bn3(conv3(y)) + shortcut)); in mode 3 the second
// operand row `x1` carries the residual, rounded into the activation dtype after the add like the unfused modules.
template <typename T, int kPool>
__global__ __launch_bounds__(kTailThreads) void tail_apply_kernel(const void* __restrict__ x,
                                                                  const void* __restrict__ res, long out_rows, int C,
                                                                  const float* __restrict__ scale_shift,
                                                                  void* __restrict__ y) {
  constexpr int kStride = kPool == 2 ? 2 : 1;
  const int groups = C >> 3;
  const int cg = threadIdx.x % groups;
  const long slot = (static_cast<long>(blockIdx.x) * kTailThreads + threadIdx.x) / groups;
  const long slots_total = static_cast<long>(gridDim.x) * kTailThreads / groups;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = scale_shift[8 * cg + i];
    b[i] = scale_shift[C + 8 * cg + i];
  }
  auto one = [&](const float (&x0)[8], const float (&x1)[8], long o) {
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float z = Row8<T>::round(fmaf(x0[i], a[i], b[i]));
      if (kPool == 2) z = fmaxf(z, Row8<T>::round(fmaf(x1[i], a[i], b[i])));
      if (kPool == 3) z = Row8<T>::round(z + x1[i]);
      out[i] = kPool == 4 ? silu_f(z) : fmaxf(z, 0.0f);     // max(relu(z0), relu(z1)) == relu(max(z0, z1))
    }
    Row8<T>::store(y, o * C + 8 * cg, out);
  };
  long o = slot;
  for (; o + slots_total < out_rows; o += 2 * slots_total) {       // two output rows (4 loads) in flight
    float x0[2][8], x1[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long oo = o + u * slots_total;
      Row8<T>::load(x, (kStride * oo) * C + 8 * cg, x0[u]);
      if (kPool == 2) Row8<T>::load(x, (kStride * oo + 1) * C + 8 * cg, x1[u]);
      if (kPool == 3) Row8<T>::load(res, oo * C + 8 * cg, x1[u]);
    }
    one(x0[0], x1[0], o);
    one(x0[1], x1[1], o + slots_total);
  }
  for (; o < out_rows; o += slots_total) {
    float x0[8], x1[8];
    Row8<T>::load(x, (kStride * o) * C + 8 * cg, x0);
    if (kPool == 2) Row8<T>::load(x, (kStride * o + 1) * C + 8 * cg, x1);
    if (kPool == 3) Row8<T>::load(res, o * C + 8 * cg, x1);
    one(x0, x1, o);
  }
}

// ------------------------------------------------------------------------------------------- backward

// Gradient routing of one output element: which of the pair receives dy (first wins ties, like max_pool2d),
// and whether the ReLU lets it through.  Returns the receiving element's index (0 / 1) or -1.
template <typename T, int kPool>
__device__ __forceinline__ int route(float x0, float x1, float a, float b) {
  if (kPool == 3) return fmaxf(Row8<T>::round(Row8<T>::round(fmaf(x0, a, b)) + x1), 0.0f) > 0.0f ? 0 : -1;
  const float r0 = fmaxf(Row8<T>::round(fmaf(x0, a, b)), 0.0f);
  if (kPool == 1) return r0 > 0.0f ? 0 : -1;
  const float r1 = fmaxf(Row8<T>::round(fmaf(x1, a, b)), 0.0f);
  const int sel = r1 > r0 ? 1 : 0;
  return (sel ? r1 : r0) > 0.0f ? sel : -1;
}

template <typename T, int kPool>
__global__ __launch_bounds__(kTailThreads) void tail_bwd_reduce_kernel(const void* __restrict__ x,
                                                                       const void* __restrict__ res,
                                                                       const void* __restrict__ dy, long out_rows,
                                                                       int C, const float* __restrict__ scale_shift,
                                                                       const float* __restrict__ mean_invstd,
                                                                       float* __restrict__ partials) {
  constexpr int kStride = kPool == 2 ? 2 : 1;
  __shared__ float lds[kTailThreads * 17];
  const int groups = C >> 3;
  const int cg = threadIdx.x % groups;
  const long slot = (static_cast<long>(blockIdx.x) * kTailThreads + threadIdx.x) / groups;
  const long slots_total = static_cast<long>(gridDim.x) * kTailThreads / groups;
  float a[8], b[8], mean[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = scale_shift[8 * cg + i];
    b[i] = scale_shift[C + 8 * cg + i];
    mean[i] = mean_invstd[8 * cg + i];
  }
  float acc[2][8];                                  // sum dz, sum dz * (x - mean)   (invstd applied at the end)
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = acc[1][i] = 0.0f;
  auto one = [&](const float (&x0)[8], const float (&x1)[8], const float (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float dz, xs;
      if (kPool == 4) {
        dz = g[i] * silu_grad_f(Row8<T>::round(fmaf(x0[i], a[i], b[i])));
        xs = x0[i];
      } else {
        const int sel = route<T, kPool>(x0[i], x1[i], a[i], b[i]);
        dz = sel >= 0 ? g[i] : 0.0f;
        xs = sel == 1 ? x1[i] : x0[i];
      }
      acc[0][i] += dz;
      acc[1][i] = fmaf(dz, xs - mean[i], acc[1][i]);
    }
  };
  long o = slot;
  for (; o + slots_total < out_rows; o += 2 * slots_total) {
    float x0[2][8], x1[2][8], g[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long oo = o + u * slots_total;
      Row8<T>::load(x, (kStride * oo) * C + 8 * cg, x0[u]);
      if (kPool == 2) Row8<T>::load(x, (kStride * oo + 1) * C + 8 * cg, x1[u]);
      if (kPool == 3) Row8<T>::load(res, oo * C + 8 * cg, x1[u]);
      Row8<T>::load(dy, oo * C + 8 * cg, g[u]);
    }
    one(x0[0], x1[0], g[0]);
    one(x0[1], x1[1], g[1]);
  }
  for (; o < out_rows; o += slots_total) {
    float x0[8], x1[8], g[8];
    Row8<T>::load(x, (kStride * o) * C + 8 * cg, x0);
    if (kPool == 2) Row8<T>::load(x, (kStride * o + 1) * C + 8 * cg, x1);
    if (kPool == 3) Row8<T>::load(res, o * C + 8 * cg, x1);
    Row8<T>::load(dy, o * C + 8 * cg, g);
    one(x0, x1, g);
  }
  block_partials<2>(acc, C, partials, lds);
}

// dweight = sum dz xhat, dbias = sum dz; coefficients of dx = a dz + p + q x  with
//   q = -a * mean(dz xhat) * invstd,  p = -a * mean(dz) - q * mean_x.
__global__ __launch_bounds__(kFinalThreads) void tail_bwd_final_kernel(long rows, int C,
                                                                      const float* __restrict__ partials, int nblocks,
                                                                      const float* __restrict__ scale_shift,
                                                                      const float* __restrict__ mean_invstd,
                                                                      float* __restrict__ dweight,
                                                                      float* __restrict__ dbias,
                                                                      float* __restrict__ coef /* [2][C] p, q */) {
  __shared__ double red[kFinalThreads];
  const int tid = threadIdx.x;
  const int item = tid & 31;
  const int s = item >> 4, ch = blockIdx.x * kFinalChannels + (item & 15);
  const int lane_row = tid >> 5;
  double sum = 0.0;
  if (ch < C) {
#pragma unroll 8
    for (int b = lane_row; b < nblocks; b += kFinalThreads / 32)
      sum += static_cast<double>(partials[(static_cast<long>(s) * nblocks + b) * C + ch]);
  }
  red[tid] = sum;
  __syncthreads();
  if (tid < 32) {
    for (int k = 1; k < kFinalThreads / 32; ++k) sum += red[tid + 32 * k];
    red[tid] = sum;
  }
  __syncthreads();
  if (tid < kFinalChannels && ch < C) {
    const double invstd = mean_invstd[C + ch], mean = mean_invstd[ch], a = scale_shift[ch];
    const double s_dz = red[tid], s_dzx = red[tid + 16] * invstd;      // sum dz xhat
    const double n = static_cast<double>(rows);
    dbias[ch] = static_cast<float>(s_dz);
    dweight[ch] = static_cast<float>(s_dzx);
    const double q = -a * (s_dzx / n) * invstd;
    coef[ch] = static_cast<float>(-a * (s_dz / n) - q * mean);
    coef[C + ch] = static_cast<float>(q);
  }
}

template <typename T, int kPool>
__global__ __launch_bounds__(kTailThreads) void tail_bwd_apply_kernel(const void* __restrict__ x,
                                                                      const void* __restrict__ res,
                                                                      const void* __restrict__ dy, long out_rows,
                                                                      int C, const float* __restrict__ scale_shift,
                                                                      const float* __restrict__ coef,
                                                                      void* __restrict__ dx, void* __restrict__ dres) {
  constexpr int kStride = kPool == 2 ? 2 : 1;
  const int groups = C >> 3;
  const int cg = threadIdx.x % groups;
  const long slot = (static_cast<long>(blockIdx.x) * kTailThreads + threadIdx.x) / groups;
  const long slots_total = static_cast<long>(gridDim.x) * kTailThreads / groups;
  float a[8], b[8], p[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = scale_shift[8 * cg + i];
    b[i] = scale_shift[C + 8 * cg + i];
    p[i] = coef[8 * cg + i];
    q[i] = coef[C + 8 * cg + i];
  }
  auto one = [&](const float (&x0)[8], const float (&x1)[8], const float (&g)[8], long o) {
    float d0[8], d1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float base0 = fmaf(q[i], x0[i], p[i]);
      if (kPool == 4) {
        d0[i] = fmaf(a[i], g[i] * silu_grad_f(Row8<T>::round(fmaf(x0[i], a[i], b[i]))), base0);
        continue;
      }
      const int sel = route<T, kPool>(x0[i], x1[i], a[i], b[i]);
      d0[i] = sel == 0 ? fmaf(a[i], g[i], base0) : base0;
      if (kPool == 2) {
        const float base1 = fmaf(q[i], x1[i], p[i]);
        d1[i] = sel == 1 ? fmaf(a[i], g[i], base1) : base1;
      }
      if (kPool == 3) d1[i] = sel == 0 ? g[i] : 0.0f;
    }
    Row8<T>::store(dx, (kStride * o) * C + 8 * cg, d0);
    if (kPool == 2) Row8<T>::store(dx, (kStride * o + 1) * C + 8 * cg, d1);
    if (kPool == 3) Row8<T>::store(dres, o * C + 8 * cg, d1);            // gradient of the residual = masked dy
  };
  long o = slot;
  for (; o + slots_total < out_rows; o += 2 * slots_total) {
    float x0[2][8], x1[2][8], g[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long oo = o + u * slots_total;
      Row8<T>::load(x, (kStride * oo) * C + 8 * cg, x0[u]);
      if (kPool == 2) Row8<T>::load(x, (kStride * oo + 1) * C + 8 * cg, x1[u]);
      if (kPool == 3) Row8<T>::load(res, oo * C + 8 * cg, x1[u]);
      Row8<T>::load(dy, oo * C + 8 * cg, g[u]);
    }
    one(x0[0], x1[0], g[0], o);
    one(x0[1], x1[1], g[1], o + slots_total);
  }
  for (; o < out_rows; o += slots_total) {
    float x0[8], x1[8], g[8];
    Row8<T>::load(x, (kStride * o) * C + 8 * cg, x0);
    if (kPool == 2) Row8<T>::load(x, (kStride * o + 1) * C + 8 * cg, x1);
    if (kPool == 3) Row8<T>::load(res, o * C + 8 * cg, x1);
    Row8<T>::load(dy, o * C + 8 * cg, g);
    one(x0, x1, g, o);
  }
}

static int tail_check(const char* who, int64_t rows, int C, int pool) {
  if (rows <= 0 || C <= 0) return fail(kErrInvalidArgument, std::string(who) + ": rows and C must be positive");
  if (pool != 1 && pool != 2 && pool != 4)
    return fail(kErrUnsupported, std::string(who) + ": pool must be 1 (none), 2 (MaxPool2d((1,2))) or 4 (none, SiLU)");
  if (C % 8 != 0 || kTailThreads % (C / 8) != 0)
    return fail(kErrUnsupported, std::string(who) + ": C must be 8 * (a divisor of 256)");
  if (pool == 2 && rows % 2 != 0) return fail(kErrUnsupported, std::string(who) + ": rows must be a multiple of the pool width");
  return kOk;
}

// Partial-sum rows of a statistics / reduction pass: every thread should have >= 8 rows to add, at most kTailStatBlocks
// workgroups.  (A fixed 1024 made the finalise kernels of the ResNet's small late-stage activations -- 53 BatchNorm
// layers per iteration -- add 1024 partial rows of a tensor with 2048 rows.)
static int tail_stat_blocks(long work_rows, int C) {
  const long per_block = kTailThreads / (C / 8);
  long blocks = (work_rows + per_block * 8 - 1) / (per_block * 8);
  if (blocks < 32) blocks = 32;
  if (blocks > kTailStatBlocks) blocks = kTailStatBlocks;
  return static_cast<int>(blocks);
}

static unsigned tail_grid(const DeviceState* st, long work_rows, int C) {
  const long per_block = kTailThreads / (C / 8);
  long blocks = (work_rows + per_block - 1) / per_block;
  const long cap = static_cast<long>(st->num_cus > 0 ? st->num_cus : 256) * 8;
  if (blocks > cap) blocks = cap;
  return static_cast<unsigned>(blocks < 1 ? 1 : blocks);
}

}  // namespace seld

extern "C" {

int64_t seld_conv_tail_workspace_floats(int C) { return 2LL * seld::kTailStatBlocks * C + 2LL * C; }

int seld_conv_tail_forward(const void* x, const void* residual, int is_bf16, int64_t rows, int C, int pool,
                           const float* weight,
                           const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                           int training, void* y, float* mean_invstd, float* scale_shift, float* workspace,
                           void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (int rc = tail_check("seld_conv_tail_forward", rows, C, pool)) return rc;
  if (residual && pool != 1) return fail(kErrUnsupported, "seld_conv_tail_forward: a residual needs pool = 1");   // (mode 4: none)
  if (!x || !y || !mean_invstd || !scale_shift || (training && !workspace))
    return fail(kErrInvalidArgument, "seld_conv_tail_forward: null pointer");
  if (!training && (!running_mean || !running_var))
    return fail(kErrInvalidArgument, "seld_conv_tail_forward: eval mode needs the running statistics");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int nblocks = tail_stat_blocks(rows, C);
  if (training) {
    if (is_bf16) hipLaunchKernelGGL(tail_stats_kernel<__hip_bfloat16>, dim3(nblocks), dim3(kTailThreads), 0, stream, x,
                                    static_cast<long>(rows), C, workspace);
    else hipLaunchKernelGGL(tail_stats_kernel<float>, dim3(nblocks), dim3(kTailThreads), 0, stream, x,
                            static_cast<long>(rows), C, workspace);
  }
  const dim3 fgrid((C + kFinalChannels - 1) / kFinalChannels);
  if (is_bf16) hipLaunchKernelGGL(tail_stats_final_kernel<__hip_bfloat16>, fgrid, dim3(kFinalThreads), 0, stream, x,
                                  static_cast<long>(rows), C, workspace, nblocks, weight, bias, running_mean,
                                  running_var, momentum, eps, training, mean_invstd, scale_shift);
  else hipLaunchKernelGGL(tail_stats_final_kernel<float>, fgrid, dim3(kFinalThreads), 0, stream, x,
                          static_cast<long>(rows), C, workspace, nblocks, weight, bias, running_mean, running_var,
                          momentum, eps, training, mean_invstd, scale_shift);
  const long out_rows = pool == 2 ? rows / 2 : rows;
  const dim3 grid(tail_grid(st, out_rows, C));
  const int mode = residual ? 3 : pool;
#define SELD_TAIL_APPLY(T, P)                                                                                     \
  hipLaunchKernelGGL((tail_apply_kernel<T, P>), grid, dim3(kTailThreads), 0, stream, x, residual, out_rows, C, \
                     scale_shift, y)
  if (is_bf16) {
    if (mode == 2) SELD_TAIL_APPLY(__hip_bfloat16, 2);
    else if (mode == 3) SELD_TAIL_APPLY(__hip_bfloat16, 3);
    else if (mode == 4) SELD_TAIL_APPLY(__hip_bfloat16, 4);
    else SELD_TAIL_APPLY(__hip_bfloat16, 1);
  } else {
    if (mode == 2) SELD_TAIL_APPLY(float, 2);
    else if (mode == 3) SELD_TAIL_APPLY(float, 3);
    else if (mode == 4) SELD_TAIL_APPLY(float, 4);
    else SELD_TAIL_APPLY(float, 1);
  }
#undef SELD_TAIL_APPLY
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_conv_tail_backward(const void* x, const void* residual, const void* dy, int is_bf16, int64_t rows, int C,
                            int pool, const float* mean_invstd, const float* scale_shift, void* dx, void* dresidual,
                            float* dweight, float* dbias, float* workspace, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (int rc = tail_check("seld_conv_tail_backward", rows, C, pool)) return rc;
  if ((residual != nullptr) != (dresidual != nullptr) || (residual && pool != 1))
    return fail(kErrInvalidArgument, "seld_conv_tail_backward: residual and dresidual go together, with pool = 1");
  if (!x || !dy || !dx || !mean_invstd || !scale_shift || !dweight || !dbias || !workspace)
    return fail(kErrInvalidArgument, "seld_conv_tail_backward: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long out_rows = pool == 2 ? rows / 2 : rows;
  const int nblocks = tail_stat_blocks(out_rows, C);
  float* coef = workspace + 2L * kTailStatBlocks * C;        // [2][C] after the (at most kTailStatBlocks) partial sums
#define SELD_TAIL_BWD(T, P)                                                                                      \
  do {                                                                                                           \
    hipLaunchKernelGGL((tail_bwd_reduce_kernel<T, P>), dim3(nblocks), dim3(kTailThreads), 0, stream, x, residual, \
                       dy, out_rows, C, scale_shift, mean_invstd, workspace);                                    \
    hipLaunchKernelGGL(tail_bwd_final_kernel, dim3((C + kFinalChannels - 1) / kFinalChannels), dim3(kFinalThreads), \
                       0, stream, static_cast<long>(rows), C, workspace, nblocks, scale_shift, mean_invstd,      \
                       dweight, dbias, coef);                                                                    \
    hipLaunchKernelGGL((tail_bwd_apply_kernel<T, P>), dim3(tail_grid(st, out_rows, C)), dim3(kTailThreads), 0,   \
                       stream, x, residual, dy, out_rows, C, scale_shift, coef, dx, dresidual);                  \
  } while (0)
  const int mode = residual ? 3 : pool;
  if (is_bf16) {
    if (mode == 2) SELD_TAIL_BWD(__hip_bfloat16, 2);
    else if (mode == 3) SELD_TAIL_BWD(__hip_bfloat16, 3);
    else if (mode == 4) SELD_TAIL_BWD(__hip_bfloat16, 4);
    else SELD_TAIL_BWD(__hip_bfloat16, 1);
  } else {
    if (mode == 2) SELD_TAIL_BWD(float, 2);
    else if (mode == 3) SELD_TAIL_BWD(float, 3);
    else if (mode == 4) SELD_TAIL_BWD(float, 4);
    else SELD_TAIL_BWD(float, 1);
  }
#undef SELD_TAIL_BWD
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
