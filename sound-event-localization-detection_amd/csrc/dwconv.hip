// Depthwise 1-D convolution over time in channels-last layout [B][T][D] for gfx950 -- the k = 31 depthwise stage of the
// Conformer convolution module (model_conformer.py:71-96: nn.Conv1d(d, d, 31, padding=15, groups=d) on [B, D, T]).
//
// MIOpen has no bf16 solver for this shape on gfx950 and falls back to naive_conv_* kernels (38 us forward, 81 us
// backward-data and 38 us backward-weight per module on 4 MB of activations) and the [B, D, T] layout costs two
// transposes per pointwise convolution around it.  Here the whole module stays in the [B, T, D] layout of the
// surrounding LayerNorm / Linear layers: a thread owns 8 consecutive channels (16-byte bf16 accesses) of one time
// step and walks the K taps (rows of the same 64-channel column block: L1 / L2 hits), the K x 64 weights of the block
// sit in LDS.  HBM-bound in principle (read x, write y: 4 B per element); at 4 MB per tensor the launches are
// latency-sized (a few microseconds).  Backward data = the same kernel with the taps flipped; backward weight = one
// block per (batch row, 64 channels, 50 time steps) that slides an 8-tap register window over its LDS-staged chunk
// (fp32 partials per block, summed by the caller: deterministic).
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kDwMaxTaps = 32;

template <typename T> struct Dw8;
template <> struct Dw8<__hip_bfloat16> {
  static __device__ __forceinline__ void load(const void* p, long e, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(p) + e);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(void* p, long e, const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(f[2 * i]))) |
             (static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(f[2 * i + 1]))) << 16);
    *reinterpret_cast<uint4*>(static_cast<unsigned short*>(p) + e) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ float load1(const void* p, long e) {
    return __uint_as_float(static_cast<unsigned>(static_cast<const unsigned short*>(p)[e]) << 16);
  }
};
template <> struct Dw8<float> {
  static __device__ __forceinline__ void load(const void* p, long e, float (&f)[8]) {
    const float4* q = reinterpret_cast<const float4*>(static_cast<const float*>(p) + e);
    const float4 a = q[0], b = q[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  static __device__ __forceinline__ void store(void* p, long e, const float (&f)[8]) {
    float4* q = reinterpret_cast<float4*>(static_cast<float*>(p) + e);
    q[0] = make_float4(f[0], f[1], f[2], f[3]);
    q[1] = make_float4(f[4], f[5], f[6], f[7]);
  }
  static __device__ __forceinline__ float load1(const void* p, long e) { return static_cast<const float*>(p)[e]; }
};

// y[b][t][d] = bias[d] + sum_k w[d][k] x[b][t + k - pad][d]   (flip: w[d][K-1-k], the data gradient)
// grid (ceil(T / 32), D / 64, B), 256 threads = 32 time steps x 8 channel groups of 8.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int Tn, int D, int K, int flip,
                                                     void* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float wl[kDwMaxTaps][64];
  const int tid = threadIdx.x;
  const int d0 = blockIdx.y * 64;
  for (int i = tid; i < K * 64; i += 256) {
    const int k = i >> 6, c = i & 63;
    wl[k][c] = w[static_cast<long>(d0 + c) * K + (flip ? K - 1 - k : k)];
  }
  __syncthreads();
  const int cg = tid & 7, t = blockIdx.x * 32 + (tid >> 3);
  if (t >= Tn) return;
  const int pad = (K - 1) / 2;
  const long row0 = static_cast<long>(blockIdx.z) * Tn;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = bias ? bias[d0 + 8 * cg + c] : 0.0f;
  for (int k = 0; k < K; ++k) {
    const int ti = t + k - pad;
    if (ti < 0 || ti >= Tn) continue;
    float xv[8];
    Dw8<T>::load(x, (row0 + ti) * D + d0 + 8 * cg, xv);
    const float4 w0 = *reinterpret_cast<const float4*>(&wl[k][8 * cg]);
    const float4 w1 = *reinterpret_cast<const float4*>(&wl[k][8 * cg + 4]);
    acc[0] = fmaf(w0.x, xv[0], acc[0]); acc[1] = fmaf(w0.y, xv[1], acc[1]);
    acc[2] = fmaf(w0.z, xv[2], acc[2]); acc[3] = fmaf(w0.w, xv[3], acc[3]);
    acc[4] = fmaf(w1.x, xv[4], acc[4]); acc[5] = fmaf(w1.y, xv[5], acc[5]);
    acc[6] = fmaf(w1.z, xv[6], acc[6]); acc[7] = fmaf(w1.w, xv[7], acc[7]);
  }
  Dw8<T>::store(y, (row0 + t) * D + d0 + 8 * cg, acc);
}

// dw_partial[row][d][k] = sum_{t in chunk} dy[b][t][d] x[b][t + k - pad][d]  (k < K);  slot 31 = sum_t dy (bias);
// row = b * chunks + chunk.  grid (D / 64, B, chunks), 256 threads = 64 channels x 4 tap groups of 8.
// Round 1 ran one block per (batch row, 64 channels) that walked all T steps with two dependent 2-byte loads per step:
// 128 blocks, 250 serial HBM/L2 round trips, 119 us for 4 MB.  Now a block takes kDwChunk time steps: the chunk of dy and
// the chunk + halo of x are staged in LDS once (16-byte loads, 8 channels per lane), then every thread slides its 8-tap
// register window over the chunk from LDS (conflict free: consecutive lanes = consecutive channels).
constexpr int kDwChunk = 50;

template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                           int Tn, int D, int K, float* __restrict__ partial) {
  __shared__ float xs[kDwChunk + kDwMaxTaps][64];        // rows t0 - pad .. t0 + chunk + pad (zero outside the sequence)
  __shared__ float gs[kDwChunk][64];
  const int tid = threadIdx.x;
  const int d0 = blockIdx.x * 64;
  const int pad = (K - 1) / 2;
  const int t0 = blockIdx.z * kDwChunk;
  const int len = Tn - t0 < kDwChunk ? Tn - t0 : kDwChunk;
  const long row0 = static_cast<long>(blockIdx.y) * Tn;
  // stage: 8 lanes cover the 64 channels of a row
  for (int i = tid; i < (kDwChunk + kDwMaxTaps) * 8; i += 256) {
    const int r = i >> 3, cg = i & 7;
    const int ti = t0 - pad + r;
    float v[8];
    if (r < len + 2 * pad && ti >= 0 && ti < Tn) Dw8<T>::load(x, (row0 + ti) * D + d0 + 8 * cg, v);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xs[r][8 * cg + j] = v[j];
  }
  for (int i = tid; i < kDwChunk * 8; i += 256) {
    const int r = i >> 3, cg = i & 7;
    float v[8];
    if (r < len) Dw8<T>::load(dy, (row0 + t0 + r) * D + d0 + 8 * cg, v);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) gs[r][8 * cg + j] = v[j];
  }
  __syncthreads();
  const int c = tid & 63, kg = tid >> 6;
  float acc[8], win[8], bsum = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] = 0.0f;
    win[j] = xs[8 * kg + j][c];                           // x[t0 + (8 kg + j) - pad]
  }
#pragma unroll 5
  for (int t = 0; t < kDwChunk; ++t) {                   // rows past `len` hold zero gradients
    const float g = gs[t][c];
    const float next = xs[t + 8 * kg + 8][c];
    bsum += g;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(g, win[j], acc[j]);
#pragma unroll
    for (int j = 0; j < 7; ++j) win[j] = win[j + 1];
    win[7] = next;
  }
  const long prow = static_cast<long>(blockIdx.y) * gridDim.z + blockIdx.z;
  float* out = partial + (prow * D + d0 + c) * kDwMaxTaps + 8 * kg;
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = acc[j];
  if (kg == 3) out[7] = bsum;                             // slot 31: never a tap (K <= 31)
}

static int dw_check(const char* who, int64_t B, int64_t T, int D, int K) {
  if (B <= 0 || T <= 0 || D <= 0) return fail(kErrInvalidArgument, std::string(who) + ": bad extents");
  if (D % 64 != 0) return fail(kErrUnsupported, std::string(who) + ": D must be a multiple of 64");
  if (K < 1 || K > 31 || K % 2 == 0) return fail(kErrUnsupported, std::string(who) + ": odd kernel size <= 31");
  if (T > 2147483647 / 4 || B > 65535) return fail(kErrUnsupported, std::string(who) + ": T or B too large");
  return kOk;
}

}  // namespace seld

extern "C" {

int seld_dwconv1d(const void* x, int is_bf16, const float* weight, const float* bias, int64_t B, int64_t T, int D, int K,
                  int flip_taps, void* y, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (int rc = dw_check("seld_dwconv1d", B, T, D, K)) return rc;
  if (!x || !weight || !y) return fail(kErrInvalidArgument, "seld_dwconv1d: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid(static_cast<unsigned>((T + 31) / 32), static_cast<unsigned>(D / 64), static_cast<unsigned>(B));
  if (is_bf16) hipLaunchKernelGGL(dwconv_kernel<__hip_bfloat16>, grid, dim3(256), 0, stream, x, weight, bias,
                                  static_cast<int>(T), D, K, flip_taps, y);
  else hipLaunchKernelGGL(dwconv_kernel<float>, grid, dim3(256), 0, stream, x, weight, bias, static_cast<int>(T), D, K,
                          flip_taps, y);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_dwconv1d_wgrad(const void* x, const void* dy, int is_bf16, int64_t B, int64_t T, int D, int K, float* partial,
                        void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (int rc = dw_check("seld_dwconv1d_wgrad", B, T, D, K)) return rc;
  if (!x || !dy || !partial) return fail(kErrInvalidArgument, "seld_dwconv1d_wgrad: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid(static_cast<unsigned>(D / 64), static_cast<unsigned>(B), static_cast<unsigned>((T + kDwChunk - 1) / kDwChunk));
  if (is_bf16) hipLaunchKernelGGL(dwconv_wgrad_kernel<__hip_bfloat16>, grid, dim3(256), 0, stream, x, dy,
                                  static_cast<int>(T), D, K, partial);
  else hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, stream, x, dy, static_cast<int>(T), D, K,
                          partial);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int64_t seld_dwconv1d_wgrad_rows(int64_t B, int64_t T) { return B * ((T + seld::kDwChunk - 1) / seld::kDwChunk); }

}  // extern "C"
