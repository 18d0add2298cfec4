// Fused softmax + MSE class loss (forward value and gradient in one pass) for gfx950.
//
// Replaces loss.py:43-54 (class_mse_loss: F.softmax + F.mse_loss over [B,T,648,14]) and its
// autograd backward: unfused that is >= 6 passes over 9.07 MB-per-window tensors plus a dense
// 290 MB/step label upload; here the logits are read once, the gradient written once, and the
// labels come either as the dense float tensor the reference uses or as the compact uint16
// class mask (labels.hip), expanded in registers.
//
//   p = softmax(z),  L = mean_{cells,classes} (p - y)^2
//   dL/dz_k = (2/(n*M)) * p_k * [ (p_k - y_k) - sum_c p_c (p_c - y_c) ]
//
// HBM-bound: 56 B read + 56 B written per cell (fp32 logits, M = 14).  A 256-thread block
// stages 256 cells (3584 floats) through LDS so that global traffic is full 16-B lanes while
// each thread owns one cell (stride-14 LDS reads as 7 x ds_read_b64: conflict free).
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kLossBlock = 256;

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) {
  return __uint_as_float(static_cast<unsigned>(b) << 16);
}

__device__ __forceinline__ unsigned short float_to_bf16_bits(float f) {
  return __bfloat16_as_ushort(__float2bfloat16(f));   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
}

template <int M, bool kBf16, bool kMaskLabels, bool kGrad>
__global__ __launch_bounds__(kLossBlock) void softmax_mse_kernel(const void* __restrict__ logits_,
                                                                  const uint16_t* __restrict__ mask,
                                                                  const float* __restrict__ dense,
                                                                  int n_cells, float grad_scale,
                                                                  double* __restrict__ partials,
                                                                  void* __restrict__ grad_) {
  __shared__ __attribute__((aligned(16))) float tile[kLossBlock * M];
  __shared__ double wave_sums[kLossBlock / 64];
  const int tid = threadIdx.x;
  double local = 0.0;
  // All index arithmetic is 32-bit on purpose (the host checks n_cells * M < 2^31).  With 64-bit extents hipcc
  // (ROCm 7.2, gfx950) emitted, for some instantiations of this kernel, a v_cmp_lt_i64 followed by s_cselect
  // WITHOUT copying VCC to SCC for `min(n_cells - cell0, 256)`: the ragged last tile was then processed as a full
  // one and read 256 cells of whatever lies behind the logits / labels (wrong sums, NaN, or a memory fault).
  // s_min_i32 cannot go wrong that way; tests/test_loss_gpu.py::test_every_variant_at_ragged_sizes covers it.
  const int n_tiles = (n_cells + kLossBlock - 1) / kLossBlock;

  for (int tileno = blockIdx.x; tileno < n_tiles; tileno += gridDim.x) {
    const int cell0 = tileno * kLossBlock;
    const int cells_here = min(n_cells - cell0, kLossBlock);
    const int elems = cells_here * M;
    // ---- coalesced load of the logits tile into LDS (as fp32)
    if (kBf16 && cells_here == kLossBlock) {
      // full tile: 7168 B = 448 sixteen-byte pieces (the tile base is 16-byte aligned: 256 cells x 28 B)
      const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(logits_) + static_cast<long>(cell0) * M);
      for (int piece = tid; piece < kLossBlock * M / 8; piece += kLossBlock) {
        const uint4 v = src[piece];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f[2 * k] = __uint_as_float(w[k] << 16);
          f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
        *reinterpret_cast<float4*>(tile + piece * 8) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(tile + piece * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
      }
    } else if (kBf16) {
      const unsigned short* src = static_cast<const unsigned short*>(logits_) + static_cast<long>(cell0) * M;
      for (int i = tid * 2; i < elems; i += kLossBlock * 2) {   // elems is even (M = 14)
        const unsigned v = *reinterpret_cast<const unsigned*>(src + i);
        tile[i] = bf16_bits_to_float(static_cast<unsigned short>(v & 0xffffu));
        tile[i + 1] = bf16_bits_to_float(static_cast<unsigned short>(v >> 16));
      }
    } else {
      const float* src = static_cast<const float*>(logits_) + static_cast<long>(cell0) * M;
      if ((cell0 * M) % 4 == 0 && elems % 4 == 0) {
        for (int i = tid * 4; i < elems; i += kLossBlock * 4)
          *reinterpret_cast<float4*>(tile + i) = *reinterpret_cast<const float4*>(src + i);
      } else {
        for (int i = tid; i < elems; i += kLossBlock) tile[i] = src[i];
      }
    }
    __syncthreads();

    // ---- one cell per thread
    float g[M];
    if (tid < cells_here) {
      float z[M], y[M];
#pragma unroll
      for (int c = 0; c < M; c += 2) {
        const float2 v = *reinterpret_cast<const float2*>(tile + tid * M + c);
        z[c] = v.x;
        z[c + 1] = v.y;
      }
      if (kMaskLabels) {
        const unsigned m = mask[cell0 + tid];
#pragma unroll
        for (int c = 0; c < M; ++c) y[c] = ((m >> c) & 1u) ? 1.0f : 0.0f;
        if (m == 0u) y[M - 1] = 1.0f;                       // background rule, dataset.py:114-117
      } else {
        const float* yp = dense + static_cast<long>(cell0 + tid) * M;
#pragma unroll
        for (int c = 0; c < M; c += 2) {
          const float2 v = *reinterpret_cast<const float2*>(yp + c);
          y[c] = v.x;
          y[c + 1] = v.y;
        }
      }
      float zmax = z[0];
#pragma unroll
      for (int c = 1; c < M; ++c) zmax = fmaxf(zmax, z[c]);
      float denom = 0.0f;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        z[c] = __expf(z[c] - zmax);
        denom += z[c];
      }
      const float inv = 1.0f / denom;
      float sq = 0.0f, dot = 0.0f;
#pragma unroll
      for (int c = 0; c < M; ++c) {
        const float p = z[c] * inv;
        const float d = p - y[c];
        sq = fmaf(d, d, sq);
        dot = fmaf(p, d, dot);
        z[c] = p;
        y[c] = d;
      }
      local += static_cast<double>(sq);
      if (kGrad) {
#pragma unroll
        for (int c = 0; c < M; ++c) g[c] = grad_scale * z[c] * (y[c] - dot);
      }
    }
    if (kGrad) {
      __syncthreads();   // everyone has read its logits: reuse the tile for the gradient
      if (tid < cells_here) {
#pragma unroll
        for (int c = 0; c < M; c += 2) *reinterpret_cast<float2*>(tile + tid * M + c) = make_float2(g[c], g[c + 1]);
      }
      __syncthreads();
      if (kBf16 && cells_here == kLossBlock) {
        uint4* dst = reinterpret_cast<uint4*>(static_cast<unsigned short*>(grad_) + static_cast<long>(cell0) * M);
        for (int piece = tid; piece < kLossBlock * M / 8; piece += kLossBlock) {
          const float4 a = *reinterpret_cast<const float4*>(tile + piece * 8);
          const float4 b = *reinterpret_cast<const float4*>(tile + piece * 8 + 4);
          uint4 v;
          v.x = float_to_bf16_bits(a.x) | (static_cast<unsigned>(float_to_bf16_bits(a.y)) << 16);
          v.y = float_to_bf16_bits(a.z) | (static_cast<unsigned>(float_to_bf16_bits(a.w)) << 16);
          v.z = float_to_bf16_bits(b.x) | (static_cast<unsigned>(float_to_bf16_bits(b.y)) << 16);
          v.w = float_to_bf16_bits(b.z) | (static_cast<unsigned>(float_to_bf16_bits(b.w)) << 16);
          dst[piece] = v;
        }
      } else if (kBf16) {
        unsigned short* dst = static_cast<unsigned short*>(grad_) + static_cast<long>(cell0) * M;
        for (int i = tid * 2; i < elems; i += kLossBlock * 2) {
          const unsigned lo = float_to_bf16_bits(tile[i]);
          const unsigned hi = float_to_bf16_bits(tile[i + 1]);
          *reinterpret_cast<unsigned*>(dst + i) = lo | (hi << 16);
        }
      } else {
        float* dst = static_cast<float*>(grad_) + static_cast<long>(cell0) * M;
        if ((cell0 * M) % 4 == 0 && elems % 4 == 0) {
          for (int i = tid * 4; i < elems; i += kLossBlock * 4)
            *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(tile + i);
        } else {
          for (int i = tid; i < elems; i += kLossBlock) dst[i] = tile[i];
        }
      }
    }
    __syncthreads();
  }

  // ---- deterministic block reduction of the squared-error sum (double)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((tid & 63) == 0) wave_sums[tid >> 6] = local;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < kLossBlock / 64; ++w) s += wave_sums[w];
    partials[blockIdx.x] = s;
  }
}

// Fixed-order final reduction -> mean squared error as float (256 lanes, deterministic tree).
__global__ __launch_bounds__(256) void loss_finish_kernel(const double* __restrict__ partials, int n, double inv_count,
                                                          float* __restrict__ out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<float>(part[0] * inv_count);
}

// grad *= *scale, where the scale is a DEVICE scalar (the upstream gradient of the loss, known only to the stream).
// loss.backward() hands the loss an upstream gradient of exactly 1: every block then returns after one load and the
// 145 MB gradient is not touched again -- without the host ever reading the scalar.
template <bool kBf16>
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(void* __restrict__ data, long n8,
                                                              const float* __restrict__ scale) {
  const float s = *scale;
  if (s == 1.0f) return;
  for (long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; i < n8; i += static_cast<long>(gridDim.x) * 256) {
    if (kBf16) {
      uint4 v = static_cast<uint4*>(data)[i];
      unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(w[k] << 16) * s, hi = __uint_as_float(w[k] & 0xffff0000u) * s;
        w[k] = float_to_bf16_bits(lo) | (static_cast<unsigned>(float_to_bf16_bits(hi)) << 16);
      }
      static_cast<uint4*>(data)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      float4* p = static_cast<float4*>(data) + 2 * i;
      float4 a = p[0], b = p[1];
      a.x *= s; a.y *= s; a.z *= s; a.w *= s;
      b.x *= s; b.y *= s; b.z *= s; b.w *= s;
      p[0] = a;
      p[1] = b;
    }
  }
}

template <bool kBf16, bool kMaskLabels, bool kGrad>
static void launch(unsigned blocks, hipStream_t stream, const void* logits, const uint16_t* mask, const float* dense,
                   int n_cells, float grad_scale, double* partials, void* grad) {
  hipLaunchKernelGGL((softmax_mse_kernel<14, kBf16, kMaskLabels, kGrad>), dim3(blocks), dim3(kLossBlock), 0, stream,
                     logits, mask, dense, n_cells, grad_scale, partials, grad);
}

}  // namespace seld

extern "C" {

int64_t seld_softmax_mse_workspace_bytes(void) { return 4096 * sizeof(double); }

int seld_scale_by_device_scalar(void* data, int is_bf16, int64_t n, const float* scale, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (n < 0 || n % 8 != 0) return fail(kErrInvalidArgument, "seld_scale_by_device_scalar: n must be a multiple of 8");
  if (n == 0) return kOk;
  if (!data || !scale) return fail(kErrInvalidArgument, "seld_scale_by_device_scalar: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long n8 = n / 8;
  long blocks = (n8 + 255) / 256;
  const long cap = static_cast<long>(st->num_cus > 0 ? st->num_cus : 256) * 8;
  if (blocks > cap) blocks = cap;
  if (is_bf16) hipLaunchKernelGGL(scale_by_scalar_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                                  stream, data, n8, scale);
  else hipLaunchKernelGGL(scale_by_scalar_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                          data, n8, scale);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_softmax_mse(const void* logits, int logits_is_bf16, const uint16_t* mask, const float* dense_labels,
                     int64_t n_cells, int num_classes, float grad_scale, float* loss_out, void* grad,
                     void* workspace, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (num_classes != 14) return fail(kErrUnsupported, "seld_softmax_mse: built for 14 classes (config.py:40)");
  if (n_cells <= 0) return fail(kErrInvalidArgument, "seld_softmax_mse: n_cells must be positive");
  if (n_cells > (2147483647LL - 4096) / num_classes)
    return fail(kErrUnsupported, "seld_softmax_mse: n_cells * num_classes must stay below 2^31 (32-bit tile indexing)");
  if (!logits || !loss_out || !workspace) return fail(kErrInvalidArgument, "seld_softmax_mse: null pointer");
  if ((mask == nullptr) == (dense_labels == nullptr))
    return fail(kErrInvalidArgument, "seld_softmax_mse: pass exactly one of mask / dense_labels");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  long tiles = (n_cells + kLossBlock - 1) / kLossBlock;
  long blocks = static_cast<long>(st->num_cus) * 8;
  if (blocks > tiles) blocks = tiles;
  if (blocks > 4096) blocks = 4096;
  double* partials = static_cast<double*>(workspace);
  const unsigned nb = static_cast<unsigned>(blocks);
  const bool bf = logits_is_bf16 != 0, mk = mask != nullptr, gr = grad != nullptr;
  const int n = static_cast<int>(n_cells);
#define SELD_DISPATCH(B, K, G) \
  if (bf == B && mk == K && gr == G) launch<B, K, G>(nb, stream, logits, mask, dense_labels, n, grad_scale, partials, grad)
  SELD_DISPATCH(false, false, false); SELD_DISPATCH(false, false, true);
  SELD_DISPATCH(false, true, false);  SELD_DISPATCH(false, true, true);
  SELD_DISPATCH(true, false, false);  SELD_DISPATCH(true, false, true);
  SELD_DISPATCH(true, true, false);   SELD_DISPATCH(true, true, true);
#undef SELD_DISPATCH
  SELD_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, stream, partials, static_cast<int>(nb),
                     1.0 / (static_cast<double>(n_cells) * num_classes), loss_out);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
