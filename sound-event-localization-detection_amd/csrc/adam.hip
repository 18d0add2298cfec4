// Multi-tensor Adam for the fp32-master / bf16-working-weight training mode (trainer.MasterWeightAdam), gfx950.
//
// The optimiser step of trainer.py:112-116, 179 upstream -- torch.optim.Adam(lr, weight_decay as L2 added to the gradient)
// -- ran here as THREE multi-tensor launches: bf16 gradients -> fp32 (seld_multi_cast), the framework's fused Adam on the
// fp32 masters, fp32 masters -> bf16 working copies (seld_multi_cast): 40 bytes of HBM traffic per parameter.  This
// kernel reads the bf16 gradient itself and writes the bf16 working copy itself: gradient 2 (or 4) + master 4 + exp_avg 4
// + exp_avg_sq 4 read, 4 + 4 + 4 (+ 2) written = 28 B per parameter, one launch per 48 tensors, descriptors by value
// (the seld_multi_cast pattern: gradient tensors that autograd re-allocates every iteration need no descriptor upload).
// Pure HBM-bound elementwise work.
//
// Arithmetic = the framework's fused kernel (ATen fused_adam_utils.cuh, ADAM_MODE::ORIGINAL, amsgrad off, maximize off),
// in fp32:   g += weight_decay * p;   m = m + (1 - beta1) (g - m);   v = beta2 v + (1 - beta2) g g;
//            p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
// with lr and step read from DEVICE scalars (graph replay: a ReduceLROnPlateau change needs no re-capture; the caller
// increments step before the launch).
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kAdamThreads = 256;
constexpr int kAdamPerThread = 16;                       // 2 x 8 elements
constexpr int kAdamChunk = kAdamThreads * kAdamPerThread;
constexpr int kAdamBatch = 48;

struct AdamBatch {
  unsigned long long grad[kAdamBatch];                   // bf16 or fp32 (flags bit 0: bf16)
  unsigned long long param[kAdamBatch];                  // fp32 master / parameter
  unsigned long long m[kAdamBatch];                      // exp_avg, fp32
  unsigned long long v[kAdamBatch];                      // exp_avg_sq, fp32
  unsigned long long low[kAdamBatch];                    // bf16 working copy, or 0
  long n[kAdamBatch];
  int first_block[kAdamBatch];                           // work list: no empty workgroups
  int flags[kAdamBatch];
  int count;
};

struct AdamScalars {
  const float* lr;
  const float* step;
  float beta1, beta2, eps, weight_decay, grad_scale;
};

__device__ __forceinline__ unsigned adam_pack_bf16x2(float lo, float hi) {
  return static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(lo))) |
         (static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(hi))) << 16);
}

__global__ __launch_bounds__(kAdamThreads) void multi_adam_kernel(const AdamBatch b, const AdamScalars s) {
  int t = 0;
  while (t + 1 < b.count && static_cast<int>(blockIdx.x) >= b.first_block[t + 1]) ++t;      // uniform
  const long n = b.n[t];
  const long base = static_cast<long>(static_cast<int>(blockIdx.x) - b.first_block[t]) * kAdamChunk;
  const bool grad_bf16 = b.flags[t] & 1;
  const unsigned short* gh = reinterpret_cast<const unsigned short*>(b.grad[t]);
  const float* gf = reinterpret_cast<const float*>(b.grad[t]);
  float* p = reinterpret_cast<float*>(b.param[t]);
  float* m = reinterpret_cast<float*>(b.m[t]);
  float* v = reinterpret_cast<float*>(b.v[t]);
  unsigned short* low = reinterpret_cast<unsigned short*>(b.low[t]);
  const float lr = *s.lr, step = *s.step;
  const float bc1 = 1.0f - __powf(s.beta1, step), bc2 = 1.0f - __powf(s.beta2, step);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  const bool aligned = ((b.grad[t] | b.param[t] | b.m[t] | b.v[t] | b.low[t]) & 15ull) == 0;
#pragma unroll
  for (int k = 0; k < kAdamPerThread / 8; ++k) {
    const long i = base + (static_cast<long>(k) * kAdamThreads + threadIdx.x) * 8;
    if (i >= n) break;
    float g[8], pp[8], mm[8], vv[8];
    const bool full = aligned && i + 8 <= n;
    if (full) {
      if (grad_bf16) {
        const uint4 w = *reinterpret_cast<const uint4*>(gh + i);
        const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[2 * j] = __uint_as_float(ww[j] << 16);
          g[2 * j + 1] = __uint_as_float(ww[j] & 0xffff0000u);
        }
      } else {
        const float4 a = *reinterpret_cast<const float4*>(gf + i), c = *reinterpret_cast<const float4*>(gf + i + 4);
        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = c.x; g[5] = c.y; g[6] = c.z; g[7] = c.w;
      }
      const float4 p0 = *reinterpret_cast<const float4*>(p + i), p1 = *reinterpret_cast<const float4*>(p + i + 4);
      const float4 m0 = *reinterpret_cast<const float4*>(m + i), m1 = *reinterpret_cast<const float4*>(m + i + 4);
      const float4 v0 = *reinterpret_cast<const float4*>(v + i), v1 = *reinterpret_cast<const float4*>(v + i + 4);
      pp[0] = p0.x; pp[1] = p0.y; pp[2] = p0.z; pp[3] = p0.w; pp[4] = p1.x; pp[5] = p1.y; pp[6] = p1.z; pp[7] = p1.w;
      mm[0] = m0.x; mm[1] = m0.y; mm[2] = m0.z; mm[3] = m0.w; mm[4] = m1.x; mm[5] = m1.y; mm[6] = m1.z; mm[7] = m1.w;
      vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long e = i + j < n ? i + j : n - 1;
        g[j] = grad_bf16 ? __uint_as_float(static_cast<unsigned>(gh[e]) << 16) : gf[e];
        pp[j] = p[e];
        mm[j] = m[e];
        vv[j] = v[e];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gj = g[j] * s.grad_scale;
      gj = fmaf(s.weight_decay, pp[j], gj);
      mm[j] = fmaf(1.0f - s.beta1, gj - mm[j], mm[j]);
      vv[j] = fmaf(s.beta2, vv[j], (1.0f - s.beta2) * gj * gj);
      const float denom = sqrtf(vv[j]) / bc2_sqrt + s.eps;
      pp[j] -= step_size * mm[j] / denom;
    }
    if (full) {
      *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *reinterpret_cast<float4*>(p + i + 4) = make_float4(pp[4], pp[5], pp[6], pp[7]);
      *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *reinterpret_cast<float4*>(m + i + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
      *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      *reinterpret_cast<float4*>(v + i + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
      if (low) {
        uint4 w;
        w.x = adam_pack_bf16x2(pp[0], pp[1]);
        w.y = adam_pack_bf16x2(pp[2], pp[3]);
        w.z = adam_pack_bf16x2(pp[4], pp[5]);
        w.w = adam_pack_bf16x2(pp[6], pp[7]);
        *reinterpret_cast<uint4*>(low + i) = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i + j < n) {
          p[i + j] = pp[j];
          m[i + j] = mm[j];
          v[i + j] = vv[j];
          if (low) low[i + j] = __bfloat16_as_ushort(__float2bfloat16(pp[j]));
        }
      }
    }
  }
}

}  // namespace seld

extern "C" {

int seld_multi_adam(const void* const* grad, const int32_t* grad_is_bf16, float* const* param, float* const* exp_avg,
                    float* const* exp_avg_sq, void* const* low_bf16, const int64_t* lengths, int count, const float* lr,
                    const float* step, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (count < 0) return fail(kErrInvalidArgument, "seld_multi_adam: negative count");
  if (count == 0) return kOk;
  if (!grad || !grad_is_bf16 || !param || !exp_avg || !exp_avg_sq || !low_bf16 || !lengths || !lr || !step)
    return fail(kErrInvalidArgument, "seld_multi_adam: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const AdamScalars s{lr, step, beta1, beta2, eps, weight_decay, grad_scale};
  for (int first = 0; first < count; first += kAdamBatch) {
    AdamBatch b;
    const int here = count - first < kAdamBatch ? count - first : kAdamBatch;
    long blocks = 0;
    for (int i = 0; i < here; ++i) {
      const int k = first + i;
      if (lengths[k] <= 0 || !grad[k] || !param[k] || !exp_avg[k] || !exp_avg_sq[k])
        return fail(kErrInvalidArgument, "seld_multi_adam: bad tensor descriptor");
      b.grad[i] = reinterpret_cast<unsigned long long>(grad[k]);
      b.param[i] = reinterpret_cast<unsigned long long>(param[k]);
      b.m[i] = reinterpret_cast<unsigned long long>(exp_avg[k]);
      b.v[i] = reinterpret_cast<unsigned long long>(exp_avg_sq[k]);
      b.low[i] = reinterpret_cast<unsigned long long>(low_bf16[k]);
      b.n[i] = lengths[k];
      b.flags[i] = grad_is_bf16[k] ? 1 : 0;
      b.first_block[i] = static_cast<int>(blocks);
      blocks += (lengths[k] + kAdamChunk - 1) / kAdamChunk;
      if (blocks >= (1L << 31)) return fail(kErrUnsupported, "seld_multi_adam: too many elements for one launch");
    }
    b.count = here;
    hipLaunchKernelGGL(multi_adam_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kAdamThreads), 0, stream, b, s);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
