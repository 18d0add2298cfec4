// Multi-tensor dtype casts for the fp32-master / bf16-working-weight training mode (trainer.MasterWeightAdam).
//
// Under autocast every conv / Linear / GRU weight is cast fp32 -> bf16 once per iteration and every weight gradient
// bf16 -> fp32: one framework kernel each (36 per CRNN iteration, 336 per ResNet50-Conformer iteration, ~5 us
// apiece on a GPU-bound step).  Here ONE launch casts up to 96 tensors: blockIdx.y picks the tensor, blockIdx.x an
// 8192-element chunk of it; the (source, destination, length) descriptors travel BY VALUE in the kernel arguments
// (2.3 KB), so gradient tensors that autograd re-allocates every iteration need no descriptor upload.
// HBM-bound (6 bytes per element), round-to-nearest-even like the framework's casts.
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kCastThreads = 256;
constexpr int kCastPerThread = 32;                       // 4 x 8 elements
constexpr int kCastChunk = kCastThreads * kCastPerThread;

__device__ __forceinline__ unsigned cast_pack_bf16x2(float lo, float hi) {
  return static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(lo))) |
         (static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(hi))) << 16);
}

constexpr int kCastBatch = 96;
struct CastBatch {
  unsigned long long src[kCastBatch];
  unsigned long long dst[kCastBatch];
  long n[kCastBatch];
};

template <bool kToFloat>
__global__ __launch_bounds__(kCastThreads) void multi_cast_kernel(const CastBatch b) {
  const int t = blockIdx.y;
  const long n = b.n[t];
  const long base = static_cast<long>(blockIdx.x) * kCastChunk;
  if (base >= n) return;
  const unsigned long long src_addr = b.src[t], dst_addr = b.dst[t];
  const unsigned short* hsrc = reinterpret_cast<const unsigned short*>(src_addr);
  const float* fsrc = reinterpret_cast<const float*>(src_addr);
  unsigned short* hdst = reinterpret_cast<unsigned short*>(dst_addr);
  float* fdst = reinterpret_cast<float*>(dst_addr);
  const bool aligned = ((src_addr | dst_addr) & 15ull) == 0;
#pragma unroll
  for (int k = 0; k < kCastPerThread / 8; ++k) {
    const long i = base + (static_cast<long>(k) * kCastThreads + threadIdx.x) * 8;
    if (i >= n) break;
    if (aligned && i + 8 <= n) {
      if (kToFloat) {
        const uint4 v = *reinterpret_cast<const uint4*>(hsrc + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float4 a, b;
        a.x = __uint_as_float(w[0] << 16); a.y = __uint_as_float(w[0] & 0xffff0000u);
        a.z = __uint_as_float(w[1] << 16); a.w = __uint_as_float(w[1] & 0xffff0000u);
        b.x = __uint_as_float(w[2] << 16); b.y = __uint_as_float(w[2] & 0xffff0000u);
        b.z = __uint_as_float(w[3] << 16); b.w = __uint_as_float(w[3] & 0xffff0000u);
        *reinterpret_cast<float4*>(fdst + i) = a;
        *reinterpret_cast<float4*>(fdst + i + 4) = b;
      } else {
        const float4 a = *reinterpret_cast<const float4*>(fsrc + i);
        const float4 b = *reinterpret_cast<const float4*>(fsrc + i + 4);
        uint4 v;
        v.x = cast_pack_bf16x2(a.x, a.y);
        v.y = cast_pack_bf16x2(a.z, a.w);
        v.z = cast_pack_bf16x2(b.x, b.y);
        v.w = cast_pack_bf16x2(b.z, b.w);
        *reinterpret_cast<uint4*>(hdst + i) = v;
      }
    } else {
      for (long j = i; j < n && j < i + 8; ++j) {
        if (kToFloat) fdst[j] = __uint_as_float(static_cast<unsigned>(hsrc[j]) << 16);
        else hdst[j] = __bfloat16_as_ushort(__float2bfloat16(fsrc[j]));
      }
    }
  }
}

}  // namespace seld

// A one-wavefront kernel that holds its stream for a given time: seld_overlap puts it at the head of the side stream
// so that the persistent BiGRU recurrence (16 workgroups that each need a whole CU's LDS) is resident before the
// weight-gradient GEMMs flood every CU.  Bounded by the constant 100 MHz wall clock: every launch terminates.
namespace seld {
__global__ void __launch_bounds__(64) stream_delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace seld

extern "C" {

int seld_multi_cast(const void* const* src, void* const* dst, const int64_t* lengths, int count, int bf16_to_fp32,
                    void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (count < 0) return fail(kErrInvalidArgument, "seld_multi_cast: negative count");
  if (count == 0) return kOk;
  if (!src || !dst || !lengths) return fail(kErrInvalidArgument, "seld_multi_cast: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  for (int first = 0; first < count; first += kCastBatch) {
    CastBatch b;
    const int here = count - first < kCastBatch ? count - first : kCastBatch;
    long longest = 0;
    for (int i = 0; i < here; ++i) {
      if (lengths[first + i] < 0 || (lengths[first + i] > 0 && (!src[first + i] || !dst[first + i])))
        return fail(kErrInvalidArgument, "seld_multi_cast: bad tensor descriptor");
      b.src[i] = reinterpret_cast<unsigned long long>(src[first + i]);
      b.dst[i] = reinterpret_cast<unsigned long long>(dst[first + i]);
      b.n[i] = lengths[first + i];
      if (b.n[i] > longest) longest = b.n[i];
    }
    if (longest == 0) continue;
    const dim3 grid(static_cast<unsigned>((longest + kCastChunk - 1) / kCastChunk), static_cast<unsigned>(here));
    if (bf16_to_fp32) hipLaunchKernelGGL(multi_cast_kernel<true>, grid, dim3(kCastThreads), 0, stream, b);
    else hipLaunchKernelGGL(multi_cast_kernel<false>, grid, dim3(kCastThreads), 0, stream, b);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_stream_delay(int64_t nanoseconds, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (nanoseconds < 0 || nanoseconds > 1000000) return fail(kErrInvalidArgument, "seld_stream_delay: 0 .. 1e6 ns");
  if (nanoseconds == 0) return kOk;
  hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_),
                     static_cast<long long>(nanoseconds / 10));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
