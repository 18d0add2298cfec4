// Persistent bidirectional GRU recurrence for gfx950 (forward and backward).
//
// Replaces the recurrent part of nn.GRU(2048 -> 256, 2 layers, bidirectional) at
// model_crnn.py:65-72 / :119.  The input projection  gi = x W_ih^T + b_ih  for all time steps is
// one large GEMM done by the host (hipBLASLt, MFMA); what remains is 250 strictly sequential
// steps per direction of
//     gh = h W_hh^T + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ; n = tanh(gi_n + r*gh_n)
//     h' = (1 - z) n + z h
// which a library executes as ~2 tiny kernels per step.  Here ONE workgroup per (direction,
// 16-sequence batch tile) runs all T steps with W_hh resident on the CU:
//   * 8 wavefronts; wavefront w owns hidden units [32w, 32w+32) for all three gates, so the gate
//     math of a (sequence, unit) pair is lane-local and h stays in fp32 registers;
//   * W_hh (bf16, 393 KB per direction) does not fit one place: the r and z gate rows live in
//     VGPRs as MFMA B-fragments (128 registers per lane), the n gate rows in LDS (128 KB) laid
//     out fragment-major so every ds_read_b128 is a linear conflict-free 1 KB read;
//   * h_{t-1} (bf16) is exchanged through a double-buffered 16 x 256 LDS tile (row pitch 528 B:
//     conflict-free A-fragment reads), one barrier per step;
//   * v_mfma_f32_16x16x32_bf16: per step 48 MFMAs per wavefront, fp32 accumulate, fp32 gates.
// At M = 16 rows the step is MFMA-issue bound on its CU (~0.65 us); the 250-step recurrence of one
// layer costs about a quarter of a millisecond instead of hundreds of launches.
//
// The backward kernel mirrors it: dgh (bf16) goes through LDS as the A operand, W_hh^T fragments
// are register / LDS resident, dh is carried in registers; it emits the per-step gate gradients
// from which the host forms dW_ih, dW_hh, dx with three large GEMMs.
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kH = 256;            // hidden size (config.py:45 CRNN_RNN_HIDDEN)
constexpr int kG = 3 * kH;         // gate rows r | z | n
constexpr int kRows = 16;          // sequences per workgroup (MFMA M)
constexpr int kGruThreads = 512;   // 8 wavefronts
constexpr int kHPitch = kH + 8;    // bf16 elements per h row in LDS (528 B)
constexpr int kDghPitch = kG + 8;  // bf16 elements per dgh row in LDS (1552 B)
constexpr int kWnBytes = 8 * 2 * 8 * 64 * 16;   // [wave][tile][kstep][lane] x 16 B = 131072

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_float<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float c = fminf(fmaxf(x, -15.0f), 15.0f);
  const float e = __expf(2.0f * c);
  return (e - 1.0f) / (e + 1.0f);
}

struct GruFwdArgs {
  const void* gi;        // [B][T][2][3H]   (x W_ih^T + b_ih, both directions)
  const __hip_bfloat16* w_hh;   // [2][3H][H]
  const float* b_hh;     // [2][3H]
  void* y;               // [B][T][2H]
  float* saved;          // [B][T][2][4][H]  r, z, n, gh_n, always fp32 (nullptr: inference)
  long B, T;
};

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_forward_kernel(GruFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);
  __hip_bfloat16* hbuf = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);   // [2][16][kHPitch]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int dir = blockIdx.y;
  const long row0 = static_cast<long>(blockIdx.x) * kRows;
  const __hip_bfloat16* w = a.w_hh + static_cast<long>(dir) * kG * kH;
  const float* bh = a.b_hh + dir * kG;
  const T* gi = static_cast<const T*>(a.gi);
  T* y = static_cast<T*>(a.y);
  float* saved = a.saved;

  // ---- resident weights: r,z gates -> registers, n gate -> LDS (fragment-major)
  bf16x8 wr[2][8], wz[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * wave + 16 * s + c;              // B-fragment column = gate row of this unit
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k0 = 32 * kk + 8 * q;
      wr[s][kk] = *reinterpret_cast<const bf16x8*>(w + static_cast<long>(unit) * kH + k0);
      wz[s][kk] = *reinterpret_cast<const bf16x8*>(w + static_cast<long>(kH + unit) * kH + k0);
      wn_lds[((wave * 2 + s) * 8 + kk) * 64 + lane] =
          *reinterpret_cast<const bf16x8*>(w + static_cast<long>(2 * kH + unit) * kH + k0);
    }
  }
  for (int i = tid; i < 2 * kRows * kHPitch; i += kGruThreads) hbuf[i] = __float2bfloat16(0.0f);

  float bias_r[2], bias_z[2], bias_n[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * wave + 16 * s + c;
    bias_r[s] = bh[unit];
    bias_z[s] = bh[kH + unit];
    bias_n[s] = bh[2 * kH + unit];
  }
  float h_prev[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) h_prev[s][i] = 0.0f;
  __syncthreads();

  // gi of step `t` for this lane's (row 4q+i, unit) pairs: [s][gate][i]
  auto load_gi = [&](long tt, float (&g)[2][3][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long b = row0 + 4 * q + i;
      const bool ok = b < a.B;
      const T* p = gi + ((ok ? b : 0) * a.T + tt) * (2 * kG) + dir * kG;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int unit = 32 * wave + 16 * s + c;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) g[s][gate][i] = ok ? to_float(p[gate * kH + unit]) : 0.0f;
      }
    }
  };

  float g_cur[2][3][4];
  load_gi(dir == 0 ? 0 : a.T - 1, g_cur);

  for (long t = 0; t < a.T; ++t) {
    const long tt = dir == 0 ? t : a.T - 1 - t;
    const int cur = static_cast<int>(t & 1), nxt = cur ^ 1;
    float g_next[2][3][4];
    if (t + 1 < a.T) load_gi(dir == 0 ? t + 1 : a.T - 2 - t, g_next);

    // ---- gh = h W_hh^T : A fragments of h_{t-1} from LDS
    const __hip_bfloat16* hrow = hbuf + (cur * kRows + c) * kHPitch + 8 * q;

    f32x4 acc_r[2], acc_z[2], acc_n[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc_r[s][i] = g_cur[s][0][i] + bias_r[s];
        acc_z[s][i] = g_cur[s][1][i] + bias_z[s];
        acc_n[s][i] = bias_n[s];
      }
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(hrow + 32 * kk);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc_r[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, wr[s][kk], acc_r[s], 0, 0, 0);
        acc_z[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, wz[s][kk], acc_z[s], 0, 0, 0);
        const bf16x8 wn = wn_lds[((wave * 2 + s) * 8 + kk) * 64 + lane];
        acc_n[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, wn, acc_n[s], 0, 0, 0);
      }
    }

    // ---- gates (lane-local) and state update
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int unit = 32 * wave + 16 * s + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float r = sigmoid_f(acc_r[s][i]);
        const float z = sigmoid_f(acc_z[s][i]);
        const float ghn = acc_n[s][i];
        const float n = tanh_f(fmaf(r, ghn, g_cur[s][2][i]));
        const float h = fmaf(z, h_prev[s][i] - n, n);
        h_prev[s][i] = h;
        const int row = 4 * q + i;
        hbuf[(nxt * kRows + row) * kHPitch + unit] = __float2bfloat16(h);
        const long b = row0 + row;
        if (b < a.B) {
          y[(b * a.T + tt) * (2 * kH) + dir * kH + unit] = from_float<T>(h);
          if (saved) {
            float* sp = saved + (((b * a.T + tt) * 2 + dir) * 4) * kH + unit;
            sp[0] = r;
            sp[kH] = z;
            sp[2 * kH] = n;
            sp[3 * kH] = ghn;
          }
        }
      }
    }
    __syncthreads();
    if (t + 1 < a.T) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
          for (int i = 0; i < 4; ++i) g_cur[s][gate][i] = g_next[s][gate][i];
    }
  }
}

struct GruBwdArgs {
  const void* dy;        // [B][T][2H]
  const void* y;         // [B][T][2H]        forward outputs (h_t)
  const float* saved;    // [B][T][2][4][H]   r, z, n, gh_n (fp32)
  const __hip_bfloat16* w_hh_t;   // [2][H][3H]   W_hh transposed per direction
  void* dg;              // [B][T][2][4][H]   da_r, da_z, da_n, da_n * r
  long B, T;
};

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_backward_kernel(GruBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);                               // k-range of the n gate
  __hip_bfloat16* dgh = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);       // [16][kDghPitch]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int dir = blockIdx.y;
  const long row0 = static_cast<long>(blockIdx.x) * kRows;
  const __hip_bfloat16* wt = a.w_hh_t + static_cast<long>(dir) * kH * kG;         // [H][3H]
  const T* dy = static_cast<const T*>(a.dy);
  const T* y = static_cast<const T*>(a.y);
  const float* saved = a.saved;
  T* dg = static_cast<T*>(a.dg);

  // dh_prev[m][u] = sum_k dgh[m][k] W_hh[k][u]: B fragment (k, n=u) = W_hh^T[u][k..k+7]
  bf16x8 wrz[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * wave + 16 * s + c;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      wrz[s][kk] = *reinterpret_cast<const bf16x8*>(wt + static_cast<long>(unit) * kG + 32 * kk + 8 * q);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      wn_lds[((wave * 2 + s) * 8 + kk) * 64 + lane] =
          *reinterpret_cast<const bf16x8*>(wt + static_cast<long>(unit) * kG + 2 * kH + 32 * kk + 8 * q);
  }
  float dh[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) dh[s][i] = 0.0f;
  __syncthreads();

  for (long t = a.T - 1; t >= 0; --t) {            // reverse of the forward processing order
    const long tt = dir == 0 ? t : a.T - 1 - t;
    const long tprev = dir == 0 ? tt - 1 : tt + 1;  // time index of h_{t-1} in the forward recurrence
    const bool has_prev = t > 0;
    float keep[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int unit = 32 * wave + 16 * s + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * q + i;
        const long b = row0 + row;
        float da_r = 0.0f, da_z = 0.0f, da_n = 0.0f, dghn = 0.0f, carry = 0.0f;
        if (b < a.B) {
          const float* sp = saved + (((b * a.T + tt) * 2 + dir) * 4) * kH + unit;
          const float r = sp[0], z = sp[kH], n = sp[2 * kH], ghn = sp[3 * kH];
          const float hp = has_prev ? to_float(y[(b * a.T + tprev) * (2 * kH) + dir * kH + unit]) : 0.0f;
          const float dtot = to_float(dy[(b * a.T + tt) * (2 * kH) + dir * kH + unit]) + dh[s][i];
          const float dn = dtot * (1.0f - z);
          const float dz = dtot * (hp - n);
          da_n = dn * (1.0f - n * n);
          da_z = dz * z * (1.0f - z);
          da_r = da_n * ghn * r * (1.0f - r);
          dghn = da_n * r;
          carry = dtot * z;
          T* gp = dg + (((b * a.T + tt) * 2 + dir) * 4) * kH + unit;
          gp[0] = from_float<T>(da_r);
          gp[kH] = from_float<T>(da_z);
          gp[2 * kH] = from_float<T>(da_n);
          gp[3 * kH] = from_float<T>(dghn);
        }
        keep[s][i] = carry;
        __hip_bfloat16* drow = dgh + row * kDghPitch + unit;
        drow[0] = __float2bfloat16(da_r);
        drow[kH] = __float2bfloat16(da_z);
        drow[2 * kH] = __float2bfloat16(dghn);
      }
    }
    __syncthreads();

    f32x4 acc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[s][i] = keep[s][i];
    const __hip_bfloat16* arow = dgh + c * kDghPitch + 8 * q;
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
      const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(arow + 32 * kk);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bfrag = kk < 16 ? wrz[s][kk < 16 ? kk : 0] : wn_lds[((wave * 2 + s) * 8 + (kk - 16)) * 64 + lane];
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag, acc[s], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) dh[s][i] = acc[s][i];
    __syncthreads();
  }
}

}  // namespace seld

extern "C" {

int seld_gru_forward(const void* gi, int is_bf16, const void* w_hh_bf16, const float* b_hh, int64_t B, int64_t T,
                     int64_t H, void* y, float* saved, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (H != kH) return fail(kErrUnsupported, "seld_gru_forward: built for hidden size 256 (config.py:45)");
  if (B <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_forward: B and T must be positive");
  if (!gi || !w_hh_bf16 || !b_hh || !y) return fail(kErrInvalidArgument, "seld_gru_forward: null pointer");
  GruFwdArgs a{gi, static_cast<const __hip_bfloat16*>(w_hh_bf16), b_hh, y, saved, B, T};
  const dim3 grid(static_cast<unsigned>((B + kRows - 1) / kRows), 2);
  const size_t lds = kWnBytes + 2 * kRows * kHPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (is_bf16) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(gru_forward_kernel<__hip_bfloat16>, grid, dim3(kGruThreads), lds, stream, a);
  } else {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(gru_forward_kernel<float>, grid, dim3(kGruThreads), lds, stream, a);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gru_backward(const void* dy, const void* y, const float* saved, int is_bf16, const void* w_hh_t_bf16,
                      int64_t B, int64_t T, int64_t H, void* dg, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (H != kH) return fail(kErrUnsupported, "seld_gru_backward: built for hidden size 256 (config.py:45)");
  if (B <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_backward: B and T must be positive");
  if (!dy || !y || !saved || !w_hh_t_bf16 || !dg) return fail(kErrInvalidArgument, "seld_gru_backward: null pointer");
  GruBwdArgs a{dy, y, saved, static_cast<const __hip_bfloat16*>(w_hh_t_bf16), dg, B, T};
  const dim3 grid(static_cast<unsigned>((B + kRows - 1) / kRows), 2);
  const size_t lds = kWnBytes + kRows * kDghPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (is_bf16) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(gru_backward_kernel<__hip_bfloat16>, grid, dim3(kGruThreads), lds, stream, a);
  } else {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL(gru_backward_kernel<float>, grid, dim3(kGruThreads), lds, stream, a);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
