// gfx950 log-mel kernel: one 64-lane wavefront per workgroup, no inter-wave traffic.
// Algorithm and lane mapping: logmel_core.h.  Replaces dataset.py:27-58 (reference).
#include "seld_common.h"

namespace seld {

struct LogmelArgs {
  const void* pcm;      // [rows][L], rows = N*C
  float* out;
  long rows, C, L, F;   // F = 1 + L/480 frames
  long groups;          // ceil(F / 16) frame groups per row
  long items;           // rows * groups
  long chunk;           // consecutive items per wavefront
  long sN, sC, sM, sT;  // output strides (elements): clip, channel, mel band, frame
  LogmelTables tab;
};

template <typename T>
__global__ __launch_bounds__(64, 2) void logmel_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;

  // Zero the whole tile once: pad cells are otherwise never written and the mel phase
  // multiplies over-read cells by a zero weight (0 * NaN would poison the sum).
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;

  LaneConst k;
  load_lane_const(lane, a.tab, k);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const long begin = static_cast<long>(blockIdx.x) * a.chunk;
  const long end = begin + a.chunk < a.items ? begin + a.chunk : a.items;
  const T* pcm = static_cast<const T*>(a.pcm);
  const int h = lane >> 5;

  for (long item = begin; item < end; ++item) {
    const long row = item / a.groups;
    const long g = item - row * a.groups;
    const long t0 = g * kFramesPerGroup;
    const T* rowp = pcm + row * a.L;
    const long n = row / a.C;
    const long c = row - n * a.C;
    float* outp = a.out + n * a.sN + c * a.sC + lane * a.sM;
    // all 48-sample columns of all 4 iterations inside [0, L): no reflection, no clamping
    const bool interior = (t0 >= 1) && (static_cast<long>(kHop) * (t0 + kFramesPerGroup) <= a.L);

#pragma unroll 1
    for (int it = 0; it < kItersPerGroup; ++it) {
      const long tf = t0 + it * kFramesPerIter;       // first frame of this iteration
      if (tf >= a.F) break;
      const long fa = tf + 2 * h;
      if (interior) {
        phase_a<T, true>(lane, rowp, a.L, fa, k, lds);
      } else {
        phase_a<T, false>(lane, rowp, a.L, fa, k, lds);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      float zr[kN2], zi[kN2];
      phase_b(lane, lds, zr, zi);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      phase_b_store(lane, lds, zr, zi);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      float mr[16], mi[16];
      phase_c_load(lane, lds, mr, mi);
      phase_c_store(lane, lds, zr, zi, mr, mi);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      LaneAcc acc;
      phase_d_accumulate(lane, lds, a.tab, k, acc);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float db[kFramesPerIter];
      phase_d_finish(lane, lds, acc, db);
#pragma unroll
      for (int s = 0; s < kFramesPerIter; ++s) {
        const long t = tf + s;
        if (t < a.F) outp[t * a.sT] = db[s];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename T>
static int launch_logmel(const T* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout,
                         hipStream_t stream) {
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!pcm || !out) return fail(kErrInvalidArgument, "seld_logmel: null pointer");
  if (N <= 0 || C <= 0) return fail(kErrInvalidArgument, "seld_logmel: N and C must be positive");
  if (L <= kNfft / 2)
    return fail(kErrInvalidArgument, "seld_logmel: reflect padding needs L > n_fft/2 = 480 samples");
  if (layout != 0 && layout != 1) return fail(kErrInvalidArgument, "seld_logmel: layout must be 0 or 1");

  LogmelArgs a;
  a.pcm = pcm;
  a.out = out;
  a.rows = N * C;
  a.C = C;
  a.L = L;
  a.F = 1 + L / kHop;
  a.groups = (a.F + kFramesPerGroup - 1) / kFramesPerGroup;
  a.items = a.rows * a.groups;
  if (layout == 0) {            // [N, C, 64, F]  (reference layout, dataset.py:53)
    a.sT = 1;
    a.sM = a.F;
    a.sC = kMels * a.F;
    a.sN = C * kMels * a.F;
  } else {                      // [N, F, C, 64]  (time-major: what the windows slice)
    a.sM = 1;
    a.sC = kMels;
    a.sT = C * kMels;
    a.sN = a.F * C * kMels;
  }
  a.tab = st->tables();
  const long max_waves = static_cast<long>(st->num_cus) * 8;   // LDS/VGPR budget: 8 wavefronts per CU
  const long waves = a.items < max_waves ? a.items : max_waves;
  a.chunk = (a.items + waves - 1) / waves;
  const long grid = (a.items + a.chunk - 1) / a.chunk;
  const size_t lds_bytes = kLdsFloatsPerWave * sizeof(float);
  hipLaunchKernelGGL(logmel_kernel<T>, dim3(static_cast<unsigned>(grid)), dim3(64), lds_bytes, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // namespace seld

extern "C" {

int seld_logmel_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, void* stream) {
  return seld::launch_logmel<float>(pcm, N, C, L, out, layout, static_cast<hipStream_t>(stream));
}

int seld_logmel_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, void* stream) {
  return seld::launch_logmel<int16_t>(pcm, N, C, L, out, layout, static_cast<hipStream_t>(stream));
}

int64_t seld_num_frames(int64_t L) { return 1 + L / seld::kHop; }

}  // extern "C"
