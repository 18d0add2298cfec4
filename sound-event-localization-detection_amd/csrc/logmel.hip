// gfx950 log-mel kernel: 8 independent 64-lane wavefront pipelines per workgroup sharing read-only LDS tables.
// Algorithm and lane mapping: logmel_core.h.  Replaces dataset.py:27-58 (reference).
#include "seld_common.h"

namespace seld {

constexpr int kWavesPerWg = 4;                       // one wavefront per SIMD: 512 VGPRs each, nothing ever spills
constexpr int kLogmelLdsBytes = (kTabFloats + kWavesPerWg * kLdsFloatsPerWave) * 4;   // 93184 B (one WG per CU)

struct LogmelArgs {
  const void* pcm;      // [rows][L], rows = N*C
  float* out;
  long rows, C, L, F;   // F = 1 + L/480 frames
  long iters_per_row;   // ceil(F / 4): one iteration = 4 consecutive frames of one row
  long total_iters;     // rows * iters_per_row
  long chunk;           // consecutive iterations per wavefront
  long sN, sC, sM, sT;  // output strides (elements): clip, channel, mel band, frame
  LogmelTables tab;
};

struct IterCtx {
  long L, F, C;
  long sN, sC, sM, sT;
  long row, itr;        // this iteration: (clip*C + channel), 4-frame index inside the row
  long nrow, nitr;      // the next one (prefetch target)
  bool have_next;
};

__device__ __forceinline__ bool iter_is_interior(long it_in_row, long L) {
  // every 48-sample column of both half-wavefronts inside [0, L): no reflection, no clamping
  const long tf = it_in_row * kFramesPerIter;
  return (tf >= 1) && (static_cast<long>(kHop) * (tf + kFramesPerIter) <= L);
}

// One iteration = 4 frames of one row.  The samples of the NEXT iteration are requested at the top of this
// one and consumed at the top of the next, so a wavefront never waits on HBM.  The kernel runs ONE wavefront
// per SIMD (512 VGPRs): current samples, prefetched samples, DFT data and temporaries all stay in registers.
// kFast: the next iteration exists and is interior and THIS one stores all 4 frames: no divergent or
// conditional memory operation, so the compiler counts outstanding loads / stores exactly (no vmcnt(0)).
// Edge iterations (first / last of a row, last of the run) take the generic body, which ends with an
// explicit drain so both paths meet at the loop head in a known state.
template <typename T, bool kFast>
__device__ __forceinline__ void logmel_iteration(const IterCtx& x, const T* pcm, float* out, const float* tab,
                                                 float* lds, int lane, int b0, float (&s_cur)[48], float (&s_next)[48]) {
  const int h = lane >> 5;
  const long tf = x.itr * kFramesPerIter;
  const long n = x.row / x.C;
  const long c = x.row - n * x.C;
  float* outp = out + n * x.sN + c * x.sC + lane * x.sM + tf * x.sT;

  // ---- prefetch the next iteration's samples first: a whole iteration of cover for their HBM latency
  if (kFast) {
    load_samples<T, true>(lane, pcm + x.nrow * x.L, x.L, x.nitr * kFramesPerIter + 2 * h, s_next);
  } else if (x.have_next) {
    const T* rowp = pcm + x.nrow * x.L;
    const long fa = x.nitr * kFramesPerIter + 2 * h;
    if (iter_is_interior(x.nitr, x.L)) load_samples<T, true>(lane, rowp, x.L, fa, s_next);
    else load_samples<T, false>(lane, rowp, x.L, fa, s_next);
  }

  phase_a(lane, s_cur, tab, lds);
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
  phase_d_accumulate(lane, lds, tab, b0, acc);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float db[kFramesPerIter];
  phase_d_finish(lane, lds, acc, db);
  if (kFast) {
#pragma unroll
    for (int s = 0; s < kFramesPerIter; ++s) outp[s * x.sT] = db[s];
  } else {
#pragma unroll
    for (int s = 0; s < kFramesPerIter; ++s)
      if (tf + s < x.F) outp[s * x.sT] = db[s];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (!kFast) __builtin_amdgcn_s_waitcnt(0);         // vmcnt(0) lgkmcnt(0): known state at the join
#pragma unroll
  for (int m = 0; m < 48; ++m) s_cur[m] = s_next[m];
}

// One wavefront = one independent pipeline over a contiguous run of iterations; the 8 wavefronts of a
// workgroup only share the read-only LDS tables (one barrier, at start-up).
template <typename T>
__global__ __launch_bounds__(kWavesPerWg * 64, 1) void logmel_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  float* tab = smem;
  float* lds = smem + kTabFloats + wave * kLdsFloatsPerWave;

  for (int e = tid; e < kTabFloats; e += kWavesPerWg * 64) tab[e] = table_value(a.tab, e);
  // Zero the wavefront's tile once: pad cells are otherwise never written and the mel phase multiplies
  // over-read cells by a zero weight (0 * NaN would poison the sum).
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;
  const int b0 = a.tab.mel_b0[lane];
  __syncthreads();

  const long gw = static_cast<long>(blockIdx.x) * kWavesPerWg + wave;
  const long begin = gw * a.chunk;
  const long end = begin + a.chunk < a.total_iters ? begin + a.chunk : a.total_iters;
  if (begin >= end) return;
  const T* pcm = static_cast<const T*>(a.pcm);

  IterCtx x;
  x.L = a.L; x.F = a.F; x.C = a.C;
  x.sN = a.sN; x.sC = a.sC; x.sM = a.sM; x.sT = a.sT;
  x.row = begin / a.iters_per_row;
  x.itr = begin - x.row * a.iters_per_row;

  float s_cur[48], s_next[48];
  {
    const T* rowp = pcm + x.row * a.L;
    const long fa = x.itr * kFramesPerIter + 2 * (lane >> 5);
    if (iter_is_interior(x.itr, a.L)) load_samples<T, true>(lane, rowp, a.L, fa, s_cur);
    else load_samples<T, false>(lane, rowp, a.L, fa, s_cur);
  }

#pragma unroll 1
  for (long it = begin; it < end; ++it) {
    x.nrow = x.row;
    x.nitr = x.itr + 1;
    if (x.nitr == a.iters_per_row) { x.nitr = 0; ++x.nrow; }
    x.have_next = it + 1 < end;
    const bool fast = x.have_next && iter_is_interior(x.nitr, a.L) && (x.itr * kFramesPerIter + kFramesPerIter <= a.F);
    if (fast) logmel_iteration<T, true>(x, pcm, a.out, tab, lds, lane, b0, s_cur, s_next);
    else logmel_iteration<T, false>(x, pcm, a.out, tab, lds, lane, b0, s_cur, s_next);
    x.row = x.nrow;
    x.itr = x.nitr;
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
  if (L >= (1L << 30)) return fail(kErrUnsupported, "seld_logmel: at most 2^30 samples per channel");
  if (layout != 0 && layout != 1) return fail(kErrInvalidArgument, "seld_logmel: layout must be 0 or 1");

  LogmelArgs a;
  a.pcm = pcm;
  a.out = out;
  a.rows = N * C;
  a.C = C;
  a.L = L;
  a.F = 1 + L / kHop;
  a.iters_per_row = (a.F + kFramesPerIter - 1) / kFramesPerIter;
  a.total_iters = a.rows * a.iters_per_row;
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
  // one 4-wavefront workgroup per CU (one wavefront per SIMD); every wavefront gets a contiguous run of >= 16 iterations
  const long max_waves = static_cast<long>(st->num_cus) * kWavesPerWg;
  long waves = (a.total_iters + 15) / 16;
  if (waves > max_waves) waves = max_waves;
  a.chunk = (a.total_iters + waves - 1) / waves;
  const long used_waves = (a.total_iters + a.chunk - 1) / a.chunk;
  const long grid = (used_waves + kWavesPerWg - 1) / kWavesPerWg;
  static bool attr_done[2] = {false, false};
  const int which = sizeof(T) == 4 ? 0 : 1;
  if (!attr_done[which]) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLogmelLdsBytes));
    attr_done[which] = true;
  }
  hipLaunchKernelGGL(logmel_kernel<T>, dim3(static_cast<unsigned>(grid)), dim3(kWavesPerWg * 64), kLogmelLdsBytes,
                     stream, a);
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
