// gfx950 log-mel kernel: 8 independent 64-lane wavefront pipelines per workgroup sharing read-only LDS tables.
// Algorithm and lane mapping: logmel_core.h.  Replaces dataset.py:27-58 (reference).
#include "seld_common.h"

namespace seld {

constexpr int kMainWaves = 8;    // main kernel: 512 threads, two wavefronts per SIMD (<= 256 VGPRs, no spills)
constexpr int kEdgeWaves = 4;    // edge kernel: 256 threads
constexpr int kMainLdsBytes = (kTabFloats + kMainWaves * kLdsFloatsPerWave) * 4;   // 152576 B: one workgroup per CU
constexpr int kEdgeLdsBytes = (kTabFloats + kEdgeWaves * kLdsFloatsPerWave) * 4;

struct LogmelArgs {
  const void* pcm;      // [rows][L], rows = N*C
  float* out;
  long rows, C, L, F;   // F = 1 + L/480 frames
  long iters_per_row;   // ceil(F / 4): one iteration = 4 consecutive frames of one row
  long interior;        // iterations 1..interior of every row touch only samples inside [0, L) and store 4 frames
  long edge_per_row;    // the others (first / last of a row): iters_per_row - interior
  long chunk;           // consecutive interior iterations per wavefront (main kernel)
  long sN, sC, sM, sT;  // output strides (elements): clip, channel, mel band, frame
  LogmelTables tab;
  float* spec;          // kSpec 1: the un-packed spectra as well, [rows][F][481] complex64; kSpec 2: the Q15 phasors,
                        // [rows][F][kPhasorPitch] words (logmel_core.h phase_c_store_phasors); else unused
  long main_blocks;     // main kernel: workgroups [0, main_blocks) run interior pipelines, the rest edge iterations
};

__device__ __forceinline__ float* out_pointer(const LogmelArgs& a, long row, long itr, int lane) {
  const long n = row / a.C;
  const long c = row - n * a.C;
  return a.out + n * a.sN + c * a.sC + lane * a.sM + itr * kFramesPerIter * a.sT;
}

__device__ __forceinline__ void fill_tables(const LogmelArgs& a, float* tab, int tid, int nthreads) {
  for (int e = tid; e < kTabFloats; e += nthreads) tab[e] = table_value(a.tab, e);
}

// Value of `v` in the lane below (0 in lane 0): DPP wave_shr:1 -- the B_j hand-off of the mel phase without LDS.
__device__ __forceinline__ float lane_below(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}

#define SELD_WAVE_SYNC()                                   \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

// One edge iteration (index e of rows * edge_per_row) by one wavefront: generic load path (reflection, frames past the end
// masked), no pipelining.
template <typename T, int kSpec>
__device__ __forceinline__ void edge_iteration(const LogmelArgs& a, long e, int lane, const float* tab, float* lds, int seg,
                                               float* const (&pp)[16], const LaneConsts& consts) {
  const int h = lane >> 5;
  const long row = e / a.edge_per_row;
  const long j = e - row * a.edge_per_row;
  const long itr = a.interior >= 1 ? (j == 0 ? 0 : a.interior + j) : j;
  const long tf = itr * kFramesPerIter;
  const T* pcm = static_cast<const T*>(a.pcm);

  float s[48];
  load_samples<T, false>(lane, pcm + row * a.L, a.L, tf + 2 * h, s);
  phase_a(lane, s, consts, lds);
  SELD_WAVE_SYNC();
  cf z[kN2];
  phase_b(lane, lds, z);
  SELD_WAVE_SYNC();
  phase_b_store(lane, lds, z);
  SELD_WAVE_SYNC();
  cf m[16];
  phase_c_load(lane, lds, m);
  if (kSpec == 1) {
    const long fa = tf + 2 * h;
    float* base = a.spec + (row * a.F + fa) * (2 * kBins);
    phase_c_spectrum(lane, z, m, fa < a.F ? base : nullptr, fa + 1 < a.F ? base + 2 * kBins : nullptr);
  }
  if (kSpec == 2) {
    const long fa = tf + 2 * h;
    unsigned* base = reinterpret_cast<unsigned*>(a.spec) + (row * a.F + tf) * kPhasorPitch;
    phase_c_store_phasors(lane, pp, z, m, base, static_cast<unsigned>(2 * h * kPhasorPitch + (lane & 31)), fa < a.F,
                          fa + 1 < a.F);
  } else {
    phase_c_store(lane, pp, z, m);
  }
  SELD_WAVE_SYNC();
  LaneAcc acc;
  phase_d_accumulate(lane, lds, tab, seg, acc);
  float below[kFramesPerIter], db[kFramesPerIter];
#pragma unroll
  for (int f = 0; f < kFramesPerIter; ++f) below[f] = lane_below(acc.ab[f].y);
  phase_d_finish(acc, below, db);
  float* outp = out_pointer(a, row, itr, lane);
#pragma unroll
  for (int f = 0; f < kFramesPerIter; ++f)
    if (tf + f < a.F) outp[f * a.sT] = db[f];
}

// ---- Main kernel: the interior iterations of every row (all but the first and the last one or two).
// One wavefront = one independent pipeline over a contiguous run of iterations; the 8 wavefronts of a
// workgroup only share the read-only LDS tables (one barrier, at start-up).  The body has NO divergent or
// conditional memory operation: the compiler counts the outstanding prefetch loads / output stores exactly,
// so nothing ever drains the memory queue (no vmcnt(0)), and the samples of the next iteration are requested
// right after stage B -- into the registers the window multiply has just freed -- so their HBM latency is
// covered by the un-packing / mel / store phases and by the SIMD's other wavefront.
// kSpec: the spectra the power rows are formed from are also written out (frame-major rows of 481 complex) -- the spatial
// features (csrc/spatial.hip) read them, and a second pass over the PCM through stft_kernel is not needed.
template <typename T, int kSpec>
__global__ __launch_bounds__(kMainWaves * 64, 2) void logmel_main_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5;
  float* tab = smem;
  float* lds = smem + kTabFloats + wave * kLdsFloatsPerWave;
  fill_tables(a, tab, tid, kMainWaves * 64);
  // Zero the wavefront's tile once: pad cells are otherwise never written and the mel phase multiplies
  // over-read cells by a zero weight (0 * NaN would poison the sum).
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;
  const int seg = a.tab.mel_pos[lane];
  float* pp[16];
  power_row_pointers(lane, lds, a.tab.mel_pos, pp);
  __syncthreads();
  LaneConsts consts;
  load_lane_consts(lane, tab, consts);

  // Workgroups past main_blocks run the edge iterations (the first and the last one or two of every row), 8 per
  // workgroup: they land on the CUs the interior pipelines leave free and are done long before those are (round 1 ran
  // them as a second kernel on a side stream; its workgroups could not share a CU's LDS with a main one, ran when the
  // main kernel drained, and the fork / join events and the second launch were 13 % of the feature phase).
  if (static_cast<long>(blockIdx.x) >= a.main_blocks) {
    const long e = (static_cast<long>(blockIdx.x) - a.main_blocks) * kMainWaves + wave;
    if (e < a.rows * a.edge_per_row) edge_iteration<T, kSpec>(a, e, lane, tab, lds, seg, pp, consts);
    return;
  }
  // Which iterations this wavefront runs is the same for its 64 lanes: with the wavefront index read through
  // readfirstlane the whole bookkeeping (row, iteration, clip / channel of the row, the division that starts it) lives in
  // scalar registers and the scalar unit; derived from threadIdx it was ~120 vector instructions of 64-bit division per
  // iteration (out_pointer's row / C).
  const long total = a.rows * a.interior;
  const long gw = static_cast<long>(blockIdx.x) * kMainWaves + __builtin_amdgcn_readfirstlane(wave);
  const long begin = gw * a.chunk;
  const long end = begin + a.chunk < total ? begin + a.chunk : total;
  if (begin >= end) return;
  const T* pcm = static_cast<const T*>(a.pcm);

  long row = begin / a.interior;
  long itr = 1 + (begin - row * a.interior);
  long clip = row / a.C;
  long chan = row - clip * a.C;
  float* const out_lane = a.out + lane * a.sM;
  float s[48];
  load_samples<T, true>(lane, pcm + row * a.L, a.L, itr * kFramesPerIter + 2 * h, s);

#pragma unroll 1
  for (long it = begin; it < end; ++it) {
    long nrow = row, nitr = itr, nclip = clip, nchan = chan;
    if (it + 1 < end) {                       // scalar bookkeeping only; the fetch below is unconditional
      ++nitr;
      if (nitr > a.interior) {
        nitr = 1;
        ++nrow;
        if (++nchan == a.C) { nchan = 0; ++nclip; }
      }
    }
    float* outp = out_lane + (clip * a.sN + chan * a.sC + itr * kFramesPerIter * a.sT);
    phase_a(lane, s, consts, lds);
    SELD_WAVE_SYNC();
    cf z[kN2];
    phase_b(lane, lds, z);
    SELD_WAVE_SYNC();
    phase_b_store(lane, lds, z);
    SELD_WAVE_SYNC();
    // the next iteration's samples: requested here, a whole un-pack + mel + store phase ahead -- except in the fp32
    // phasor instantiation, whose un-packing needs the 48 registers (requested after it: no scratch, 64 B per lane before)
    constexpr bool kLatePrefetch = kSpec != 0 && sizeof(T) == 4;
    if (!kLatePrefetch) load_samples<T, true>(lane, pcm + nrow * a.L, a.L, nitr * kFramesPerIter + 2 * h, s);
    cf m[16];
    phase_c_load(lane, lds, m);
    if (kSpec == 1) {                                        // interior iterations: all four frames exist
      float* base = a.spec + ((row * a.F + itr * kFramesPerIter) * kBins) * 2 + h * (2 * 2 * kBins);
      phase_c_spectrum(lane, z, m, base, base + 2 * kBins);
    }
    if (kSpec == 2) {
      unsigned* base = reinterpret_cast<unsigned*>(a.spec) + (row * a.F + itr * kFramesPerIter) * kPhasorPitch;   // scalar
      phase_c_store_phasors(lane, pp, z, m, base, static_cast<unsigned>(2 * h * kPhasorPitch + (lane & 31)), true, true);
    } else {
      phase_c_store(lane, pp, z, m);
    }
    if (kLatePrefetch) load_samples<T, true>(lane, pcm + nrow * a.L, a.L, nitr * kFramesPerIter + 2 * h, s);
    SELD_WAVE_SYNC();
    LaneAcc acc;
    phase_d_accumulate(lane, lds, tab, seg, acc);
    float below[kFramesPerIter], db[kFramesPerIter];
#pragma unroll
    for (int f = 0; f < kFramesPerIter; ++f) below[f] = lane_below(acc.ab[f].y);
    phase_d_finish(acc, below, db);
#pragma unroll
    for (int f = 0; f < kFramesPerIter; ++f) outp[f * a.sT] = db[f];
    SELD_WAVE_SYNC();
    row = nrow;
    itr = nitr;
    clip = nclip;
    chan = nchan;
  }
}

// ---- Edge kernel: the first iteration of every row (reflection on the left) and the last one or two
// (reflection on the right, frames past the end masked).  One wavefront per edge iteration, no pipelining:
// rows * edge_per_row iterations in total (256 for 32 one-minute clips) against ~96 000 interior ones.
template <typename T, int kSpec>
__global__ __launch_bounds__(kEdgeWaves * 64, 1) void logmel_edge_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  float* tab = smem;
  float* lds = smem + kTabFloats + wave * kLdsFloatsPerWave;
  fill_tables(a, tab, tid, kEdgeWaves * 64);
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;
  const int seg = a.tab.mel_pos[lane];
  float* pp[16];
  power_row_pointers(lane, lds, a.tab.mel_pos, pp);
  __syncthreads();
  LaneConsts consts;
  load_lane_consts(lane, tab, consts);

  const long e = static_cast<long>(blockIdx.x) * kEdgeWaves + wave;
  if (e >= a.rows * a.edge_per_row) return;
  edge_iteration<T, kSpec>(a, e, lane, tab, lds, seg, pp, consts);
}

// ---- STFT export (north-star addition A14: the reference only has the STFT implicitly inside torchaudio).
// Same stages A-C as the log-mel kernels; the un-packed complex spectra are written frame-major,
// out[row][t][k] (complex64, 481 bins per frame).  One wavefront per run of iterations, generic load path
// (reflection capable) for every iteration: this is an auxiliary API (it feeds the intensity-vector and
// GCC-PHAT kernels), not the training hot loop.
template <typename T>
__global__ __launch_bounds__(kEdgeWaves * 64, 2) void stft_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5;
  float* tab = smem;
  float* lds = smem + kTabFloats + wave * kLdsFloatsPerWave;
  fill_tables(a, tab, tid, kEdgeWaves * 64);
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;
  __syncthreads();
  LaneConsts consts;
  load_lane_consts(lane, tab, consts);

  const long total = a.rows * a.iters_per_row;
  const long gw = static_cast<long>(blockIdx.x) * kEdgeWaves + wave;
  const long begin = gw * a.chunk;
  const long end = begin + a.chunk < total ? begin + a.chunk : total;
  const T* pcm = static_cast<const T*>(a.pcm);
#pragma unroll 1
  for (long it = begin; it < end; ++it) {
    const long row = it / a.iters_per_row;
    const long itr = it - row * a.iters_per_row;
    const long tf = itr * kFramesPerIter;
    const bool interior = itr >= 1 && itr <= a.interior;
    float s[48];
    if (interior) load_samples<T, true>(lane, pcm + row * a.L, a.L, tf + 2 * h, s);
    else load_samples<T, false>(lane, pcm + row * a.L, a.L, tf + 2 * h, s);
    phase_a(lane, s, consts, lds);
    SELD_WAVE_SYNC();
    cf z[kN2];
    phase_b(lane, lds, z);
    SELD_WAVE_SYNC();
    phase_b_store(lane, lds, z);
    SELD_WAVE_SYNC();
    cf m[16];
    phase_c_load(lane, lds, m);
    const long fa = tf + 2 * h;
    float* base = a.out + (row * a.F + fa) * (2 * kBins);
    phase_c_spectrum(lane, z, m, fa < a.F ? base : nullptr, fa + 1 < a.F ? base + 2 * kBins : nullptr);
    SELD_WAVE_SYNC();
  }
}

// ---- Fused FOA feature pass (north-star additions A15: log-mel + intensity vectors, no spectra in HBM).
// One 4-wavefront workgroup = the channels W, X, Y, Z of one clip; it walks a contiguous run of that clip's iterations
// (edge iterations included: the generic load path, chosen by a workgroup-uniform branch).  Per iteration: stages A-C per
// channel as in the log-mel kernel -> W publishes its spectrum -> barrier -> X / Y / Z form the intensities of their bins
// (logmel_core.h iv_compute) -> every channel's own log-mel bands -> barrier -> the intensities replace the power rows and go
// through the same sparse mel pass.  LDS: tables 25.6 KB + 4 tiles 63.5 KB + W's spectrum 16 KB = 105 KB, one workgroup per
// CU; the next iteration's samples are requested (clamped indices: never out of bounds, also past the run's end) before
// the un-packing so that their latency is covered although a SIMD holds one wavefront.
constexpr int kIvLdsBytes = (kTabFloats + kIvChannels * kLdsFloatsPerWave + kIvSpecFloats) * 4;
constexpr float kIvEps = 1e-8f;

template <typename T>
__device__ __forceinline__ void iv_fetch(const LogmelArgs& a, long clip, long itr, int wave, int lane, float (&s)[48]) {
  const T* row = static_cast<const T*>(a.pcm) + (clip * kIvChannels + wave) * a.L;
  const long fa = itr * kFramesPerIter + 2 * (lane >> 5);
  if (itr >= 1 && itr <= a.interior) load_samples<T, true>(lane, row, a.L, fa, s);
  else load_samples<T, false>(lane, row, a.L, fa, s);
}

template <typename T>
__global__ __launch_bounds__(kIvChannels * 64, 1) void logmel_iv_kernel(LogmelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float* tab = smem;
  float* lds = smem + kTabFloats + wave * kLdsFloatsPerWave;
  float* wspec = smem + kTabFloats + kIvChannels * kLdsFloatsPerWave;
  fill_tables(a, tab, tid, kIvChannels * 64);
  for (int i = lane; i < kLdsFloatsPerWave; i += 64) lds[i] = 0.0f;
  const int seg = a.tab.mel_pos[lane];
  float* pp[16];
  power_row_pointers(lane, lds, a.tab.mel_pos, pp);
  __syncthreads();
  LaneConsts consts;
  load_lane_consts(lane, tab, consts);

  const long total = (a.rows / kIvChannels) * a.iters_per_row;       // (clip, iteration) pairs
  const long begin = static_cast<long>(blockIdx.x) * a.chunk;
  const long end = begin + a.chunk < total ? begin + a.chunk : total;
  if (begin >= end) return;                                           // workgroup-uniform
  long clip = begin / a.iters_per_row;
  long itr = begin - clip * a.iters_per_row;
  float s[48];
  iv_fetch<T>(a, clip, itr, wave, lane, s);

#pragma unroll 1
  for (long it = begin; it < end; ++it) {
    long nclip = clip, nitr = itr;
    if (it + 1 < end) {
      if (++nitr == a.iters_per_row) { nitr = 0; ++nclip; }
    }
    const long tf = itr * kFramesPerIter;
    phase_a(lane, s, consts, lds);
    SELD_WAVE_SYNC();
    cf z[kN2];
    phase_b(lane, lds, z);
    SELD_WAVE_SYNC();
    phase_b_store(lane, lds, z);
    SELD_WAVE_SYNC();
    iv_fetch<T>(a, nclip, nitr, wave, lane, s);                       // the next iteration's samples (this one's again at the end)
    cf m[16];
    phase_c_load(lane, lds, m);
    cf xa[16], xb[16];
    phase_c_unpack(lane, pp, z, m, xa, xb);
    if (wave == 0) iv_publish(lane, wspec, xa, xb);
    __syncthreads();
    float ia[16], ib[16];
    if (wave != 0) iv_compute(lane, wave, wspec, pp, xa, xb, kIvEps, ia, ib);
    float* outp = a.out + clip * a.sN + lane * a.sM + tf * a.sT;
    {
      LaneAcc acc;
      phase_d_accumulate(lane, lds, tab, seg, acc);
      float below[kFramesPerIter], db[kFramesPerIter];
#pragma unroll
      for (int f = 0; f < kFramesPerIter; ++f) below[f] = lane_below(acc.ab[f].y);
      phase_d_finish(acc, below, db);
#pragma unroll
      for (int f = 0; f < kFramesPerIter; ++f)
        if (tf + f < a.F) outp[wave * a.sC + f * a.sT] = db[f];
    }
    __syncthreads();                                                  // every channel is done with every power row
    if (wave != 0) {
      iv_store_rows(lane, pp, ia, ib);
      SELD_WAVE_SYNC();
      LaneAcc acc;
      phase_d_accumulate(lane, lds, tab, seg, acc);
#pragma unroll
      for (int f = 0; f < kFramesPerIter; ++f) {
        const float v = acc.ab[f].x + lane_below(acc.ab[f].y);
        if (tf + f < a.F) outp[(kIvChannels - 1 + wave) * a.sC + f * a.sT] = v;
      }
    }
    SELD_WAVE_SYNC();
    clip = nclip;
    itr = nitr;
  }
}

template <typename T>
static int launch_logmel_iv(const T* pcm, int64_t N, int64_t L, float* out, const int64_t* strides, hipStream_t stream) {
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!pcm || !out) return fail(kErrInvalidArgument, "seld_logmel_iv: null pointer");
  if (N <= 0) return fail(kErrInvalidArgument, "seld_logmel_iv: N must be positive");
  if (L <= kNfft / 2) return fail(kErrInvalidArgument, "seld_logmel_iv: reflect padding needs L > n_fft/2 = 480 samples");
  if (L >= (1L << 30)) return fail(kErrUnsupported, "seld_logmel_iv: at most 2^30 samples per channel");
  LogmelArgs a;
  a.pcm = pcm;
  a.out = out;
  a.rows = N * kIvChannels;
  a.C = kIvChannels;
  a.L = L;
  a.F = 1 + L / kHop;
  a.iters_per_row = (a.F + kFramesPerIter - 1) / kFramesPerIter;
  a.interior = (L / kHop - kFramesPerIter) / kFramesPerIter;          // as launch_logmel
  if (a.interior < 0) a.interior = 0;
  if (a.interior > a.iters_per_row - 1) a.interior = a.iters_per_row - 1;
  a.edge_per_row = a.iters_per_row - a.interior;
  a.sN = strides[0];
  a.sC = strides[1];
  a.sM = strides[2];
  a.sT = strides[3];
  a.tab = st->tables();
  a.spec = nullptr;
  a.main_blocks = 0;
  const long total = N * a.iters_per_row;
  long groups = total < st->num_cus ? total : static_cast<long>(st->num_cus);
  a.chunk = (total + groups - 1) / groups;
  groups = (total + a.chunk - 1) / a.chunk;
  const unsigned bit = sizeof(T) == 4 ? kAttrLogmelIvF32 : kAttrLogmelIvI16;
  if (need_lds(st, bit)) {                 // once per device (seld_common.h)
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_iv_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kIvLdsBytes));
    lds_attr_set(st, bit);
  }
  hipLaunchKernelGGL(logmel_iv_kernel<T>, dim3(static_cast<unsigned>(groups)), dim3(kIvChannels * 64), kIvLdsBytes, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

template <typename T>
static int launch_stft(const T* pcm, int64_t N, int64_t C, int64_t L, float* out, hipStream_t stream) {
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!pcm || !out) return fail(kErrInvalidArgument, "seld_stft: null pointer");
  if (N <= 0 || C <= 0) return fail(kErrInvalidArgument, "seld_stft: N and C must be positive");
  if (L <= kNfft / 2) return fail(kErrInvalidArgument, "seld_stft: reflect padding needs L > n_fft/2 = 480 samples");
  if (L >= (1L << 30)) return fail(kErrUnsupported, "seld_stft: at most 2^30 samples per channel");
  LogmelArgs a;
  a.pcm = pcm;
  a.out = out;
  a.rows = N * C;
  a.C = C;
  a.L = L;
  a.F = 1 + L / kHop;
  a.iters_per_row = (a.F + kFramesPerIter - 1) / kFramesPerIter;
  a.interior = (L / kHop - kFramesPerIter) / kFramesPerIter;
  if (a.interior < 0) a.interior = 0;
  if (a.interior > a.iters_per_row - 1) a.interior = a.iters_per_row - 1;
  a.edge_per_row = a.iters_per_row - a.interior;
  a.sN = a.sC = a.sM = a.sT = 0;
  a.tab = st->tables();
  const long total = a.rows * a.iters_per_row;
  const long max_waves = static_cast<long>(st->num_cus) * 8;
  long waves = (total + 3) / 4;
  if (waves > max_waves) waves = max_waves;
  a.chunk = (total + waves - 1) / waves;
  const long used = (total + a.chunk - 1) / a.chunk;
  const unsigned bit = sizeof(T) == 4 ? kAttrStftF32 : kAttrStftI16;
  if (need_lds(st, bit)) {                 // once per device (seld_common.h)
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(stft_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeLdsBytes));
    lds_attr_set(st, bit);
  }
  hipLaunchKernelGGL(stft_kernel<T>, dim3(static_cast<unsigned>((used + kEdgeWaves - 1) / kEdgeWaves)),
                     dim3(kEdgeWaves * 64), kEdgeLdsBytes, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

template <typename T, int kSpec = 0>
static int launch_logmel(const T* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout,
                         hipStream_t stream, const int64_t* strides = nullptr, float* spec = nullptr) {
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!pcm || !out || (kSpec && !spec)) return fail(kErrInvalidArgument, "seld_logmel: null pointer");
  if (N <= 0 || C <= 0) return fail(kErrInvalidArgument, "seld_logmel: N and C must be positive");
  if (L <= kNfft / 2)
    return fail(kErrInvalidArgument, "seld_logmel: reflect padding needs L > n_fft/2 = 480 samples");
  if (L >= (1L << 30)) return fail(kErrUnsupported, "seld_logmel: at most 2^30 samples per channel");
  if (!strides && layout != 0 && layout != 1) return fail(kErrInvalidArgument, "seld_logmel: layout must be 0 or 1");

  LogmelArgs a;
  a.pcm = pcm;
  a.out = out;
  a.rows = N * C;
  a.C = C;
  a.L = L;
  a.F = 1 + L / kHop;
  a.iters_per_row = (a.F + kFramesPerIter - 1) / kFramesPerIter;
  // iteration i (frames 4i..4i+3) is interior iff i >= 1 and 480*(4i+4) <= L
  a.interior = (L / kHop - kFramesPerIter) / kFramesPerIter;
  if (a.interior < 0) a.interior = 0;
  if (a.interior > a.iters_per_row - 1) a.interior = a.iters_per_row - 1;
  a.edge_per_row = a.iters_per_row - a.interior;
  if (strides) {                // caller-defined placement (e.g. channels 0..C-1 of a wider feature tensor)
    a.sN = strides[0];
    a.sC = strides[1];
    a.sM = strides[2];
    a.sT = strides[3];
  } else if (layout == 0) {     // [N, C, 64, F]  (reference layout, dataset.py:53)
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
  a.chunk = 1;
  a.spec = spec;

  const unsigned bit = kSpec == 2 ? (sizeof(T) == 4 ? kAttrLogmelPhasorF32 : kAttrLogmelPhasorI16)
                       : kSpec == 1 ? (sizeof(T) == 4 ? kAttrLogmelSpecF32 : kAttrLogmelSpecI16)
                                    : (sizeof(T) == 4 ? kAttrLogmelF32 : kAttrLogmelI16);
  if (need_lds(st, bit)) {                 // once per device (seld_common.h)
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_main_kernel<T, kSpec>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMainLdsBytes));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_edge_kernel<T, kSpec>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kEdgeLdsBytes));
    lds_attr_set(st, bit);
  }
  const long n_edge = a.rows * a.edge_per_row;
  const long total = a.rows * a.interior;
  a.main_blocks = 0;
  if (total == 0) {             // clips shorter than ~8 frames: edge iterations only
    hipLaunchKernelGGL((logmel_edge_kernel<T, kSpec>), dim3(static_cast<unsigned>((n_edge + kEdgeWaves - 1) / kEdgeWaves)),
                       dim3(kEdgeWaves * 64), kEdgeLdsBytes, stream, a);
    SELD_HIP_TRY(hipGetLastError());
    return kOk;
  }
  // ONE launch: one 8-wavefront workgroup per CU runs the interior pipelines (LDS bound; every wavefront gets a
  // contiguous run of >= 8 iterations), and the edge iterations are the trailing workgroups of the same grid.  A few CUs
  // are left to them so that they start at once and finish inside the main workgroups' lifetime.
  const long edge_blocks = (n_edge + kMainWaves - 1) / kMainWaves;
  long reserve = (edge_blocks + 7) / 8;                 // an edge workgroup lives ~1/25 of a full-length main one
  if (reserve > 4) reserve = 4;
  long cus = static_cast<long>(st->num_cus) - reserve;
  if (cus < 1) cus = 1;
  const long max_waves = cus * kMainWaves;
  long waves = (total + 7) / 8;
  if (waves > max_waves) waves = max_waves;
  a.chunk = (total + waves - 1) / waves;
  const long used_waves = (total + a.chunk - 1) / a.chunk;
  a.main_blocks = (used_waves + kMainWaves - 1) / kMainWaves;
  hipLaunchKernelGGL((logmel_main_kernel<T, kSpec>), dim3(static_cast<unsigned>(a.main_blocks + edge_blocks)),
                     dim3(kMainWaves * 64), kMainLdsBytes, stream, a);
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

int seld_logmel_f32_strided(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<float>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides);
}

int seld_logmel_i16_strided(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<int16_t>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides);
}

int seld_logmel_spectrum_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                             int64_t sM, int64_t sT, float* spec_complex, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<float, 1>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides, spec_complex);
}

int seld_logmel_spectrum_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                             int64_t sM, int64_t sT, float* spec_complex, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<int16_t, 1>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides, spec_complex);
}

int seld_logmel_phasors_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, uint32_t* phasors_q15, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<float, 2>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides,
                                       reinterpret_cast<float*>(phasors_q15));
}

int seld_logmel_phasors_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int64_t sN, int64_t sC,
                            int64_t sM, int64_t sT, uint32_t* phasors_q15, void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel<int16_t, 2>(pcm, N, C, L, out, 0, static_cast<hipStream_t>(stream), strides,
                                         reinterpret_cast<float*>(phasors_q15));
}

int seld_logmel_iv_f32(const float* pcm, int64_t N, int64_t L, float* out, int64_t sN, int64_t sC, int64_t sM, int64_t sT,
                       void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel_iv<float>(pcm, N, L, out, strides, static_cast<hipStream_t>(stream));
}

int seld_logmel_iv_i16(const int16_t* pcm, int64_t N, int64_t L, float* out, int64_t sN, int64_t sC, int64_t sM, int64_t sT,
                       void* stream) {
  const int64_t strides[4] = {sN, sC, sM, sT};
  return seld::launch_logmel_iv<int16_t>(pcm, N, L, out, strides, static_cast<hipStream_t>(stream));
}

int64_t seld_phasor_pitch(void) { return seld::kPhasorPitch; }

int seld_stft_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out_complex, void* stream) {
  return seld::launch_stft<float>(pcm, N, C, L, out_complex, static_cast<hipStream_t>(stream));
}

int seld_stft_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out_complex, void* stream) {
  return seld::launch_stft<int16_t>(pcm, N, C, L, out_complex, static_cast<hipStream_t>(stream));
}

int64_t seld_num_frames(int64_t L) { return 1 + L / seld::kHop; }

}  // extern "C"
