// Spatial features on top of the STFT export (north-star additions A15 / A16: the reference repository has
// NO intensity-vector or GCC-PHAT code -- SURVEY.md F4 -- so the definitions are the DCASE SELD-baseline ones,
// stated here and in DESIGN.md section 7, and the oracle is this project's own CPU restatement):
//
//   FOA intensity vectors (4-ch first-order ambisonics, channel 0 = W):
//       I_c[k] = Re(conj(W[k]) X_c[k]) / (eps + |W[k]|^2 + (|X_1|^2 + |X_2|^2 + |X_3|^2)[k] / 3),  c = 1..3, eps = 1e-8
//       iv[c][m] = sum_k fb[k][m] I_c[k]            (the same HTK filterbank as the log-mel features)
//
//   GCC-PHAT (C-channel microphone array, all C(C-1)/2 pairs m < n in lexicographic order):
//       R[k] = conj(X_m[k]) X_n[k] ;  cc = irfft(R / |R|, 960)  (1 where either channel is silent, |X|^2 <= 1e-12) ;
//       gcc[pair][j] = cc[j - 32 mod 960], j = 0..63
//       i.e. the 64 lags -32..31 around zero delay (np.concatenate((cc[-32:], cc[:32]))).
//
// HBM-bound: 4 x 481 x 8 B = 15.4 KB of spectra read and 768 B written per frame (intensity vectors);
// C x 3.8 KB read and C(C-1)/2 x 256 B written per frame (GCC).
#include "seld_common.h"

namespace seld {

constexpr int kIvPitch = 512;                       // floats per intensity row in LDS (481 used, rest zero)
constexpr int kIvLdsFloatsPerWave = 3 * kIvPitch + 3 * 64;
constexpr int kIvWaves = 4;

struct IvArgs {
  const float* spec;     // [N][4][F][481] complex64
  float* out;
  long N, F;
  long sN, sC, sM, sT;   // output strides (elements) of (clip, iv channel 0..2, mel band, frame)
  float eps;
  LogmelTables tab;
};

__global__ __launch_bounds__(kIvWaves * 64) void foa_iv_kernel(IvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* rows = smem + wave * kIvLdsFloatsPerWave;           // [3][kIvPitch]
  float* bs = rows + 3 * kIvPitch;                           // [3][64]
  for (int i = lane; i < 3 * kIvPitch; i += 64) rows[i] = 0.0f;
  const int b0 = a.tab.mel_b0[lane];
  float wd[kMelMaxCnt], wu[kMelMaxCnt];
#pragma unroll
  for (int i = 0; i < kMelMaxCnt; ++i) {
    wd[i] = a.tab.mel_wd[i * 64 + lane];
    wu[i] = a.tab.mel_wu[i * 64 + lane];
  }
  const long frames_total = a.N * a.F;
  for (long f = static_cast<long>(blockIdx.x) * kIvWaves + wave; f < frames_total;
       f += static_cast<long>(gridDim.x) * kIvWaves) {
    const long n = f / a.F;
    const long t = f - n * a.F;
    const float2* w_row = reinterpret_cast<const float2*>(a.spec) + ((n * 4 + 0) * a.F + t) * kBins;
    const long ch_stride = a.F * kBins;                      // complex elements between channels
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = lane + 64 * r;
      if (k < kBins) {
        const float2 w = w_row[k];
        const float2 x1 = w_row[ch_stride + k], x2 = w_row[2 * ch_stride + k], x3 = w_row[3 * ch_stride + k];
        const float e = a.eps + (w.x * w.x + w.y * w.y) +
                        ((x1.x * x1.x + x1.y * x1.y) + (x2.x * x2.x + x2.y * x2.y) + (x3.x * x3.x + x3.y * x3.y)) * (1.0f / 3.0f);
        const float inv = 1.0f / e;
        rows[k] = (w.x * x1.x + w.y * x1.y) * inv;           // Re(conj(W) X) = wr xr + wi xi
        rows[kIvPitch + k] = (w.x * x2.x + w.y * x2.y) * inv;
        rows[2 * kIvPitch + k] = (w.x * x3.x + w.y * x3.y) * inv;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float acc_a[3], acc_b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* p = rows + c * kIvPitch + b0;
      float sa = 0.0f, sb = 0.0f;
#pragma unroll
      for (int i = 0; i < kMelMaxCnt; ++i) {
        const float v = p[i];
        sa = fmaf(wd[i], v, sa);
        sb = fmaf(wu[i], v, sb);
      }
      acc_a[c] = sa;
      acc_b[c] = sb;
      bs[c * 64 + lane] = sb;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* outp = a.out + n * a.sN + lane * a.sM + t * a.sT;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float below = lane > 0 ? bs[c * 64 + lane - 1] : 0.0f;
      outp[c * a.sC] = acc_a[c] + below;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------- GCC-PHAT
// One 7-wavefront workgroup per frame: the frame's C spectra (<= 8 x 481 complex, 30.8 KB) are staged in LDS
// once and every half-wavefront runs ONE packed inverse transform that yields the cross-correlations of TWO
// microphone pairs (pair a -> real part, pair b -> imaginary part):
//     Zhat = Ra + i Rb  (Ra, Rb Hermitian)   =>   ifft(Zhat) = ra + i rb,    ifft(Z) = conj(fft(conj(Z))) / N
// so the forward 960 = 32 x 30 machinery of the log-mel kernel is reused unchanged for stage A (the "time"
// index is now the frequency bin k = 30 n1 + n2, the phase transform R/|R| is computed on the fly from the
// LDS-resident spectra), and stage B is pruned to the two output rows that hold the wanted lags:
//     lag = k1 + 32 k2 :  k2 = 0 -> lags 0..31,   k2 = 29 -> lags 928..959 = -32..-1.
constexpr int kGccWaves = 7;
constexpr int kGccSpecFloats = 8 * kBins * 2;                      // 7696 floats
constexpr int kGccTwFloats = 5 * 64 * 4;                           // two-level twiddle quads
constexpr int kGccLdsFloats = kGccSpecFloats + kGccWaves * kEFloats;     // 142 KB: one workgroup per CU

__constant__ unsigned char kPairM[28] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6};
__constant__ unsigned char kPairN[28] = {1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};

struct GccArgs {
  const float* spec;     // [N][C][F][481] complex64
  float* out;            // out[n*sN + pair*sC + j*sM + t*sT], j = 0..63 (lag j - 32)
  long N, C, F;
  long sN, sC, sM, sT;
  LogmelTables tab;
};

// pair index -> (m, n) for C channels, lexicographic (m < n)
__device__ __forceinline__ void pair_channels(int p, int C, int& m, int& n) {
  if (C == 8) {
    m = kPairM[p];
    n = kPairN[p];
    return;
  }
  m = 0;
  int rem = p;
  while (rem >= C - 1 - m) {
    rem -= C - 1 - m;
    ++m;
  }
  n = m + 1 + rem;
}

// ---- Round 2 structure.  Round 1 staged the RAW spectra channel by channel (eight dependent HBM round trips per
// frame with nothing else to do: ~16 of the 18 us a frame took) and evaluated R / |R| -- two products, a reciprocal square
// root and three selects -- for every (pair, bin) twice, 56 x 960 times per frame.  Now
//   * the phase transform factorises:  R / |R| = conj(Um) Un  with the per-CHANNEL phasors  U = X / |X|,  so the
//     normalisation is done once per (channel, bin) -- 8 x 481 per frame -- when the frame is written to LDS, and the
//     per-pair work is one complex product (two packed instructions);
//   * the next frame's spectra are requested into registers (9 x 8 bytes per lane) BEFORE the current frame is
//     transformed and are normalised / stored after it: the HBM latency is covered by the 14 packed transforms;
//   * a silent bin (|X|^2 <= kGccSilencePower) must give R / |R| = 1: such a channel's phasor is
//     stored as 0, the workgroup learns from the barrier (__syncthreads_or) whether the frame has one, and only then
//     runs the variant that turns zero products into 1.
constexpr int kGccThreads = kGccWaves * 64;
// |X|^2 at or below this is a SILENT bin (phase factor 1): below any recording's noise floor, above the ~1e-7 of its
// neighbour's amplitude that the packed two-frame transform leaves in a frame of digital silence (oracle/features.py)
constexpr float kGccSilencePower = 1e-12f;
constexpr int kGccPre = (8 * kBins + kGccThreads - 1) / kGccThreads;          // complex elements per lane and frame: 9

// conj(a) * b as two packed operations
__device__ __forceinline__ cf conj_a_times_b(cf a, cf b) {
  return cf_fma(cf_make(b.y, b.x), cf_make(a.y, -a.y), cf_scale(b, a.x));       // (ax bx + ay by, ax by - ay bx)
}
template <bool kHasZero>
__device__ __forceinline__ void gcc_transform_pairs(const GccArgs& a, const cf* u, const cf (&twc)[10], float* lds, int lane,
                                                    int q, int n_pairs, int n_packed, long n, long t) {
  const int h = lane >> 5, l = lane & 31;
  const int n2 = l < kN2 ? l : kN2 - 1;
  const bool live = q < n_packed;
  const int pa = live ? 2 * q : 0;
  const int pb = (live && 2 * q + 1 < n_pairs) ? 2 * q + 1 : pa;
  int ma, na, mb, nb;
  pair_channels(pa, static_cast<int>(a.C), ma, na);
  pair_channels(pb, static_cast<int>(a.C), mb, nb);
  const cf* uma = u + ma * kBins;
  const cf* una = u + na * kBins;
  const cf* umb = u + mb * kBins;
  const cf* unb = u + nb * kBins;
  // ---- stage A input: z[k] = conj(Ra[k] + i Rb[k]) at k = 30 n1 + n2 (Hermitian extension above bin 480), with
  //   k <= 480: Ra + i Rb = (ar - bi) + i (ai + br)  ->  z = conj(Ra) - swap(Rb)
  //   k  > 480: conj(Ra) + i conj(Rb) at 960 - k       ->  z = Ra + (bi, -br)
  cf z[kN1];
#pragma unroll
  for (int n1 = 0; n1 < kN1; ++n1) {
    const int k = kN2 * n1 + n2;
    const bool upper = k > 480;                       // n1 <= 15: never, n1 >= 17: always (a compile-time fact per n1)
    const int kk = upper ? kNfft - k : k;
    cf ra = conj_a_times_b(uma[kk], una[kk]);          // Ra / |Ra| = conj(Um) Un
    cf rb = conj_a_times_b(umb[kk], unb[kk]);
    if (kHasZero) {
      if (ra.x == 0.0f && ra.y == 0.0f) ra = cf_make(1.0f, 0.0f);
      if (rb.x == 0.0f && rb.y == 0.0f) rb = cf_make(1.0f, 0.0f);
    }
    const cf lower_z = cf_fma(cf_make(rb.y, rb.x), cf_make(-1.0f, -1.0f), cf_make(ra.x, -ra.y));
    const cf upper_z = cf_fma(cf_make(rb.y, rb.x), cf_make(1.0f, -1.0f), ra);
    if (n1 <= 15) z[n1] = lower_z;
    else if (n1 >= 17) z[n1] = upper_z;
    else z[n1] = upper ? upper_z : lower_z;
  }
  dft32(z);
  stage_a_finish(lane, z, twc, lds);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- stage B pruned to k2 = 0 and k2 = 29: S0 = sum_n2 E ,  S29 = sum_n2 E conj(W_30^{n2})
  const cf* e = reinterpret_cast<const cf*>(lds + e_index(h, l, 0));
  cf s0 = cf_make(0.0f, 0.0f), s29 = cf_make(0.0f, 0.0f);
#pragma unroll
  for (int j = 0; j < kN2; ++j) {
    const cf ev = e[j];
    const float c = __builtin_cosf(6.283185307179586f * j / 30.0f);
    const float sn = __builtin_sinf(6.283185307179586f * j / 30.0f);
    s0 = cf_add(s0, ev);
    // E * conj(W) with W = exp(-2 pi i j / 30) = c - i sn  ->  (er c - ei sn, ei c + er sn)
    s29 = cf_fma_splat(ev, c, s29);
    s29 = cf_fma_swap(ev, -sn, sn, s29);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // x = conj(S) / 960 :  pair a = Re x = Sr / 960 ,  pair b = Im x = -Si / 960
  if (live) {
    const float sc = 1.0f / 960.0f;
    float* oa = a.out + n * a.sN + pa * a.sC + t * a.sT;
    oa[(32 + l) * a.sM] = s0.x * sc;                            // lag l
    oa[l * a.sM] = s29.x * sc;                                   // lag l - 32
    if (pb != pa) {
      float* ob = a.out + n * a.sN + pb * a.sC + t * a.sT;
      ob[(32 + l) * a.sM] = -s0.y * sc;
      ob[l * a.sM] = -s29.y * sc;
    }
  }
}

__global__ __launch_bounds__(kGccThreads, 2) void gcc_phat_kernel(GccArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int h = lane >> 5;
  cf* u = reinterpret_cast<cf*>(smem);                             // [C][481] phasors of the frame in flight
  float* lds = smem + kGccSpecFloats + wave * kEFloats;
  cf twc[10];                                                      // this lane's two-level twiddles (LaneConsts order)
#pragma unroll
  for (int v = 0; v < 5; ++v) {
    twc[2 * v] = cf_make(table_value(a.tab, kTabTw + v * 256 + lane * 4), table_value(a.tab, kTabTw + v * 256 + lane * 4 + 1));
    twc[2 * v + 1] = cf_make(table_value(a.tab, kTabTw + v * 256 + lane * 4 + 2), table_value(a.tab, kTabTw + v * 256 + lane * 4 + 3));
  }
  const int n_pairs = static_cast<int>(a.C * (a.C - 1) / 2);
  const int n_packed = (n_pairs + 1) / 2;
  const int n_elems = static_cast<int>(a.C) * kBins;               // complex elements of one frame
  const long total = a.N * a.F;
  const long ch_stride = a.F * kBins;                              // complex elements between channels

  // element e = tid + kGccThreads j of a frame is bin e % 481 of channel e / 481 (recomputed per frame: a few integer
  // instructions against 21 x 8 bytes in flight; held in registers the offsets made the kernel spill)
  const int ch_stride_i = static_cast<int>(ch_stride);
  auto src_off = [&](int j) {
    const int e = tid + kGccThreads * j;
    const int c = e < n_elems ? e / kBins : 0;
    return c * ch_stride_i + (e < n_elems ? e - c * kBins : 0);
  };
  const float2* spec2 = reinterpret_cast<const float2*>(a.spec);
  auto frame_base = [&](long f) {
    const long n = f / a.F;
    return spec2 + (n * a.C * a.F + (f - n * a.F)) * kBins;
  };
  float2 pre[kGccPre];
  long f = blockIdx.x;
  if (f < total) {
    const float2* src = frame_base(f);
#pragma unroll
    for (int j = 0; j < kGccPre; ++j) pre[j] = src[src_off(j)];
  }
  for (; f < total; f += gridDim.x) {
    const long n = f / a.F;
    const long t = f - n * a.F;
    __syncthreads();                                               // the previous frame's readers are done
    int zero = 0;
#pragma unroll
    for (int j = 0; j < kGccPre; ++j) {
      const int e = tid + kGccThreads * j;
      const float mag2 = pre[j].x * pre[j].x + pre[j].y * pre[j].y;
      const bool sounding = mag2 > kGccSilencePower;
      const float inv = sounding ? rsqrtf(mag2) : 0.0f;
      zero |= sounding ? 0 : (e < n_elems ? 1 : 0);
      if (e < n_elems) u[e] = cf_make(pre[j].x * inv, pre[j].y * inv);
    }
    const int has_zero = __syncthreads_or(zero);
    // The next frame of this workgroup, requested AFTER the barrier (its fence waits for every outstanding load: issued
    // before it the loads would be waited for on the spot) and pinned ahead of the transforms that cover their latency.
    // Clamped, not conditional: the last requests re-read this frame.
    __builtin_amdgcn_sched_barrier(0);
    {
      const long fn = f + gridDim.x < total ? f + gridDim.x : f;
      const float2* src = frame_base(fn);
#pragma unroll
      for (int j = 0; j < kGccPre; ++j) pre[j] = src[src_off(j)];
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int q0 = 2 * wave; q0 < n_packed; q0 += 2 * kGccWaves) {  // wave-uniform trip count
      if (has_zero) gcc_transform_pairs<true>(a, u, twc, lds, lane, q0 + h, n_pairs, n_packed, n, t);
      else gcc_transform_pairs<false>(a, u, twc, lds, lane, q0 + h, n_pairs, n_packed, n, t);
    }
  }
}

}  // namespace seld

extern "C" {

int seld_foa_intensity(const float* spec_complex, int64_t N, int64_t F, float* out, int64_t sN, int64_t sC,
                       int64_t sM, int64_t sT, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!spec_complex || !out) return fail(kErrInvalidArgument, "seld_foa_intensity: null pointer");
  if (N <= 0 || F <= 0) return fail(kErrInvalidArgument, "seld_foa_intensity: N and F must be positive");
  IvArgs a{spec_complex, out, N, F, sN, sC, sM, sT, 1e-8f, st->tables()};
  long blocks = (N * F + kIvWaves - 1) / kIvWaves;
  const long cap = static_cast<long>(st->num_cus) * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(foa_iv_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kIvWaves * 64),
                     kIvWaves * kIvLdsFloatsPerWave * sizeof(float), static_cast<hipStream_t>(stream_), a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gcc_phat(const float* spec_complex, int64_t N, int64_t C, int64_t F, float* out, int64_t sN, int64_t sC,
                  int64_t sM, int64_t sT, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!spec_complex || !out) return fail(kErrInvalidArgument, "seld_gcc_phat: null pointer");
  if (N <= 0 || F <= 0) return fail(kErrInvalidArgument, "seld_gcc_phat: N and F must be positive");
  if (C < 2 || C > 8) return fail(kErrUnsupported, "seld_gcc_phat: 2..8 channels");
  GccArgs a{spec_complex, out, N, C, F, sN, sC, sM, sT, st->tables()};
  long blocks = N * F;
  const long cap = static_cast<long>(st->num_cus) * 4;
  if (blocks > cap) blocks = cap;
  const size_t lds = kGccLdsFloats * sizeof(float);
  if (need_lds(st, kAttrSpatial)) {        // once per device (seld_common.h)
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gcc_phat_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    lds_attr_set(st, kAttrSpatial);
  }
  hipLaunchKernelGGL(gcc_phat_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kGccWaves * 64), lds,
                     static_cast<hipStream_t>(stream_), a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
