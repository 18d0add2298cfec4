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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <type_traits>

#include "seld_common.h"

extern "C" int seld_gcc_table_host(uint16_t* table);      // include/seld_hip.h (defined below)

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
constexpr float kGccSilencePower = kSilencePower;      // logmel_core.h (the Q15 phasors apply the same rule at the source)
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
    // The next frame of this workgroup, requested here and pinned ahead of the transforms that cover its latency.
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


// ---------------------------------------------------------------------------------------- GCC-PHAT on the matrix cores
// The 64 wanted lags of a pair are a SMALL dense transform of its 481 phase factors:
//     cc[n] = (1/960) sum_k w_k ( Re R_k cos(2 pi k n / 960) - Im R_k sin(2 pi k n / 960) ),   w_0 = w_480 = 1, else 2
// (irfft), so with  C[l] = sum_k Re R_k  w_k cos(theta_k l)  and  S[l] = sum_k Im R_k (-w_k sin(theta_k l)),  l = 0..32:
//     cc[+l] = (C[l] + S[l]) / 960 ,    cc[-l] = (C[l] - S[l]) / 960 .
// Per frame that is two GEMMs  [pairs x 481] x [481 x 33]  -- M = pairs (28 -> two 16-row tiles), N = lags (33 -> three
// 16-column tiles), K = bins (481 -> 16 steps of 32) on v_mfma_f32_16x16x32_f16: the A fragments (Re R, Im R of 8
// consecutive bins of one pair per lane) are formed in registers from the frame's fp32 phasors in LDS and rounded to fp16
// once; the B fragments are a constant cosine / sine table (96 KB of fp16, fragment-major, staged in LDS once per
// workgroup); accumulation in fp32.  Rounding: |R|, |cos| <= 1 in fp16 (2^-11 relative) over 2 x 481 terms of random sign
// gives ~1e-5 rms on outputs of O(0.03..1) (measured against the float64 oracle in tests/test_spatial_gpu.py: <= 1e-4).
// The FFT kernel above needed ~7 k vector instructions per frame for the same 28 x 64 numbers.
//
// One 4-wavefront workgroup keeps TWO frames in flight (wavefront = frame slot x pair tile); the phasors of a frame are
// normalised and stored by the two wavefronts of its slot, the next two frames are requested into registers right after
// the barrier and arrive while the products run.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGmWaves = 4;
constexpr int kGmThreads = kGmWaves * 64;
constexpr int kGmUFloats = 61 * 8 * 16;                           // one frame: 61 bin groups x 8 channels x 8 bins x (re, im) = 31 232 B
constexpr int kGmKSteps = 16;                                     // 512 bins / 32
constexpr int kGmLagTiles = 3;                                    // lags 0..47 (0..32 used)
constexpr int kGmTableFrags = 2 * kGmLagTiles * kGmKSteps;        // cos | sin: 96 fragments of 1 KB
constexpr int kGmTableBytes = kGmTableFrags * 64 * 16;            // 98 304 B
constexpr int kGmLdsBytes = kGmTableBytes + 2 * kGmUFloats * 4 + 16;   // 160 784 B (+ the two silent-bin flags)

struct GccMfmaArgs {
  const float* spec;     // [N][C][F][481] complex64; the kQ15 instantiation: [N][C][F][kPhasorPitch] Q15 phasors (re | im << 16)
  float* out;            // out[n*sN + pair*sC + j*sM + t*sT], j = 0..63 (lag j - 32)
  long N, C, F;
  long sN, sC, sM, sT;
  const void* table;     // kGmTableBytes of fp16 B fragments (build_gcc_table)
};

// Value of element j (0..7) of lane `lane` of fragment (part, tile, kstep) of the table: T[n][k] with k = 32 kstep +
// 8 (lane >> 4) + j the bin and n = 16 tile + (lane & 15) the lag (the A- and the B-operand fragment of the 16x16x32
// MFMA hold the same (index, k) per lane, so the table serves either role; the kernel uses it as A).
inline double gcc_table_value(int part, int tile, int kstep, int lane, int j) {
  const int k = 32 * kstep + 8 * (lane >> 4) + j;
  const int lag = 16 * tile + (lane & 15);
  if (k > 480 || lag > 32) return 0.0;
  const double w = (k == 0 || k == 480) ? 1.0 : 2.0;
  const double ang = 2.0 * M_PI * static_cast<double>((static_cast<long>(k) * lag) % kNfft) / kNfft;
  return part == 0 ? w * cos(ang) : -w * sin(ang);
}

// LDS layout of a frame's phasors: CHUNK q = 8 (bin / 8) + channel holds 8 bins of one channel as four float4 --
// logical h = 0, 1: Re of bins 0..3, 4..7; h = 2, 3: Im -- (planar: the products below are then element-wise float4
// arithmetic, i.e. packed instructions, and a pair of neighbouring bins converts to fp16 with one v_cvt_pk), the four
// float4 ROTATED by rot(q) = (q >> 2) & 3: a wavefront's ds_read_b128 is serviced in four groups of 16 lanes that span
// two of the four bin groups g = lane >> 4 and up to eight channels; with the rotation the 16-byte bank slot
// (4 q + ((h + rot) & 3)) mod 16 is distinct for the 16 (channel, g parity) combinations of a group, so the reads are
// conflict free without padding (channel-major rows gave 45 % of all LDS cycles as conflicts, SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int gm_chunk_slot(int q, int h) { return q * 16 + (((h + (q >> 2)) & 3) << 2); }   // float index

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct GmPhasors {            // 8 bins of the two channels of a pair: Re / Im x bins 0..3 / 4..7
  f32x4 mr0, mr1, mi0, mi1, nr0, nr1, ni0, ni1;
};

__device__ __forceinline__ void gm_load(const float* u, const int (&om)[4], const int (&on)[4], int koff, GmPhasors& x) {
  x.mr0 = *reinterpret_cast<const f32x4*>(u + om[0] + koff);
  x.mr1 = *reinterpret_cast<const f32x4*>(u + om[1] + koff);
  x.mi0 = *reinterpret_cast<const f32x4*>(u + om[2] + koff);
  x.mi1 = *reinterpret_cast<const f32x4*>(u + om[3] + koff);
  x.nr0 = *reinterpret_cast<const f32x4*>(u + on[0] + koff);
  x.nr1 = *reinterpret_cast<const f32x4*>(u + on[1] + koff);
  x.ni0 = *reinterpret_cast<const f32x4*>(u + on[2] + koff);
  x.ni1 = *reinterpret_cast<const f32x4*>(u + on[3] + koff);
}

template <bool kHasZero>
__device__ __forceinline__ void gm_fragments(const GmPhasors& x, half8& a_re, half8& a_im) {
  const f32x4 mr0 = x.mr0, mr1 = x.mr1, mi0 = x.mi0, mi1 = x.mi1, nr0 = x.nr0, nr1 = x.nr1, ni0 = x.ni0, ni1 = x.ni1;
  // conj(Um) Un = (mr nr + mi ni) + i (mr ni - mi nr)
  f32x4 re0 = __builtin_elementwise_fma(mi0, ni0, mr0 * nr0), re1 = __builtin_elementwise_fma(mi1, ni1, mr1 * nr1);
  f32x4 im0 = __builtin_elementwise_fma(-mi0, nr0, mr0 * ni0), im1 = __builtin_elementwise_fma(-mi1, nr1, mr1 * ni1);
  if (kHasZero) {                                                   // a silent channel's phasor is 0: factor 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool z0 = (mr0[j] == 0.0f && mi0[j] == 0.0f) || (nr0[j] == 0.0f && ni0[j] == 0.0f);
      const bool z1 = (mr1[j] == 0.0f && mi1[j] == 0.0f) || (nr1[j] == 0.0f && ni1[j] == 0.0f);
      re0[j] = z0 ? 1.0f : re0[j];
      im0[j] = z0 ? 0.0f : im0[j];
      re1[j] = z1 ? 1.0f : re1[j];
      im1[j] = z1 ? 0.0f : im1[j];
    }
  }
  const half4 r0 = __builtin_convertvector(re0, half4), r1 = __builtin_convertvector(re1, half4);
  const half4 i0 = __builtin_convertvector(im0, half4), i1 = __builtin_convertvector(im1, half4);
  a_re = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
  a_im = __builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// cc[+l] = (C[l] + S[l]) / 960 at index 32 + l (l = 0..31),  cc[-l] = (C[l] - S[l]) / 960 at index 32 - l (l = 1..32);
// the lag stride of the output is 1 and rows are 16-byte aligned (the host checks).  A lane holds lags l0 .. l0 + 3
// (l0 = 16 tile + 4 g) of its pair: the positive ones are one aligned 16-byte store at 32 + l0; the negative ones, reversed,
// would start at 29 - l0 -- one float off alignment -- so each lane instead takes the value of lag l0 + 4 from its
// neighbour (lane + 16: the next group of four lags, or for g = 3 group 0's first lag of the next tile) and stores lags
// l0 + 4 .. l0 + 1 at 28 - l0, aligned; that also covers lag -32 (index 0), and index 32 (lag 0) is the positive store's.
// Four aligned stores per lane instead of two aligned, two unaligned and a single float (same speed, measured A/B on one
// box: the stores are a tenth of the frame and issue-bound either way; kept for the fewer memory transactions).
// `scale` = 1 / 960 times whatever factor the caller left out of its phase factors.  Called by whole wavefronts.
__device__ __forceinline__ void gm_store_pairs(const GccMfmaArgs& a, const f32x4 (&acc_c)[kGmLagTiles],
                                               const f32x4 (&acc_s)[kGmLagTiles], int pair, int g, int n_pairs, long n,
                                               long t, float scale = 1.0f / 960.0f) {
  float first[kGmLagTiles];                                          // (C - S) of each tile's first lag, this lane's group
#pragma unroll
  for (int tl = 0; tl < kGmLagTiles; ++tl) first[tl] = (acc_c[tl][0] - acc_s[tl][0]) * scale;
  const int from = ((static_cast<int>(threadIdx.x) + 16) & 63) << 2;   // the lane 16 further (g + 1; g = 3 wraps to g = 0)
  float next[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
    next[tl] = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(g == 0 ? first[tl + 1] : first[tl])));
  if (pair >= n_pairs) return;
  float* op = a.out + n * a.sN + pair * a.sC + t * a.sT;
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    const int l0 = 16 * tl + 4 * g;
    const f32x4 plus = (acc_c[tl] + acc_s[tl]) * scale;
    const f32x4 minus = (acc_c[tl] - acc_s[tl]) * scale;
    *reinterpret_cast<f32x4*>(op + 32 + l0) = plus;
    const f32x4 rev = {next[tl], minus[3], minus[2], minus[1]};      // lags -(l0 + 4) .. -(l0 + 1)
    *reinterpret_cast<f32x4*>(op + 28 - l0) = rev;
  }
}

template <bool kHasZero>
__device__ __forceinline__ void gcc_mfma_frame(const GccMfmaArgs& a, const float* u, const half8* table, int lane, int mt,
                                               int n_pairs, long n, long t) {
  const int r = lane & 15, g = lane >> 4;
  int p = 16 * mt + r;
  if (p >= n_pairs) p = n_pairs - 1;                               // rows past the last pair repeat it (never stored)
  int cm, cn;
  pair_channels(p, static_cast<int>(a.C), cm, cn);
  // float offsets of the four float4 of chunk (bin group g, channel) -- k-step ks adds 4 bin groups = 512 floats (the
  // rotation does not change: 32 chunks further) -- and of bin group 60 (bins 480..487) for the last step's upper groups
  int om[4], on[4], om_last[4], on_last[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    om[h] = gm_chunk_slot(8 * g + cm, h);
    on[h] = gm_chunk_slot(8 * g + cn, h);
    om_last[h] = g == 0 ? om[h] : gm_chunk_slot(cm, h);
    on_last[h] = g == 0 ? on[h] : gm_chunk_slot(cn, h);
  }
  const half8* tab = table + lane;
  f32x4 acc_c[kGmLagTiles], acc_s[kGmLagTiles];
#pragma unroll
  for (int tl = 0; tl < kGmLagTiles; ++tl)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_c[tl][i] = acc_s[tl][i] = 0.0f;
  // The table is the A operand (rows = lags), the phase factors the B operand (columns = pairs): a lane then holds FOUR
  // CONSECUTIVE lags of one pair, D[row = lag 16 tile + 4 g + i][col = pair r], and stores them as one 16-byte vector.
  // Everything a step reads from LDS -- the phasors AND the six table fragments -- is requested one step ahead (one
  // wavefront per SIMD: nothing else hides the LDS latency; read at the point of use a step took ~600 cycles, most of
  // them waiting for the fragments).  The last step reads bin group 60 for every lane: groups 61..63 do not exist and
  // their table rows are zero.
  struct TableFrags { half8 c[kGmLagTiles], s[kGmLagTiles]; };
  auto table_load = [&](int ks, TableFrags& tf) {
#pragma unroll
    for (int tl = 0; tl < kGmLagTiles; ++tl) {
      tf.c[tl] = tab[((0 * kGmLagTiles + tl) * kGmKSteps + ks) * 64];
      tf.s[tl] = tab[((1 * kGmLagTiles + tl) * kGmKSteps + ks) * 64];
    }
  };
  auto products = [&](const TableFrags& tf, const half8& b_re, const half8& b_im) {
#pragma unroll
    for (int tl = 0; tl < kGmLagTiles; ++tl) {
      acc_c[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf.c[tl], b_re, acc_c[tl], 0, 0, 0);
      acc_s[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf.s[tl], b_im, acc_s[tl], 0, 0, 0);
    }
  };
  GmPhasors cur, nxt;
  TableFrags tcur, tnxt;
  gm_load(u, om, on, 0, cur);
  table_load(0, tcur);
#pragma unroll
  for (int ks = 0; ks < kGmKSteps; ++ks) {                   // fully unrolled: every LDS offset is an immediate
    if (ks + 1 < kGmKSteps - 1) gm_load(u, om, on, (ks + 1) * 512, nxt);
    else if (ks + 1 == kGmKSteps - 1) gm_load(u, om_last, on_last, 60 * 8 * 16, nxt);
    if (ks + 1 < kGmKSteps) table_load(ks + 1, tnxt);
    __builtin_amdgcn_sched_barrier(0);                       // (left alone the scheduler sinks them to their use)
    half8 b_re, b_im;
    gm_fragments<kHasZero>(cur, b_re, b_im);
    products(tcur, b_re, b_im);
    cur = nxt;
    tcur = tnxt;
  }
  gm_store_pairs(a, acc_c, acc_s, 16 * mt + r, g, n_pairs, n, t);
}

// Workgroup barrier for data exchanged through LDS: wait for this wavefront's LDS operations, then s_barrier -- what hipcc
// emits for __syncthreads() on this target (checked in the ISA: no vmcnt wait, the requests for the next frame stay in
// flight across it).  Spelled out because the silent-bin flag below replaces __syncthreads_or, which costs two barriers
// and an LDS reduction per call: here it is one LDS word per iteration parity, written by the wavefronts that saw a
// silent bin and read after the barrier that the staging needs anyway.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kGmBinsPerLane = 4;                                  // bins tt, tt + 128, tt + 256, tt + 384 (< 481) of every channel

// kQ15: the input is the log-mel pass's Q15 phasors (logmel_core.h phase_c_store_phasors: X / |X| already taken, 4 B per
// bin -- half the HBM bytes of the complex64 spectra on both sides, and no reciprocal square root here); otherwise raw
// complex64 spectra (seld_gcc_phat on an STFT).
template <bool kQ15>
__global__ __launch_bounds__(kGmThreads, 1) void gcc_mfma_kernel(GccMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int slot = __builtin_amdgcn_readfirstlane(wave >> 1);      // frame slot and pair tile: wavefront-uniform, kept
  const int mt = __builtin_amdgcn_readfirstlane(wave & 1);         // scalar (frame / row arithmetic on the scalar unit)
  half8* table = reinterpret_cast<half8*>(smem_raw);
  float* u = reinterpret_cast<float*>(smem_raw + kGmTableBytes) + slot * kGmUFloats;
  {                                                                // the constant table, once per workgroup
    const uint4* src = static_cast<const uint4*>(a.table);
    uint4* dst = reinterpret_cast<uint4*>(smem_raw);
    for (int i = tid; i < kGmTableBytes / 16; i += kGmThreads) dst[i] = src[i];
    float* uz = reinterpret_cast<float*>(smem_raw + kGmTableBytes);
    for (int i = tid; i < 2 * kGmUFloats; i += kGmThreads) uz[i] = 0.0f;      // bins 481..487 and unused channels stay zero
  }
  const int n_pairs = static_cast<int>(a.C * (a.C - 1) / 2);
  const int n_tiles = (n_pairs + 15) / 16;
  const int n_ch = static_cast<int>(a.C);
  const long total = a.N * a.F;
  constexpr int kRow = kQ15 ? kPhasorPitch : kBins;                // elements (words / complex values) per (channel, frame) row
  typedef typename std::conditional<kQ15, unsigned, float2>::type In;
  const int ch_stride_i = static_cast<int>(a.F * kRow);
  const int tt = tid & 127;                                        // index within the slot's two wavefronts
  const In* spec2 = reinterpret_cast<const In*>(a.spec);
  auto frame_base = [&](long f) {
    const long n = f / a.F;
    return spec2 + (n * a.C * a.F + (f - n * a.F)) * kRow;
  };
  // Lane tt stages bins tt + 128 i (i = 0..3; the last only for tt < 97) of every channel.  Where a bin goes in its chunk
  // (gm_chunk_slot) is loop invariant: channel c adds 16 floats -- an immediate -- and channels 4..7 rotate one float4
  // further than 0..3, so four offsets per bin (Re / Im x channel half) are all the address arithmetic of the frame loop.
  int dst[kGmBinsPerLane][4];
  unsigned src_bin[kGmBinsPerLane];
#pragma unroll
  for (int i = 0; i < kGmBinsPerLane; ++i) {
    const int bin = tt + 128 * i < kBins ? tt + 128 * i : kBins - 1;      // lanes past the last bin repeat it (not stored)
    const int q0 = 8 * (bin >> 3), j = bin & 7;
    dst[i][0] = gm_chunk_slot(q0, j >> 2) + (j & 3);                       // Re, channels 0..3 (+ 16 c)
    dst[i][1] = gm_chunk_slot(q0 + 4, j >> 2) + (j & 3) - 64;              // Re, channels 4..7 (+ 16 c)
    dst[i][2] = gm_chunk_slot(q0, 2 + (j >> 2)) + (j & 3);                 // Im
    dst[i][3] = gm_chunk_slot(q0 + 4, 2 + (j >> 2)) + (j & 3) - 64;
    src_bin[i] = static_cast<unsigned>(bin);
  }
  const bool last_bin_valid = tt + 128 * (kGmBinsPerLane - 1) < kBins;
  auto request = [&](const In* src, In (&pre)[8][kGmBinsPerLane]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned coff = static_cast<unsigned>(c < n_ch ? c : 0) * static_cast<unsigned>(ch_stride_i);   // uniform
#pragma unroll
      for (int i = 0; i < kGmBinsPerLane; ++i) pre[c][i] = src[coff + src_bin[i]];
    }
  };
  const long stride = 2L * gridDim.x;
  long f = 2L * blockIdx.x + slot;                                 // this slot's frame
  // (A second register set of requests, issued two iterations ahead, was tried: no gain -- the loop is not waiting for HBM
  // -- and its 64 registers pushed the products' LDS reads back to their use.)
  In pre_a[8][kGmBinsPerLane];
  auto clamp_frame = [&](long fr) { return fr < total ? fr : (f < total ? f : total - 1); };
  request(frame_base(clamp_frame(f)), pre_a);
  int* flags = reinterpret_cast<int*>(smem_raw + kGmTableBytes + 2 * kGmUFloats * 4);   // silent-bin flag per iteration parity
  if (tid < 2) flags[tid] = 0;
  __syncthreads();                                                 // the table and the zeroed buffers are in place
  int parity = 0;
  auto iteration = [&](In (&pre)[8][kGmBinsPerLane]) {
    lds_barrier();                                                 // the previous frames' readers are done
    if (tid == 0) flags[parity ^ 1] = 0;                           // (read last in the previous iteration)
    int zero = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < n_ch) {                                              // uniform
        float nx[kGmBinsPerLane], ny[kGmBinsPerLane];
#pragma unroll
        for (int i = 0; i < kGmBinsPerLane; ++i) {
          if constexpr (kQ15) {
            const unsigned w = pre[c][i];                          // a silent bin was stored as 0 by the log-mel pass
            zero |= w == 0u ? 1 : 0;                               // (a repeated last bin flags what its owner flags)
            nx[i] = static_cast<float>(static_cast<int>(w << 16) >> 16) * (1.0f / 32767.0f);
            ny[i] = static_cast<float>(static_cast<int>(w) >> 16) * (1.0f / 32767.0f);
          } else {
            const float2 x = pre[c][i];
            const float mag2 = x.x * x.x + x.y * x.y;
            const bool sounding = mag2 > kGccSilencePower;
            const float inv = sounding ? rsqrtf(mag2) : 0.0f;
            zero |= sounding ? 0 : 1;
            nx[i] = x.x * inv;
            ny[i] = x.y * inv;
          }
        }
        float* uc = u + 16 * (c & 3) + 64 * (c >> 2);              // channel c: + 16 c floats
#pragma unroll
        for (int i = 0; i < kGmBinsPerLane - 1; ++i) {
          uc[dst[i][c >> 2]] = nx[i];
          uc[dst[i][2 + (c >> 2)]] = ny[i];
        }
        if (last_bin_valid) {
          uc[dst[kGmBinsPerLane - 1][c >> 2]] = nx[kGmBinsPerLane - 1];
          uc[dst[kGmBinsPerLane - 1][2 + (c >> 2)]] = ny[kGmBinsPerLane - 1];
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(zero != 0) != 0 && lane == 0) flags[parity] = 1;
    __builtin_amdgcn_sched_barrier(0);
    request(frame_base(clamp_frame(f + stride)), pre);             // the slot's next frame: in flight across the products
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    const int has_zero = flags[parity];
    if (f < total && mt < n_tiles) {
      const long n = f / a.F;
      const long t = f - n * a.F;
      if (has_zero) gcc_mfma_frame<true>(a, u, table, lane, mt, n_pairs, n, t);
      else gcc_mfma_frame<false>(a, u, table, lane, mt, n_pairs, n, t);
    }
    f += stride;
    parity ^= 1;
  };
  for (long f0 = 2L * blockIdx.x; f0 < total; f0 += stride) iteration(pre_a);   // uniform trip count for the workgroup
}


// ---------------------------------------------------------------------------------------------------------------------
// The Q15 kernel proper: the same matrix-core transform, fed with the phasor WORDS as they lie in HBM.
//
// gcc_mfma_kernel<true> above unpacks every word to two floats in LDS (31 KB per frame, so two frames per workgroup, two
// wavefronts per frame) and forms conj(Um) Un in fp32; its products phase is bound by what the wavefronts read from LDS
// (14 KB per 32-bin step).  Here the words stay packed:
//   * LDS holds a frame as 8 channel rows of 488 words (15.6 KB): FOUR frames fit beside the 96 KB table, one per
//     wavefront, each wavefront doing both pair tiles of its frame -- so a table fragment read from LDS feeds two MFMAs
//     and a 32-bin step reads 14 KB for twelve MFMAs instead of six;
//   * staging is a copy (global b128 -> registers -> ds_write_b128), and because a frame belongs to ONE wavefront the
//     loop has no workgroup barrier at all: LDS operations of a wavefront complete in order;
//   * conj(Um) Un of two Q15 words is two integer dot products, EXACT in int32: Re = (mr, mi) . (nr, ni),
//     Im = (mr, mi) . (ni, -nr) (v_dot2_i32_i16; the second operand is the first rotated by 16 bits with its upper half
//     negated); int32 -> fp32 -> one rounding to fp16, as before.  The 1 / 32767^2 of the Q15 scale is folded into the
//     conversion to keep the fp16 operand in [-1, 1].
// LDS bank slots: a wavefront's ds_read_b128 is serviced in groups of 16 lanes that hold up to eight channels x two
// neighbouring bin groups (8 bins = 32 B = two 16-byte units).  A channel row is 122 units, i.e. rows start at units
// 10 c mod 16 = all eight EVEN residues; the two units of every ODD bin group are stored swapped (gq_unit_slot), which
// puts the odd groups' first halves on the odd residues: 16 distinct slots, no conflicts, no padding.
constexpr int kGqRowUnits = kPhasorPitch / 4;                     // 16-byte units per (channel, frame) row: 122
constexpr int kGqRowBytes = kGqRowUnits * 16;
constexpr int kGqFrameUnits = 8 * kGqRowUnits;                    // 976
constexpr int kGqFrameBytes = kGqFrameUnits * 16;                 // 15 616
constexpr int kGqUnitsPerLane = (kGqFrameUnits + 63) / 64;        // copy rounds of 64 units: 16 (the last: lanes 0..15)
constexpr int kGqDumpBytes = 64 * 16;                              // where lanes with nothing to copy write (per wavefront: shared, never read)
constexpr int kGqFrames = 4;                                       // frames per workgroup
constexpr int kGqLdsBytes = kGmTableBytes + kGqFrames * kGqFrameBytes + kGqDumpBytes + 2 * kGqFrames * 4;   // 161 824 B
static_assert(kGqLdsBytes <= 160 * 1024, "gcc_q15_kernel: table + four frames must fit the CU's LDS");
static_assert(kPhasorPitch % 8 == 0 && kPhasorPitch >= 8 * 61, "gcc_q15_kernel: rows hold 61 whole bin groups");

// where unit i of a row (bins 4 i .. 4 i + 3) is stored: the halves of odd bin groups (i >> 1) swapped
__device__ __forceinline__ int gq_unit_slot(int i) { return (i & ~1) | ((i ^ (i >> 1)) & 1); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));


struct GqWords {               // 8 bins of the two channels of a pair, one Q15 word (re | im << 16) each
  u32x4 m0, m1, n0, n1;
};

// v_dot2_i32_i16 with a zero addend.  The builtin (__builtin_amdgcn_sdot2) is selected as the two-operand v_dot2c, whose
// accumulator IS the destination: a v_mov 0 per product, a quarter of the instructions of the fragment.  The
// three-operand encoding is written out instead, the eight products of a fragment in ONE statement: the compiler cannot
// see that an asm statement is a dot instruction, so the wait states this target wants between a dot instruction's write
// and another vector instruction that reads (3) or overwrites (4) the register are supplied here -- within the
// statement no product reads another's result, and the s_nop covers the last ones.
__device__ __forceinline__ void gq_dots(const unsigned (&wm)[8], const unsigned (&wn)[8], int (&out)[8]) {
  asm("v_dot2_i32_i16 %0, %8, %16, %24\n\tv_dot2_i32_i16 %1, %9, %17, %24\n\t"
      "v_dot2_i32_i16 %2, %10, %18, %24\n\tv_dot2_i32_i16 %3, %11, %19, %24\n\t"
      "v_dot2_i32_i16 %4, %12, %20, %24\n\tv_dot2_i32_i16 %5, %13, %21, %24\n\t"
      "v_dot2_i32_i16 %6, %14, %22, %24\n\tv_dot2_i32_i16 %7, %15, %23, %24\n\t"
      "s_nop 3"
      : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]),
        "=&v"(out[7])
      : "v"(wm[0]), "v"(wm[1]), "v"(wm[2]), "v"(wm[3]), "v"(wm[4]), "v"(wm[5]), "v"(wm[6]), "v"(wm[7]),
        "v"(wn[0]), "v"(wn[1]), "v"(wn[2]), "v"(wn[3]), "v"(wn[4]), "v"(wn[5]), "v"(wn[6]), "v"(wn[7]),
        "s"(1 << 15));                                              // the addend: rounds the product's upper half
}

// The fp16 MFMA operand is the UPPER HALF of the int32 product (a Q14 number, |.| <= 2^14), rounded by the dot product's
// addend and converted half to half (v_cvt_f16_i16 reading word 1: no shift, no int32 -> fp32 -> scale -> fp16 chain).
// Dropping the lower half costs 2^-15 of full scale per element, a tenth of the fp16 rounding that follows; the factor
// kGqOutputScale restores the scale at the output.
constexpr float kGqOutputScale = 65536.0f / (32767.0f * 32767.0f) / 960.0f;
typedef short i16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gq_fragments(const GqWords& x, bool has_zero, half8& b_re, half8& b_im) {
  unsigned wm[8], wn[8], wq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    wm[j] = j < 4 ? x.m0[j & 3] : x.m1[j & 3];
    wn[j] = j < 4 ? x.n0[j & 3] : x.n1[j & 3];
    // (ni, -nr) = (hi * 1, lo * 0xffff) mod 2^16: one packed multiply whose op_sel swaps the halves of wn (|nr| <= 32767:
    // the negation cannot overflow)
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(wq[j]) : "v"(wn[j]), "s"(0xffff0001u));
  }
  int ri[8], ii[8];
  gq_dots(wm, wn, ri);                                              // Re = (mr, mi) . (nr, ni)
  gq_dots(wm, wq, ii);                                              // Im = (mr, mi) . (ni, -nr)
  if (has_zero) {
    // (wavefront-uniform, rare: the frame has a silent bin somewhere.)  A silent channel's word is 0 and its factor is
    // 1 -- which the Q14 operand cannot carry exactly: 32767^2 / 65536 = 16383.0002 rounds to 16384 in fp16, 6.1e-5 too
    // much at lag 0 of a silent pair.  Such frames take the exact route instead: int32 -> fp32 -> 1 / 32767^2 -> ONE
    // rounding to fp16, operands in [-1, 1], and the output scale 1 / 960 (the caller picks the scale by the same flag).
    f32x4 fr[2], fi[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool z = wm[j] == 0u || wn[j] == 0u;
      const float s = 1.0f / (32767.0f * 32767.0f);
      fr[j >> 2][j & 3] = z ? 1.0f : static_cast<float>(ri[j] - (1 << 15)) * s;     // (without the rounding addend)
      fi[j >> 2][j & 3] = z ? 0.0f : static_cast<float>(ii[j] - (1 << 15)) * s;
    }
    const half4 r0 = __builtin_convertvector(fr[0], half4), r1 = __builtin_convertvector(fr[1], half4);
    const half4 i0 = __builtin_convertvector(fi[0], half4), i1 = __builtin_convertvector(fi[1], half4);
    b_re = __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7);
    b_im = __builtin_shufflevector(i0, i1, 0, 1, 2, 3, 4, 5, 6, 7);
    return;
  }
  // Word 1 of each product (its upper half) -> fp16, two products per register: v_cvt_f16_i16 with SDWA selects, the
  // even bins into the low halves (high halves zeroed), then the odd bins into the high halves (low halves kept): 16
  // instructions for the 16 values.  (Plain C++ gets a conversion per value plus a v_pack_b32_f16 per pair: 24.)  One
  // statement because this target wants a wait state between a partial-register write (dst_sel) and a read of that
  // register -- the second pass reads what the first wrote, eight instructions later, and the s_nop covers the
  // consumer of the last.
  unsigned p[8];
  asm("v_cvt_f16_i16_sdwa %0, %8 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %1, %10 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %2, %12 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %3, %14 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %4, %16 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %5, %18 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %6, %20 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %7, %22 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %0, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %1, %11 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %2, %13 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %3, %15 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %4, %17 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %5, %19 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %6, %21 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "v_cvt_f16_i16_sdwa %7, %23 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
      "s_nop 1"
      : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
      : "v"(ri[0]), "v"(ri[1]), "v"(ri[2]), "v"(ri[3]), "v"(ri[4]), "v"(ri[5]), "v"(ri[6]), "v"(ri[7]),
        "v"(ii[0]), "v"(ii[1]), "v"(ii[2]), "v"(ii[3]), "v"(ii[4]), "v"(ii[5]), "v"(ii[6]), "v"(ii[7]));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t pr = {p[0], p[1], p[2], p[3]}, pi = {p[4], p[5], p[6], p[7]};
  b_re = __builtin_bit_cast(half8, pr);
  b_im = __builtin_bit_cast(half8, pi);
}

// One pair tile of one frame by one wavefront; `u` = the frame in LDS; om / on = byte offsets of the rows of the channels
// of the lane's pair (loop invariant, and pair_channels reads a table from global memory: inside the frame loop its
// vmcnt(0) would also wait for the next frame's requests).
template <typename PerStep>
__device__ __forceinline__ void gcc_q15_tile(const unsigned char* u, const half8* table, int lane, int om, int on,
                                             bool has_zero, f32x4 (&acc_c)[kGmLagTiles], f32x4 (&acc_s)[kGmLagTiles],
                                             PerStep&& per_step) {
  const int g = lane >> 4;
  // the lane's two units (bins 0..3 | 4..7 of bin group 4 ks + g) within a row; step ks adds 128 bytes.  The last step
  // reads bin group 60 for every lane (groups 61..63 do not exist; their table rows are zero).
  const int lo = (2 * g + (g & 1)) * 16, hi = (2 * g + 1 - (g & 1)) * 16;
  const half8* tab = table + lane;
#pragma unroll
  for (int tl = 0; tl < kGmLagTiles; ++tl)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_c[tl][i] = acc_s[tl][i] = 0.0f;
  struct TableFrags { half8 c[kGmLagTiles], s[kGmLagTiles]; };
  auto table_load = [&](int ks, TableFrags& tf) {
#pragma unroll
    for (int tl = 0; tl < kGmLagTiles; ++tl) {
      tf.c[tl] = tab[((0 * kGmLagTiles + tl) * kGmKSteps + ks) * 64];
      tf.s[tl] = tab[((1 * kGmLagTiles + tl) * kGmKSteps + ks) * 64];
    }
  };
  auto words_load = [&](int ks, GqWords& w) {
    const bool last = ks == kGmKSteps - 1;                         // compile-time per unrolled step
    const int l = last ? 120 * 16 : lo + 128 * ks, h = last ? 121 * 16 : hi + 128 * ks;
    w.m0 = *reinterpret_cast<const u32x4*>(u + om + l);
    w.m1 = *reinterpret_cast<const u32x4*>(u + om + h);
    w.n0 = *reinterpret_cast<const u32x4*>(u + on + l);
    w.n1 = *reinterpret_cast<const u32x4*>(u + on + h);
  };
  // The table is the A operand (rows = lags), the phase factors the B operand (columns = pairs): a lane then holds FOUR
  // CONSECUTIVE lags of one pair and stores them as one 16-byte vector.  Single buffers (two wavefronts per SIMD: 256
  // registers each): the next step's words are requested once the fragments are formed, the next step's table
  // fragments once the products are issued; the partner wavefront of the SIMD (the frame's other tile) fills what
  // latency that leaves.
  GqWords w;
  TableFrags tf;
  words_load(0, w);
  table_load(0, tf);
#pragma unroll
  for (int ks = 0; ks < kGmKSteps; ++ks) {                   // fully unrolled: every LDS offset is an immediate
    half8 b_re, b_im;
    gq_fragments(w, has_zero, b_re, b_im);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < kGmKSteps) words_load(ks + 1, w);
    per_step(ks);                                            // (the kernel's loop: one of the next frame's requests)
    __builtin_amdgcn_sched_barrier(0);                       // (left alone the scheduler sinks the reads to their use)
#pragma unroll
    for (int tl = 0; tl < kGmLagTiles; ++tl) {
      acc_c[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf.c[tl], b_re, acc_c[tl], 0, 0, 0);
      acc_s[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tf.s[tl], b_im, acc_s[tl], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < kGmKSteps) table_load(ks + 1, tf);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// kTiles wavefronts per frame, one pair tile each; four frames ("slots") per workgroup: wavefront w takes tile w >> 2 of
// slot w & 3 (the hardware deals a workgroup's wavefronts to the four SIMDs in turn, so a frame's two tiles share a
// SIMD and one's vector work fills the other's matrix-core, LDS-issue and wait cycles: alone on its SIMD a wavefront
// needs 480 cycles per 32-bin step -- the sum of what it issues -- two together 370 each).  The loop alternates two
// intervals separated by workgroup barriers: "products" -- the matrix-core transform of the staged frame, with the next
// frame's requests spread over its steps, then the stores -- and "copy" -- the requested words to LDS.  Running the two
// halves of the workgroup in antiphase (one computes while the other copies) was measured too: 3 % slower, the lone
// wavefront's 480 cycles per step cost more than the hidden copy saves.
template <int kTiles>
__global__ __launch_bounds__(kTiles * kGqFrames * 64) void gcc_q15_kernel(GccMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int kThreads = kTiles * kGqFrames * 64;
  constexpr int kRounds = kGqUnitsPerLane / kTiles;                // copy rounds per wavefront: the frame's tiles share the copy
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tile = wave / kGqFrames, slot = wave & (kGqFrames - 1);
  const half8* table = reinterpret_cast<const half8*>(smem_raw);
  unsigned char* u = smem_raw + kGmTableBytes + slot * kGqFrameBytes;      // this wavefront's frame
  int* flags = reinterpret_cast<int*>(smem_raw + kGmTableBytes + kGqFrames * kGqFrameBytes + kGqDumpBytes);   // [kGqFrames][2]
  {                                                                // the constant table, once per workgroup
    const uint4* src = static_cast<const uint4*>(a.table);
    uint4* dst = reinterpret_cast<uint4*>(smem_raw);
    for (int i = tid; i < kGmTableBytes / 16; i += kThreads) dst[i] = src[i];
    if (tid < 2 * kGqFrames) flags[tid] = 0;
  }
  const int n_pairs = static_cast<int>(a.C * (a.C - 1) / 2);
  const int n_ch = static_cast<int>(a.C);
  // Lane `lane` of tile t copies units lane + 64 (kRounds t + i) of the frame (unit = 16 bytes = 4 bins; row c = units
  // 122 c ..): where from and where to is loop invariant.  Rows of channels the clip does not have are never read by a
  // pair.  Words 481..487 of a row are whatever the log-mel pass left there (it never writes them): they meet zero table
  // rows, and an integer product is finite whatever the words are.
  // Every lane loads and stores in every round -- a lane with nothing to copy reads unit 0 and writes to a dump slot
  // past the frames: a load or a copy under a lane mask leaves the compiler unable to tell whether the request is still
  // outstanding, and it then waits for ALL memory operations (vmcnt(0), the output stores included) wherever it reuses
  // the register.
  const uint4* spec = reinterpret_cast<const uint4*>(a.spec);
  const unsigned ch_stride = static_cast<unsigned>(a.F) * kGqRowUnits;     // units; the host checks C F 488 < 2^31
  unsigned src_off[kRounds];
  int dst_off[kRounds];
  unsigned all_mask = 0, first_mask = 0;                           // per round: all four words are bins | only the first is
  const int dump_off = (kGqFrames - slot) * kGqFrameBytes + lane * 16;     // relative to u
#pragma unroll
  for (int i = 0; i < kRounds; ++i) {
    const int unit = lane + 64 * (kRounds * tile + i);
    const int c = unit / kGqRowUnits, ir = unit - c * kGqRowUnits;
    const bool valid = unit < kGqFrameUnits && c < n_ch;
    src_off[i] = valid ? static_cast<unsigned>(c) * ch_stride + static_cast<unsigned>(ir) : 0u;
    dst_off[i] = valid ? c * kGqRowBytes + gq_unit_slot(ir) * 16 : dump_off;
    all_mask |= valid && ir < (kBins - 1) / 4 ? 1u << i : 0u;      // units 0..119: bins 0..479
    first_mask |= valid && ir == (kBins - 1) / 4 ? 1u << i : 0u;   // unit 120: bin 480 and three words past the row
  }
  // frame indices are 32-bit (the host checks N F < 2^31): a 64-bit division is ~120 instructions
  const unsigned total = static_cast<unsigned>(a.N * a.F), n_frames = static_cast<unsigned>(a.F);
  auto frame_base = [&](unsigned f) {                              // a frame past the end: the last one (copied, not used)
    f = f < total ? f : total - 1;
    const unsigned n = f / n_frames, t = f - n * n_frames;
    return spec + (static_cast<unsigned long>(n) * static_cast<unsigned long>(a.C * a.F) + t) * kGqRowUnits;
  };
  int om, on;                                                      // rows of the channels of the lane's pair
  {
    int p = 16 * tile + (lane & 15);
    if (p >= n_pairs) p = n_pairs - 1;                             // rows past the last pair repeat it (never stored)
    int cm, cn;
    pair_channels(p, n_ch, cm, cn);
    om = cm * kGqRowBytes;
    on = cn * kGqRowBytes;
  }
  uint4 pre[kRounds];
  auto request = [&](const uint4* src) {
#pragma unroll
    for (int i = 0; i < kRounds; ++i) pre[i] = src[src_off[i]];
  };
  auto copy = [&]() {                                              // requested words -> LDS, and this half's silent-bin flag
    bool zero = false;
#pragma unroll
    for (int i = 0; i < kRounds; ++i) {
      const uint4 w = pre[i];
      *reinterpret_cast<uint4*>(u + dst_off[i]) = w;
      const bool z_first = w.x == 0u;
      const bool z_all = z_first | (w.y == 0u) | (w.z == 0u) | (w.w == 0u);
      zero |= (((all_mask >> i) & 1u) != 0u && z_all) | (((first_mask >> i) & 1u) != 0u && z_first);
    }
    const bool any = __builtin_amdgcn_ballot_w64(zero) != 0;
    if (lane == 0) flags[2 * slot + tile] = any ? 1 : 0;           // (a word per wavefront: nothing to clear, no race)
  };
  const unsigned stride = kGqFrames * gridDim.x;
  const unsigned first = kGqFrames * blockIdx.x;                   // < total: the host launches at most ceil(total / 4) workgroups
  const unsigned n_iter = (total - first + stride - 1) / stride;   // frames per slot, the same for the whole workgroup
  unsigned f = first + slot;
  request(frame_base(f));
  __syncthreads();                                                 // the table and the cleared flags are in place
  copy();
  lds_barrier();
  for (unsigned it = 0; it < n_iter; ++it) {                       // uniform trip count
    {
      const int has_zero = flags[2 * slot] | flags[2 * slot + 1];
      // the next frame's requests go out one per step of the products (a CU issues vector memory instructions at about
      // 16 bytes per cycle: all wavefronts issuing theirs at once between the barriers took 2 000 cycles per frame)
      const uint4* next = frame_base(f + stride);
      auto per_step = [&](int ks) {
        if (ks < kRounds) pre[ks] = next[src_off[ks]];
      };
      // (a slot past its last frame still runs the products, on whatever its LDS holds, and stores nothing: one
      // definition of the requests in the loop, which the register allocator needs to keep them where they land)
      f32x4 acc_c[kGmLagTiles], acc_s[kGmLagTiles];
      gcc_q15_tile(u, table, lane, om, on, has_zero != 0, acc_c, acc_s, per_step);
      if (f < total) {
        const unsigned n = f / n_frames, t = f - n * n_frames;
        gm_store_pairs(a, acc_c, acc_s, 16 * tile + (lane & 15), lane >> 4, n_pairs, n, t,
                       has_zero ? 1.0f / 960.0f : kGqOutputScale);
      }
      f += stride;
    }
    lds_barrier();                                                 // every reader of the frames and of the flags is done
    copy();                                                        // (after the last frame: of a frame nobody reads)
    lds_barrier();                                                 // the next frame and its flags are in LDS
  }
}

// fp32 -> IEEE binary16 bits, round to nearest even (host side of the table build; |v| <= 2, no overflow handling needed)
static unsigned short half_bits(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  const int exp = static_cast<int>((u >> 23) & 0xff) - 127 + 15;
  unsigned man = u & 0x7fffffu;
  if (exp <= 0) {                                   // subnormal half (or zero)
    if (exp < -10) return static_cast<unsigned short>(sign);
    man |= 0x800000u;
    const int shift = 14 - exp;                     // 24-bit significand -> 10 bits at exponent 0
    const unsigned half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
    return static_cast<unsigned short>(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1 : 0)));
  }
  const unsigned half = (static_cast<unsigned>(exp) << 10) | (man >> 13), rem = man & 0x1fffu;
  return static_cast<unsigned short>(sign | (half + ((rem > 0x1000u || (rem == 0x1000u && (half & 1))) ? 1 : 0)));
}

// Called once per device by seld_init (seld_capi.hip): the library allocates nothing at call time.
int build_gcc_table(DeviceState* st) {
  if (st->gcc_table) return kOk;
  std::vector<unsigned short> host(kGmTableBytes / 2);
  if (int rc = seld_gcc_table_host(host.data())) return rc;
  SELD_HIP_TRY(hipMalloc(&st->gcc_table, kGmTableBytes));
  SELD_HIP_TRY(hipMemcpy(st->gcc_table, host.data(), kGmTableBytes, hipMemcpyHostToDevice));
  return kOk;
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

int seld_gcc_table_host(uint16_t* table) {
  using namespace seld;
  if (!table) return fail(kErrInvalidArgument, "seld_gcc_table_host: null pointer");
  for (int part = 0; part < 2; ++part)
    for (int tile = 0; tile < kGmLagTiles; ++tile)
      for (int ks = 0; ks < kGmKSteps; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j)
            table[((((part * kGmLagTiles + tile) * kGmKSteps + ks) * 64) + lane) * 8 + j] =
                half_bits(static_cast<float>(gcc_table_value(part, tile, ks, lane, j)));
  return kOk;
}

static int launch_gcc_q15(seld::DeviceState* st, const uint32_t* phasors, int64_t N, int64_t C, int64_t F, float* out,
                          int64_t sN, int64_t sC, int64_t sM, int64_t sT, void* stream_) {
  using namespace seld;
  GccMfmaArgs a{reinterpret_cast<const float*>(phasors), out, N, C, F, sN, sC, sM, sT, st->gcc_table};
  if (need_lds(st, kAttrGccMfmaQ15)) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gcc_mfma_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kGmLdsBytes));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gcc_q15_kernel<1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kGqLdsBytes));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gcc_q15_kernel<2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kGqLdsBytes));
    lds_attr_set(st, kAttrGccMfmaQ15);
  }
  // SELD_GCC=planar selects the kernel that unpacks the words to fp32 in LDS (the round-3a path): developer A/B, read per
  // call because the tests run both kernels in one process
  const char* which = getenv("SELD_GCC");
  const bool planar = which && which[0] == 'p';
  // persistent workgroups: the 96 KB table is staged once per workgroup
  if (planar) {
    long blocks = (N * F + 1) / 2;
    if (blocks > st->num_cus) blocks = st->num_cus;
    hipLaunchKernelGGL(gcc_mfma_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(kGmThreads), kGmLdsBytes,
                       static_cast<hipStream_t>(stream_), a);
  } else {
    long blocks = (N * F + kGqFrames - 1) / kGqFrames;
    if (blocks > st->num_cus) blocks = st->num_cus;
    if (C * (C - 1) / 2 > 16)                                 // 7 or 8 channels: two tiles of 16 pairs, a wavefront each
      hipLaunchKernelGGL(gcc_q15_kernel<2>, dim3(static_cast<unsigned>(blocks)), dim3(2 * kGqFrames * 64), kGqLdsBytes,
                         static_cast<hipStream_t>(stream_), a);
    else
      hipLaunchKernelGGL(gcc_q15_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(kGqFrames * 64), kGqLdsBytes,
                         static_cast<hipStream_t>(stream_), a);
  }
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gcc_phat_q15(const uint32_t* phasors, int64_t N, int64_t C, int64_t F, float* out, int64_t sN, int64_t sC,
                      int64_t sM, int64_t sT, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!phasors || !out) return fail(kErrInvalidArgument, "seld_gcc_phat_q15: null pointer");
  if (N <= 0 || F <= 0) return fail(kErrInvalidArgument, "seld_gcc_phat_q15: N and F must be positive");
  if (C < 2 || C > 8) return fail(kErrUnsupported, "seld_gcc_phat_q15: 2..8 channels");
  if (C * F * kPhasorPitch >= (1L << 31)) return fail(kErrUnsupported, "seld_gcc_phat_q15: a clip's phasors exceed 2^31 words");
  if (N * F + 4L * 1024 >= (1L << 31)) return fail(kErrUnsupported, "seld_gcc_phat_q15: more than 2^31 frames");
  // the kernel stores four consecutive lags as one vector: unit lag stride, 16-byte aligned rows
  const bool vector_rows = sM == 1 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && sN % 4 == 0 && sC % 4 == 0 && sT % 4 == 0;
  if (!vector_rows) return fail(kErrUnsupported, "seld_gcc_phat_q15: needs unit lag stride and 16-byte aligned rows");
  return launch_gcc_q15(st, phasors, N, C, F, out, sN, sC, sM, sT, stream_);
}

int seld_gcc_phat(const float* spec_complex, int64_t N, int64_t C, int64_t F, float* out, int64_t sN, int64_t sC,
                  int64_t sM, int64_t sT, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!spec_complex || !out) return fail(kErrInvalidArgument, "seld_gcc_phat: null pointer");
  if (N <= 0 || F <= 0) return fail(kErrInvalidArgument, "seld_gcc_phat: N and F must be positive");
  if (C < 2 || C > 8) return fail(kErrUnsupported, "seld_gcc_phat: 2..8 channels");
  if (C * F * kBins >= (1L << 31)) return fail(kErrUnsupported, "seld_gcc_phat: a clip's spectra exceed 2^31 complex values");
  // SELD_GCC=fft selects the round-2a kernel (14 packed pruned inverse FFTs per frame on the vector units): developer A/B
  const char* which = getenv("SELD_GCC");                          // read per call: the tests run both kernels in one process
  const bool use_fft = which && which[0] == 'f';
  // the matrix-core kernel stores four consecutive lags as one vector: unit lag stride, 16-byte aligned rows
  const bool vector_rows = sM == 1 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && sN % 4 == 0 && sC % 4 == 0 && sT % 4 == 0;
  if (!use_fft && vector_rows) {
    GccMfmaArgs a{spec_complex, out, N, C, F, sN, sC, sM, sT, st->gcc_table};
    long blocks = (N * F + 1) / 2;
    if (blocks > st->num_cus) blocks = st->num_cus;         // persistent: the 96 KB table is staged once per workgroup
    if (need_lds(st, kAttrGccMfma)) {
      SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gcc_mfma_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kGmLdsBytes));
      lds_attr_set(st, kAttrGccMfma);
    }
    hipLaunchKernelGGL(gcc_mfma_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(kGmThreads), kGmLdsBytes,
                       static_cast<hipStream_t>(stream_), a);
    SELD_HIP_TRY(hipGetLastError());
    return kOk;
  }
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
