// Log-mel feature kernel -- per-lane phase functions (shared by the gfx950 kernel and by the
// host-side lane emulator used in CPU tests, tests/emu/).
//
// Replaces the reference's torchaudio call chain dataset.py:27-58
// (MelSpectrogram -> AmplitudeToDB) with ONE fused pass:
//   PCM -> reflect pad -> 960-sample frames, hop 480 -> periodic Hann -> 960-pt DFT
//       -> |X|^2 (481 bins) -> sparse HTK mel (64) -> 10*log10(max(., 1e-10))
//
// Mapping onto a 64-lane CDNA4 wavefront (no inter-wave communication at all):
//   * Two REAL frames (t, t+1 of one channel) are packed as re/im of one complex 960-pt
//     transform; each half-wavefront (32 lanes) owns one packed transform, so one wavefront
//     iteration produces 4 consecutive frames of one (clip, channel) row.
//   * 960 = 32 x 30.  Stage A: lane n2 (<30) holds x[30*n1 + n2], n1 = 0..31, in registers
//     and runs a straight-line 32-pt DFT (fft_gen.h), multiplies by W_960^{n2*k1}
//     (two-level twiddles from the workgroup's LDS table) and writes column n2 of a [32][31] complex LDS tile.
//     Because hop = 480 = 16*30, frame t+1's first half is frame t's second half IN THE SAME
//     LANE, so a packed pair costs 48 coalesced dword loads per lane, not 64.
//     Stage B: lane k1 reads row k1 (pitch 31 complex -> conflict free) and runs the
//     straight-line 30-pt prime-factor DFT: lane k1 ends with Z[k1 + 32*k2], k2 = 0..29.
//   * Un-packing needs Z[960-k]: the upper half of Z goes through LDS once (mirror read),
//     then |Xa|^2, |Xb|^2 for bins 0..480 are written to a per-frame power row in LDS.
//   * Mel: every bin feeds at most two adjacent triangular filters.  Bin k is owned by the
//     lane of its LOWER filter j; that lane accumulates A_j += fb[k][j] P[k] and
//     B_j += fb[k][j+1] P[k] over its <=24 contiguous bins; mel[j] = A_j + B_{j-1}.
//   * dB and store: lane j holds mel band j of 4 frames.
#pragma once

#include <stdint.h>

#include "fft_gen.h"

namespace seld {

constexpr int kNfft = 960;
constexpr int kHop = 480;
constexpr int kBins = 481;
constexpr int kMels = 64;
constexpr int kN1 = 32;                 // stage-A in-lane transform length (index n1 / k1)
constexpr int kN2 = 30;                 // stage-B in-lane transform length (index n2 / k2)
constexpr int kEPitch = 31;             // complex elements per E row (30 used + 1 pad)
constexpr int kEFloats = 2 * kN1 * kEPitch * 2;          // 3968 floats: exchange tile, 2 halves
constexpr int kZmFloatsPerHalf = kBins * 2;              // 962: mirror buffer (aliases E)
// Power rows (|X|^2 per bin, one row per frame slot): they alias the mirror buffer, which is dead once phase_c_load has
// its values in registers.  A row is NOT in bin order: the bins of mel filter j (those lane j accumulates) form a
// contiguous SEGMENT starting at seg[j], a multiple of 4 floats, and the segments are placed so that (seg[j] / 4) mod 16
// is distinct within each of the four 16-lane groups in which the LDS services a ds_read_b128
// (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) -- lane j's i-th 16-byte read then
// hits its own four of the 64 banks.  Bin order put the lanes' segments 2..23 floats apart and read them dword by
// dword: 3-way conflicts on average, a third of all LDS cycles of the kernel (profiles/r01_pmc_logmel_summary.json:
// SQ_LDS_BANK_CONFLICT 38.7 M of SQ_LDS_IDX_ACTIVE 110.4 M).  place_power_rows (logmel_tables.h) computes the placement.
constexpr int kPOff = 0;
constexpr int kPPitch = 800;
constexpr int kPDummy = kPPitch - 1;                     // where bins nobody accumulates are written (never read with weight != 0)
constexpr int kLdsFloatsPerWave = kEFloats;              // 3968 floats = 15872 B
constexpr int kMelMaxCnt = 24;          // max bins owned by one lane (checked when tables are built)
constexpr int kFramesPerIter = 4;
constexpr float kAmin = 1e-10f;

static_assert(kPOff + 4 * kPPitch <= kEFloats, "power rows must fit in the E tile");
constexpr int kMelPosInts = 64 + 16 * 32;   // seg[64] followed by the power-row offset of bin l + 32 r at [64 + 32 r + l]

// Device-resident constant tables (built once by seld_init / seld_set_mel_filterbank).
struct LogmelTables {
  const float* window;    // [960]      periodic Hann
  const float* twiddle;   // [30][32][2] W_960^{n2*k1} (re, im)
  const int* mel_b0;      // [64]       first bin owned by lane j
  const float* mel_wd;    // [24][64]   fb[b0_j+i][j]    (i-major: one coalesced 256-B load per i)
  const float* mel_wu;    // [24][64]   fb[b0_j+i][j+1]
  const int* mel_pos;     // [64 + 512] seg[j], then the power-row offset of every bin (kMelPosInts)
};

// Workgroup-shared constant tables in LDS, stored lane-major in float4 quads so that every read is a
// linear, conflict-free ds_read_b128 (lane l reads quad [q][l]).  Nothing constant lives in VGPRs:
// the registers are spent on data (current samples, the next iteration's prefetched samples, the DFT).
constexpr int kTabWin = 0;                         // [8][64][4]  Hann window w[30*n1 + n2], n1 = 4q+j
constexpr int kTabTw = kTabWin + 8 * 64 * 4;       // [5][64][4]  10 complex: lo[1..7] = W^{n2*b}, hi[1..3] = W^{n2*8a}
constexpr int kTabMel = kTabTw + 5 * 64 * 4;       // [12][64][4] (wd_i, wu_i, wd_{i+1}, wu_{i+1}), i = 2q
constexpr int kTabFloats = kTabMel + 12 * 64 * 4;  // 6400 floats = 25600 B

struct alignas(16) PowerQuad { float p[4]; };     // four consecutive bins of a power row: ONE ds_read_b128

// the 16-lane group that services lane `lane`'s part of a ds_read_b128 (0..3)
SELD_HD int b128_lane_group(int lane) {
  const int l = lane & 31;
  const bool first = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
  return (first ? 0 : 1) + 2 * (lane >> 5);
}

// per frame slot: (A_j, B_j) -- the sums over this lane's bins weighted for its own filter and for the next one
struct LaneAcc {
  cf ab[kFramesPerIter];
};

SELD_HD int e_index(int h, int k1, int n2) { return ((h * kN1 + k1) * kEPitch + n2) * 2; }
SELD_HD int zm_index(int h, int idx) { return h * kZmFloatsPerHalf + idx * 2; }

SELD_HD float sample_to_float(float v) { return v; }
SELD_HD float sample_to_float(int16_t v) { return static_cast<float>(v) * (1.0f / 32768.0f); }

// Value of flat element `e` of the LDS table block (used to fill it, on the GPU and in the emulator).
SELD_HD float table_value(const LogmelTables& t, int e) {
  const int lane = (e & 255) >> 2;
  const int j = e & 3;
  const int l = lane & 31;
  const int n2 = l < kN2 ? l : kN2 - 1;
  if (e < kTabTw) {
    const int n1 = 4 * (e >> 8) + j;
    // half the window: the packed transform then yields X/2, so |Xa|^2 = ar^2 + ai^2 needs no 1/4
    return 0.5f * t.window[kN2 * n1 + n2];
  }
  if (e < kTabMel) {
    const int ci = 2 * ((e - kTabTw) >> 8) + (j >> 1);         // 0..9
    const int k1 = ci < 7 ? ci + 1 : 8 * (ci - 6);
    return t.twiddle[(n2 * kN1 + k1) * 2 + (j & 1)];
  }
  const int i = 2 * ((e - kTabMel) >> 8) + (j >> 1);
  return (j & 1) ? t.mel_wu[i * 64 + lane] : t.mel_wd[i * 64 + lane];
}

// ---- Sample fetch: the 48 samples x[480*(fa-1) + 30*m + n2], m = 0..47 of this lane (3 half-frames).
// `fa` = index of the first (real-part) frame of this half-wavefront's pair.  kInterior: every index
// is inside [0, L) -- no reflection, immediate-offset loads.
template <typename T, bool kInterior>
SELD_HD void load_samples(int lane, const T* row, long L, long fa, float (&s)[48]) {
  const int l = lane & 31;
  const int n2 = l < kN2 ? l : kN2 - 1;
  if (kInterior) {
    const T* p = row + static_cast<long>(kHop) * (fa - 1) + n2;
#pragma unroll
    for (int m = 0; m < 48; ++m) s[m] = sample_to_float(p[kN2 * m]);
  } else {
    const int len = static_cast<int>(L);
    const int base = kHop * (static_cast<int>(fa) - 1) + n2;
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      int idx = base + kN2 * m;
      idx = idx < 0 ? -idx : idx;                            // reflect (center=True, pad_mode='reflect')
      idx = idx >= len ? 2 * (len - 1) - idx : idx;
      idx = idx < 0 ? 0 : (idx >= len ? len - 1 : idx);      // frames past the end: any finite value
      s[m] = sample_to_float(row[idx]);
    }
  }
}

// Per-lane constants of stage A, read once per wavefront from the workgroup's table block and kept in registers
// across the iterations (32 VGPRs; round 1 re-read them from LDS every iteration because its 234-VGPR body had no room:
// 8 of the ~105 KB of LDS traffic per wavefront-iteration).
struct LaneConsts {
  float win[kN1];     // 0.5 * Hann at 30 n1 + n2
  const float* twq;   // this lane's quad of the twiddle rows ([5][64][4] floats, row v at +256 v): lo[1..7] = W_960^{n2 b},
                      // hi[1..3] = W_960^{n2 8 a} -- re-read every iteration (with them the body spills)
};

SELD_HD void load_twiddles(const float* twq, cf (&tw)[10]) {
#pragma unroll
  for (int v = 0; v < 5; ++v) {
    tw[2 * v] = cf_make(twq[v * 256], twq[v * 256 + 1]);
    tw[2 * v + 1] = cf_make(twq[v * 256 + 2], twq[v * 256 + 3]);
  }
}

SELD_HD void load_lane_consts(int lane, const float* tab, LaneConsts& k) {
  const float* tl = tab + lane * 4;                    // this lane's quad in every table row
#pragma unroll
  for (int n1 = 0; n1 < kN1; ++n1) k.win[n1] = tl[kTabWin + (n1 >> 2) * 256 + (n1 & 3)];
  k.twq = tl + kTabTw;
}

// Tail of stage A: multiply the 32-pt DFT outputs by the two-level twiddles W_960^{n2*k1} = hi[k1>>3]*lo[k1&7] and store
// column l of this half-wavefront's [32][31]-complex exchange tile (one 8-byte store per element).
SELD_HD void stage_a_finish(int lane, const cf (&z)[kN1], const cf (&tw)[10], float* lds) {
  const int h = lane >> 5;
  const int l = lane & 31;
  if (l < kN2) {
    float* e = lds + e_index(h, 0, l);                 // column l of this half's tile; row k1 is at +62*k1
    // lo twiddles for all 32, then hi twiddles: consecutive products are independent of each other
    cf y[kN1];
#pragma unroll
    for (int k1 = 0; k1 < kN1; ++k1) y[k1] = (k1 & 7) != 0 ? cf_cmul(z[k1], tw[(k1 & 7) - 1]) : z[k1];
#pragma unroll
    for (int k1 = 0; k1 < kN1; ++k1) {
      if ((k1 >> 3) != 0) y[k1] = cf_cmul(y[k1], tw[6 + (k1 >> 3)]);
      *reinterpret_cast<cf*>(e + k1 * kEPitch * 2) = y[k1];
    }
  }
}

// NOTE on addressing: every LDS access below is written as  (one per-lane base pointer)[compile-time
// constant]  so that it becomes a single base VGPR plus the DS instruction's immediate offset.  Spelling
// the index as one expression makes the compiler hoist a separate loop-invariant address register for
// every distinct constant (dozens of VGPRs, which then spill).

// ---- Phase A: window, pack two frames, 32-pt DFT, twiddle, LDS column store.
SELD_HD void phase_a(int lane, const float (&s)[48], const LaneConsts& k, float* lds) {
  cf z[kN1];
#pragma unroll
  for (int n1 = 0; n1 < kN1; ++n1) {
    // frame fa -> real part, frame fa+1 (the same samples shifted by 480 = 16*30) -> imaginary part
    z[n1] = cf_make(k.win[n1] * s[n1], k.win[n1] * s[n1 + 16]);
  }
  dft32(z);
  cf tw[10];
  load_twiddles(k.twq, tw);
  stage_a_finish(lane, z, tw, lds);
}

// ---- Phase B: 30-pt DFT along n2 for row k1 = l, then park the upper half-spectrum for the mirror read.
SELD_HD void phase_b(int lane, float* lds, cf (&z)[kN2]) {
  const int h = lane >> 5;
  const int l = lane & 31;
  const cf* e = reinterpret_cast<const cf*>(lds + e_index(h, l, 0));   // row l
#pragma unroll
  for (int n2 = 0; n2 < kN2; ++n2) z[n2] = e[n2];
  dft30(z);   // z[k2] = Z[l + 32*k2]
}

SELD_HD void phase_b_store(int lane, float* lds, const cf (&z)[kN2]) {
  const int h = lane >> 5;
  const int l = lane & 31;
  cf* zm = reinterpret_cast<cf*>(lds + zm_index(h, l));  // entry (l + 32*k2 - 480) is at +32*(k2-15) complex
#pragma unroll
  for (int k2 = 15; k2 < kN2; ++k2) zm[32 * (k2 - 15)] = z[k2];
  if (l == 0) zm[480] = z[0];   // Z[960] == Z[0]
}

// ---- Phase C: un-pack the two real spectra and write |X|^2 for bins 0..480 of both frames.
//   Xa[k] = (Z[k] + conj Z[N-k]) / 2 ,  Xb[k] = (Z[k] - conj Z[N-k]) / (2i); the 1/2 is in the window table
SELD_HD void phase_c_load(int lane, const float* lds, cf (&m)[16]) {
  const int h = lane >> 5;
  const int l = lane & 31;
  // mirror entry 480 - (l + 32 r) = (0 - l) + 32*(15 - r): base at r = 15 (entry -l), ascending by 32 complex
  const cf* mp = reinterpret_cast<const cf*>(lds + zm_index(h, 0) - 2 * l);
#pragma unroll
  for (int r = 0; r < 15; ++r) m[r] = mp[32 * (15 - r)];
  m[15] = *reinterpret_cast<const cf*>(lds + zm_index(h, 0));   // r = 15: bin 480 (lane 0 only; others read a dummy)
}

// Where this lane's 16 bins (l + 32 r) live in the power row of frame slot 2h (slot 2h+1 is kPPitch further): loop
// invariant, kept in registers so that every store is one address register + an immediate offset.
SELD_HD void power_row_pointers(int lane, float* lds, const int* mel_pos, float* (&pp)[16]) {
  const int h = lane >> 5;
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) pp[r] = lds + kPOff + 2 * h * kPPitch + mel_pos[64 + 32 * r + l];
}

// With Z = (zr, zi) and the mirror M = (mr, mi):  Xa = (zr + mr, zi - mi),  Xb = (zi + mi, -(zr - mr)).  The two powers
// come out as ONE pair in four packed operations:  U = (zr + mr, zr - mr),  V = (zi - mi, zi + mi),
// (|Xa|^2, |Xb|^2) = U * U + V * V  (splats of zr, mr, zi, mi are operand selects).
SELD_HD void phase_c_store(int lane, float* const (&pp)[16], const cf (&z)[kN2], const cf (&m)[16]) {
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < 15 || l == 0) {
      const cf u = cf_fma(cf_make(m[r].x, m[r].x), cf_make(1.0f, -1.0f), cf_make(z[r].x, z[r].x));
      const cf v = cf_fma(cf_make(m[r].y, m[r].y), cf_make(-1.0f, 1.0f), cf_make(z[r].y, z[r].y));
      const cf p = cf_fma(v, v, cf_mul(u, u));
      pp[r][0] = p.x;
      pp[r][kPPitch] = p.y;
    }
  }
}

// Variant of phase C for the GCC-PHAT feature set: besides the power rows, every bin's PHASOR X / |X| goes to global memory
// as a pair of signed 16-bit fixed-point numbers (re | im << 16, scale 32767: |component| <= 1, step 3.05e-5 -- sixteen
// times finer than fp16 near 1) -- 4 B per bin instead of the 8 B of the complex64 spectrum, and the phase transform's
// reciprocal square root is taken here, once per (channel, bin), from the |X|^2 that is being formed anyway.  A silent bin
// (|X|^2 <= kSilencePower) is stored as 0 (csrc/spatial.hip turns its products into 1).  Rows of kPhasorPitch words.
constexpr float kSilencePower = 1e-12f;
constexpr int kPhasorPitch = 488;            // 481 bins + 7: rows stay 16-byte aligned (bins 481..487 are never written)

SELD_HD unsigned phasor_q15(float re, float im, float power) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rsq_f32 (1 ulp, far below the 3e-5 quantisation step), then ONE v_cvt_pknorm_i16_f32: both components clamped to
  // [-1, 1], scaled by 32767 and rounded to nearest, packed (re | im << 16)
  const float rs = power > kSilencePower ? __frsqrt_rn(power) : 0.0f;
  typedef short short2_t __attribute__((ext_vector_type(2)));
  const short2_t q = __builtin_amdgcn_cvt_pknorm_i16(re * rs, im * rs);
  return __builtin_bit_cast(unsigned, q);
#else
  const float inv = power > kSilencePower ? 32767.0f * (1.0f / sqrtf(power)) : 0.0f;
  const int qr = static_cast<int>(rintf(re * inv)), qi = static_cast<int>(rintf(im * inv));
  return (static_cast<unsigned>(qr) & 0xffffu) | (static_cast<unsigned>(qi) << 16);
#endif
}

// ``base`` is wavefront-uniform (a scalar register pair), ``off_a`` this lane's 32-bit word offset of bin l of frame a
// (frame b one row further): every store is scalar base + one offset register + an immediate.
SELD_HD void phase_c_store_phasors(int lane, float* const (&pp)[16], const cf (&z)[kN2], const cf (&m)[16],
                                   unsigned* base, unsigned off_a, bool have_a, bool have_b) {
  const int l = lane & 31;
  unsigned* pa = have_a ? base + off_a : nullptr;                      // bin l + 32 r is at +32 r
  unsigned* pb = have_b ? base + off_a + kPhasorPitch : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < 15 || l == 0) {
      const cf u = cf_fma(cf_make(m[r].x, m[r].x), cf_make(1.0f, -1.0f), cf_make(z[r].x, z[r].x));   // (zr + mr, zr - mr)
      const cf v = cf_fma(cf_make(m[r].y, m[r].y), cf_make(-1.0f, 1.0f), cf_make(z[r].y, z[r].y));   // (zi - mi, zi + mi)
      const cf p = cf_fma(v, v, cf_mul(u, u));
      pp[r][0] = p.x;
      pp[r][kPPitch] = p.y;
      if (pa) pa[32 * r] = phasor_q15(u.x, v.x, p.x);                  // Xa = (zr + mr, zi - mi)
      if (pb) pb[32 * r] = phasor_q15(v.y, -u.y, p.y);                 // Xb = (zi + mi, mr - zr)
    }
  }
}

// Variant of phase C for the STFT export: the un-packed complex spectra of the two frames go straight to
// global memory (frame-major rows of 481 complex: 32 lanes x 8 B = one 256-B run per store).
//   Xa = (zr + mr, zi - mi) ,  Xb = (zi + mi, mr - zr)   (the 1/2 is in the window table)
SELD_HD void phase_c_spectrum(int lane, const cf (&z)[kN2], const cf (&m)[16], float* row_a, float* row_b) {
  const int l = lane & 31;
  cf* pa = row_a ? reinterpret_cast<cf*>(row_a) + l : nullptr;        // complex bin l + 32 r is at +32 r
  cf* pb = row_b ? reinterpret_cast<cf*>(row_b) + l : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < 15 || l == 0) {
      if (pa) pa[32 * r] = cf_fma(m[r], cf_make(1.0f, -1.0f), z[r]);
      if (pb) pb[32 * r] = cf_fma_swap(z[r], 1.0f, -1.0f, cf_make(m[r].y, m[r].x));
    }
  }
}

// ---- Fused FOA intensity vectors (logmel_iv_kernel, logmel.hip): the four wavefronts of a workgroup are the channels
// W, X, Y, Z of the SAME four frames, in lock-step.  Every wavefront leaves its |C|^2 rows in its own tile (all four use one
// mel_pos layout, so a lane finds the other channels' powers of ITS bins at its own offsets, one tile pitch apart), W also
// publishes its un-packed complex spectrum lane-major in a workgroup buffer, and X / Y / Z form
//     I_c[k] = Re(conj(W[k]) C[k]) / (eps + |W|^2 + (|X|^2 + |Y|^2 + |Z|^2) / 3)
// for their 16 bins x 2 frames in registers; once every wavefront is done with the power rows the I values take their place
// and the SAME sparse mel pass (phase_d_accumulate) projects them.  No spectrum ever goes to global memory.
constexpr int kIvChannels = 4;
constexpr int kIvSpecFloats = 16 * 2 * 64 * 2;          // W's spectrum: [r][frame a / b][lane] complex, 16 KB

// Phase C keeping the un-packed spectra of this lane's bins l + 32 r:  Xa = (zr + mr, zi - mi),  Xb = (zi + mi, mr - zr)
// (the 1/2 is in the window table); power rows exactly as phase_c_store writes them.
SELD_HD void phase_c_unpack(int lane, float* const (&pp)[16], const cf (&z)[kN2], const cf (&m)[16], cf (&xa)[16],
                            cf (&xb)[16]) {
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const cf u = cf_fma(cf_make(m[r].x, m[r].x), cf_make(1.0f, -1.0f), cf_make(z[r].x, z[r].x));   // (zr + mr, zr - mr)
    const cf v = cf_fma(cf_make(m[r].y, m[r].y), cf_make(-1.0f, 1.0f), cf_make(z[r].y, z[r].y));   // (zi - mi, zi + mi)
    const cf p = cf_fma(v, v, cf_mul(u, u));
    xa[r] = cf_make(u.x, v.x);
    xb[r] = cf_make(v.y, -u.y);
    if (r < 15 || l == 0) {
      pp[r][0] = p.x;
      pp[r][kPPitch] = p.y;
    }
  }
}

// W's wavefront: spectrum of (r, frame) at complex index (2 r + frame) * 64 + lane -- linear 8-byte stores / loads
SELD_HD void iv_publish(int lane, float* wspec, const cf (&xa)[16], const cf (&xb)[16]) {
  cf* w = reinterpret_cast<cf*>(wspec) + lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    w[(2 * r) * 64] = xa[r];
    w[(2 * r + 1) * 64] = xb[r];
  }
}

// Channel `chan` (1..3): the intensities of this lane's bins.  ``pp`` are the lane's OWN power-row pointers; channel k's
// tile is (k - chan) * kLdsFloatsPerWave floats away.
SELD_HD void iv_compute(int lane, int chan, const float* wspec, float* const (&pp)[16], const cf (&xa)[16],
                        const cf (&xb)[16], float eps, float (&ia)[16], float (&ib)[16]) {
  const int l = lane & 31;
  const cf* w = reinterpret_cast<const cf*>(wspec) + lane;
  const int t0 = -chan * kLdsFloatsPerWave;              // W's tile relative to this channel's
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    ia[r] = 0.0f;
    ib[r] = 0.0f;
    if (r < 15 || l == 0) {
      const float* q = pp[r] + t0;
      const float ea = eps + q[0] + ((q[kLdsFloatsPerWave] + q[2 * kLdsFloatsPerWave]) + q[3 * kLdsFloatsPerWave]) * (1.0f / 3.0f);
      const float eb = eps + q[kPPitch] + ((q[kLdsFloatsPerWave + kPPitch] + q[2 * kLdsFloatsPerWave + kPPitch]) +
                                           q[3 * kLdsFloatsPerWave + kPPitch]) * (1.0f / 3.0f);
      const cf wa = w[(2 * r) * 64], wb = w[(2 * r + 1) * 64];
      ia[r] = (wa.x * xa[r].x + wa.y * xa[r].y) * (1.0f / ea);          // Re(conj(W) C) = wr cr + wi ci
      ib[r] = (wb.x * xb[r].x + wb.y * xb[r].y) * (1.0f / eb);
    }
  }
}

// the intensities take the place of the power rows (same cells: the mel pass reads them with the same segment layout)
SELD_HD void iv_store_rows(int lane, float* const (&pp)[16], const float (&ia)[16], const float (&ib)[16]) {
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < 15 || l == 0) {
      pp[r][0] = ia[r];
      pp[r][kPPitch] = ib[r];
    }
  }
}

// ---- Phase D: sparse mel.  Lane j accumulates over its own contiguous bins for all 4 frames.
// The filter weights come from the workgroup's LDS table (12 linear ds_read_b128 per iteration).
SELD_HD void phase_d_accumulate(int lane, float* lds, const float* tab, int seg, LaneAcc& acc) {
  const float* tl = tab + kTabMel + lane * 4;
  // this lane's segment of the power rows; seg and kPPitch are multiples of 4 floats, the tile is 16-byte aligned
  const PowerQuad* p = reinterpret_cast<const PowerQuad*>(lds + kPOff + seg);
#pragma unroll
  for (int s = 0; s < kFramesPerIter; ++s) acc.ab[s] = cf_make(0.0f, 0.0f);
#pragma unroll
  for (int iq = 0; iq < kMelMaxCnt / 4; ++iq) {
    cf w[4];                       // (wd, wu) of bins 4 iq .. 4 iq + 3 of this lane's segment: pairs as the table stores them
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = cf_make(tl[(2 * iq + (k >> 1)) * 256 + 2 * (k & 1)], tl[(2 * iq + (k >> 1)) * 256 + 2 * (k & 1) + 1]);
    PowerQuad q[kFramesPerIter];                               // one ds_read_b128 each, conflict free (see kPPitch)
#pragma unroll
    for (int s = 0; s < kFramesPerIter; ++s) q[s] = p[s * (kPPitch / 4) + iq];
    // bins in ascending order for every sum; the four frames' chains interleaved (a packed op that consumes the result of
    // the packed op right before it costs a wait state)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int s = 0; s < kFramesPerIter; ++s)
        acc.ab[s] = cf_fma(w[k], cf_make(q[s].p[k], q[s].p[k]), acc.ab[s]);
  }
}

// 10*log10(max(p, 1e-10)) as 10*log10(2) * log2(p): one v_log_f32 (<= 1 ulp of log2, i.e. < 1e-5 dB) instead
// of the ~15-instruction log10f.  The floor is returned as exactly -100 dB (what the CPU path yields).
SELD_HD float power_to_db(float p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float l2 = __log2f(p);
#else
  const float l2 = log2f(p);
#endif
  return p > kAmin ? 3.0102999566398120f * l2 : -100.0f;
}

// mel[j] = A_j + B_{j-1}: `below[s]` is lane j - 1's B sum of frame slot s (0 for lane 0) -- one cross-lane move on the
// GPU (lane_below in logmel.hip: a DPP wave shift, no LDS round trip), the neighbour's accumulator in the lane emulator.
SELD_HD void phase_d_finish(const LaneAcc& acc, const float (&below)[kFramesPerIter], float (&db)[kFramesPerIter]) {
#pragma unroll
  for (int s = 0; s < kFramesPerIter; ++s) db[s] = power_to_db(acc.ab[s].x + below[s]);
}

}  // namespace seld
