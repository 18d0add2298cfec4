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
//     (register-resident twiddles) and writes column n2 of a [32][31] complex LDS tile.
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
constexpr int kPOff = 2 * kZmFloatsPerHalf;              // 1924: power rows start (aliases E)
constexpr int kPPitch = 484;
constexpr int kBsOff = kEFloats;                         // 3968: B_j hand-off, [4][64]
constexpr int kLdsFloatsPerWave = kEFloats + 4 * 64;     // 4224 floats = 16896 B
constexpr int kMelMaxCnt = 24;          // max bins owned by one lane (checked when tables are built)
constexpr int kFramesPerIter = 4;
constexpr int kItersPerGroup = 4;
constexpr int kFramesPerGroup = kFramesPerIter * kItersPerGroup;   // 16
constexpr float kAmin = 1e-10f;

static_assert(kPOff + 4 * kPPitch + kMelMaxCnt <= kEFloats, "power rows (+over-read) must fit in the E tile");

// Device-resident constant tables (built once by seld_init / seld_set_mel_filterbank).
struct LogmelTables {
  const float* window;    // [960]      periodic Hann
  const float* twiddle;   // [30][32][2] W_960^{n2*k1} (re, im)
  const int* mel_b0;      // [64]       first bin owned by lane j
  const float* mel_wd;    // [24][64]   fb[b0_j+i][j]    (i-major: one coalesced 256-B load per i)
  const float* mel_wu;    // [24][64]   fb[b0_j+i][j+1]
};

// Per-lane constants kept in registers for the whole kernel.
struct LaneConst {
  float win[kN1];
  // two-level twiddles: W_960^{n2*k1} = hi[k1>>3] * lo[k1&7]  (10 complex registers, not 32)
  float hir[4], hii[4];   // W^{n2*8a}, a = 0..3   (a = 0 is 1)
  float lor[8], loi[8];   // W^{n2*b},  b = 0..7   (b = 0 is 1)
  int b0;
};

struct LaneAcc {
  float a[kFramesPerIter];
  float b[kFramesPerIter];
};

SELD_HD int e_index(int h, int k1, int n2) { return ((h * kN1 + k1) * kEPitch + n2) * 2; }
SELD_HD int zm_index(int h, int idx) { return h * kZmFloatsPerHalf + idx * 2; }
SELD_HD int p_index(int slot, int k) { return kPOff + slot * kPPitch + k; }

SELD_HD float sample_to_float(float v) { return v; }
SELD_HD float sample_to_float(int16_t v) { return static_cast<float>(v) * (1.0f / 32768.0f); }

SELD_HD void load_lane_const(int lane, const LogmelTables& t, LaneConst& k) {
  const int l = lane & 31;
  const int n2 = l < kN2 ? l : kN2 - 1;
#pragma unroll
  for (int n1 = 0; n1 < kN1; ++n1) {
    k.win[n1] = t.window[kN2 * n1 + n2];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    k.hir[a] = t.twiddle[(n2 * kN1 + 8 * a) * 2 + 0];
    k.hii[a] = t.twiddle[(n2 * kN1 + 8 * a) * 2 + 1];
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    k.lor[b] = t.twiddle[(n2 * kN1 + b) * 2 + 0];
    k.loi[b] = t.twiddle[(n2 * kN1 + b) * 2 + 1];
  }
  k.b0 = t.mel_b0[lane];
}

// ---- Phase A: load 3 half-frames, window, pack two frames, 32-pt DFT, twiddle, LDS column store.
// `fa` = index of the first (real-part) frame of this half-wavefront's pair.
template <typename T, bool kInterior>
SELD_HD void phase_a(int lane, const T* row, long L, long fa, const LaneConst& k, float* lds) {
  const int h = lane >> 5;
  const int l = lane & 31;
  const int n2 = l < kN2 ? l : kN2 - 1;
  const long base = static_cast<long>(kHop) * (fa - 1) + n2;
  float s[48];
#pragma unroll
  for (int m = 0; m < 48; ++m) {
    long idx = base + kN2 * m;
    if (!kInterior) {
      if (idx < 0) idx = -idx;                       // reflect (center=True, pad_mode='reflect')
      if (idx >= L) idx = 2 * (L - 1) - idx;
      idx = idx < 0 ? 0 : (idx >= L ? L - 1 : idx);  // frames past the end: any finite value
    }
    s[m] = sample_to_float(row[idx]);
  }
  float re[kN1], im[kN1];
#pragma unroll
  for (int n1 = 0; n1 < kN1; ++n1) {
    re[n1] = k.win[n1] * s[n1];            // frame fa
    im[n1] = k.win[n1] * s[n1 + 16];       // frame fa+1 = same samples shifted by 480 = 16*30
  }
  dft32(re, im);
  if (l < kN2) {
#pragma unroll
    for (int k1 = 0; k1 < kN1; ++k1) {
      const int a = k1 >> 3, b = k1 & 7;
      float yr = re[k1], yi = im[k1];
      if (b != 0) {
        const float tr = yr * k.lor[b] - yi * k.loi[b];
        yi = yr * k.loi[b] + yi * k.lor[b];
        yr = tr;
      }
      if (a != 0) {
        const float tr = yr * k.hir[a] - yi * k.hii[a];
        yi = yr * k.hii[a] + yi * k.hir[a];
        yr = tr;
      }
      const int o = e_index(h, k1, l);
      lds[o] = yr;
      lds[o + 1] = yi;
    }
  }
}

// ---- Phase B: 30-pt DFT along n2 for row k1 = l, then park the upper half-spectrum for the mirror read.
SELD_HD void phase_b(int lane, float* lds, float (&zr)[kN2], float (&zi)[kN2]) {
  const int h = lane >> 5;
  const int l = lane & 31;
#pragma unroll
  for (int n2 = 0; n2 < kN2; ++n2) {
    const int o = e_index(h, l, n2);
    zr[n2] = lds[o];
    zi[n2] = lds[o + 1];
  }
  dft30(zr, zi);   // zr/zi[k2] = Z[l + 32*k2]
}

SELD_HD void phase_b_store(int lane, float* lds, const float (&zr)[kN2], const float (&zi)[kN2]) {
  const int h = lane >> 5;
  const int l = lane & 31;
#pragma unroll
  for (int k2 = 15; k2 < kN2; ++k2) {
    const int o = zm_index(h, l + 32 * k2 - 480);
    lds[o] = zr[k2];
    lds[o + 1] = zi[k2];
  }
  if (l == 0) {   // Z[960] == Z[0]
    const int o = zm_index(h, 480);
    lds[o] = zr[0];
    lds[o + 1] = zi[0];
  }
}

// ---- Phase C: un-pack the two real spectra and write |X|^2 for bins 0..480 of both frames.
//   Xa[k] = (Z[k] + conj Z[N-k]) / 2 ,  Xb[k] = (Z[k] - conj Z[N-k]) / (2i)
SELD_HD void phase_c_load(int lane, const float* lds, float (&mr)[16], float (&mi)[16]) {
  const int h = lane >> 5;
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = l + 32 * r;
    const int idx = (r < 15 || l == 0) ? 480 - k : 0;
    const int o = zm_index(h, idx);
    mr[r] = lds[o];
    mi[r] = lds[o + 1];
  }
}

SELD_HD void phase_c_store(int lane, float* lds, const float (&zr)[kN2], const float (&zi)[kN2],
                                  const float (&mr)[16], const float (&mi)[16]) {
  const int h = lane >> 5;
  const int l = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < 15 || l == 0) {
      const int k = l + 32 * r;
      const float ar = zr[r] + mr[r], ai = zi[r] - mi[r];
      const float br = zi[r] + mi[r], bi = zr[r] - mr[r];
      lds[p_index(2 * h, k)] = 0.25f * (ar * ar + ai * ai);
      lds[p_index(2 * h + 1, k)] = 0.25f * (br * br + bi * bi);
    }
  }
}

// ---- Phase D: sparse mel.  Lane j accumulates over its own contiguous bins for all 4 frames.
// The weights are streamed from the (L1/L2-resident, 12 KB) table every iteration rather than
// pinned in 48 VGPRs: the kernel is tuned for 2 wavefronts per SIMD (256 VGPRs).
SELD_HD void phase_d_accumulate(int lane, float* lds, const LogmelTables& t, const LaneConst& k, LaneAcc& acc) {
#pragma unroll
  for (int s = 0; s < kFramesPerIter; ++s) acc.a[s] = acc.b[s] = 0.0f;
#pragma unroll
  for (int i = 0; i < kMelMaxCnt; ++i) {
    const float wd = t.mel_wd[i * 64 + lane];
    const float wu = t.mel_wu[i * 64 + lane];
#pragma unroll
    for (int s = 0; s < kFramesPerIter; ++s) {
      const float p = lds[p_index(s, k.b0 + i)];
      acc.a[s] = fmaf(wd, p, acc.a[s]);
      acc.b[s] = fmaf(wu, p, acc.b[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < kFramesPerIter; ++s) lds[kBsOff + s * 64 + lane] = acc.b[s];
}

// 10*log10(max(p, 1e-10)); the floor is returned as exactly -100 dB (what the CPU path yields),
// independent of the last-ulp behaviour of the device log10f.
SELD_HD float power_to_db(float p) { return p > kAmin ? 10.0f * log10f(p) : -100.0f; }

SELD_HD void phase_d_finish(int lane, const float* lds, const LaneAcc& acc, float (&db)[kFramesPerIter]) {
#pragma unroll
  for (int s = 0; s < kFramesPerIter; ++s) {
    const float below = lane > 0 ? lds[kBsOff + s * 64 + lane - 1] : 0.0f;
    db[s] = power_to_db(acc.a[s] + below);
  }
}

}  // namespace seld
