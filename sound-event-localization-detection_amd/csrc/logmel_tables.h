// Host-side construction of the log-mel constant tables (pure C++, no HIP): shared by
// libseld_hip.so (seld_capi.hip) and by the CPU lane emulator used in tests (tests/emu/).
#pragma once

#include <math.h>

#include <vector>

#include "logmel_core.h"

namespace seld {

// torch.hann_window(960, periodic=True)
inline void hann_window(std::vector<float>& w) {
  w.resize(kNfft);
  for (int n = 0; n < kNfft; ++n) w[n] = static_cast<float>(0.5 - 0.5 * cos(2.0 * M_PI * n / kNfft));
}

// W_960^{n2*k1}, [30][32][2]
inline void stage_twiddles(std::vector<float>& tw) {
  tw.resize(kN2 * kN1 * 2);
  for (int n2 = 0; n2 < kN2; ++n2)
    for (int k1 = 0; k1 < kN1; ++k1) {
      const double ang = -2.0 * M_PI * static_cast<double>(n2 * k1) / kNfft;
      tw[(n2 * kN1 + k1) * 2 + 0] = static_cast<float>(cos(ang));
      tw[(n2 * kN1 + k1) * 2 + 1] = static_cast<float>(sin(ang));
    }
}

// ---- default HTK mel filterbank (torchaudio melscale_fbanks, norm=None), fp32 [481][64].
// The Python host overrides it with the table built by the very torch ops the reference's
// dependency uses (seld_set_mel_filterbank) so the weights agree to the last bit.
inline void default_mel_filterbank(std::vector<float>& fb) {
  const int nf = kBins, nm = kMels;
  const double sr_half = 12000.0;
  auto hz_to_mel = [](double f) { return 2595.0 * log10(1.0 + f / 700.0); };
  const float m_min = static_cast<float>(hz_to_mel(0.0));
  const float m_max = static_cast<float>(hz_to_mel(sr_half));
  std::vector<float> f_pts(nm + 2);
  const float step = (m_max - m_min) / static_cast<float>(nm + 1);
  for (int i = 0; i < nm + 2; ++i) {
    const float m = (i < (nm + 2) / 2) ? m_min + step * static_cast<float>(i)
                                       : m_max - step * static_cast<float>(nm + 1 - i);
    f_pts[i] = static_cast<float>(700.0 * (pow(10.0, static_cast<double>(m) / 2595.0) - 1.0));
  }
  fb.assign(static_cast<size_t>(nf) * nm, 0.0f);
  for (int k = 0; k < nf; ++k) {
    const float f = static_cast<float>(sr_half * k / (nf - 1));
    for (int j = 0; j < nm; ++j) {
      const float down = (-1.0f * (f_pts[j] - f)) / (f_pts[j + 1] - f_pts[j]);
      const float up = (f_pts[j + 2] - f) / (f_pts[j + 2] - f_pts[j + 1]);
      const float v = fminf(down, up);
      fb[static_cast<size_t>(k) * nm + j] = v > 0.0f ? v : 0.0f;
    }
  }
}

// Build the per-lane sparse description of fb: bin k is owned by the lane of its lowest
// non-zero filter.  Returns an error text (nullptr = ok) if a bin feeds more than two (adjacent) filters or a lane would own
// more than kMelMaxCnt bins.
inline const char* build_sparse_mel(const std::vector<float>& fb, std::vector<int>& b0,
                                    std::vector<float>& wd, std::vector<float>& wu) {
  b0.assign(kMels, 0);
  wd.assign(kMels * kMelMaxCnt, 0.0f);
  wu.assign(kMels * kMelMaxCnt, 0.0f);
  std::vector<int> cnt(kMels, 0), last(kMels, -1);
  for (int k = 0; k < kBins; ++k) {
    int lo = -1, nnz = 0;
    for (int j = 0; j < kMels; ++j)
      if (fb[static_cast<size_t>(k) * kMels + j] != 0.0f) {
        if (lo < 0) lo = j;
        ++nnz;
        if (j > lo + 1) return "mel filterbank: a bin feeds non-adjacent filters";
      }
    if (nnz == 0) continue;
    if (cnt[lo] == 0) b0[lo] = k;
    if (cnt[lo] > 0 && last[lo] != k - 1)
      return "mel filterbank: bins owned by one filter are not contiguous";
    if (cnt[lo] >= kMelMaxCnt) return "mel filterbank: a filter owns more than 24 bins";
    wd[cnt[lo] * kMels + lo] = fb[static_cast<size_t>(k) * kMels + lo];
    wu[cnt[lo] * kMels + lo] = (lo + 1 < kMels) ? fb[static_cast<size_t>(k) * kMels + lo + 1] : 0.0f;
    last[lo] = k;
    ++cnt[lo];
  }
  for (int j = 0; j < kMels; ++j)   // lanes that own nothing read (and zero-weight) bins near b0 = 0
    if (cnt[j] == 0) b0[j] = 0;
  return nullptr;
}

// Placement of the power rows (logmel_core.h, kPPitch): pos[0..63] = seg[j], the start of lane j's segment, chosen
// greedily in filter order as the first multiple of 4 floats behind the previous segment whose 16-byte bank group,
// (seg / 4) mod 16, is still free in the lane's ds_read_b128 service group; pos[64 + 32 r + l] = where bin l + 32 r is
// written (kPDummy for bins no filter uses and for l + 32 r > 480).  Error text if the rows do not fit.
inline const char* place_power_rows(const std::vector<float>& fb, const std::vector<int>& b0, std::vector<int>& pos) {
  std::vector<int> cnt(kMels, 0), owner(kBins, -1);
  for (int k = 0; k < kBins; ++k)
    for (int j = 0; j < kMels; ++j)
      if (fb[static_cast<size_t>(k) * kMels + j] != 0.0f) {
        owner[k] = j;
        ++cnt[j];
        break;
      }
  pos.assign(kMelPosInts, kPDummy);
  bool used[4][16] = {};
  int end = 0;
  for (int j = 0; j < kMels; ++j) {
    const int g = b128_lane_group(j);
    int s = (end + 3) & ~3;
    while (used[g][(s >> 2) & 15]) s += 4;
    used[g][(s >> 2) & 15] = true;
    pos[j] = s;
    end = s + (cnt[j] > 0 ? cnt[j] : 1);
    // a lane reads kMelMaxCnt cells from seg[j] on (zero weights past its own bins): they must exist and not be the dummy
    if (s + kMelMaxCnt > kPDummy) return "mel filterbank: the power rows do not fit the LDS tile";
  }
  for (int k = 0; k < kBins; ++k)
    if (owner[k] >= 0) pos[64 + k] = pos[owner[k]] + (k - b0[owner[k]]);
  return nullptr;
}

}  // namespace seld
