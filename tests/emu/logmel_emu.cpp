// TEST TOOL (not shipped, not linked into libseld_hip.so): runs the log-mel kernel's per-lane
// phase functions (csrc/logmel_core.h) on the CPU, 64 "lanes" in lock step with a plain array
// standing in for the wavefront's LDS tile.  It lets the CPU test-suite check the kernel's
// index math (32x30 factorisation, packed-frame un-mixing, sparse mel ownership, reflection at
// the clip edges) against the oracle without a GPU.  The product never calls this.
//
//   g++ -O2 -shared -fPIC -I <csrc> tests/emu/logmel_emu.cpp -o tests/emu/libseld_emu.so
#include <stdint.h>
#include <string.h>

#include <vector>

#include "logmel_core.h"
#include "logmel_tables.h"

using namespace seld;

namespace {

struct HostTables {
  std::vector<float> window, twiddle, fb, wd, wu;
  std::vector<int> b0, pos;
  LogmelTables view() const {
    return LogmelTables{window.data(), twiddle.data(), b0.data(), wd.data(), wu.data(), pos.data()};
  }
};

template <typename T>
int run(const T* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, const float* fb_opt) {
  if (L <= kNfft / 2) return -1;
  HostTables t;
  hann_window(t.window);
  stage_twiddles(t.twiddle);
  if (fb_opt)
    t.fb.assign(fb_opt, fb_opt + static_cast<size_t>(kBins) * kMels);
  else
    default_mel_filterbank(t.fb);
  if (build_sparse_mel(t.fb, t.b0, t.wd, t.wu)) return -4;
  if (place_power_rows(t.fb, t.b0, t.pos)) return -5;
  const LogmelTables tab = t.view();

  const long F = 1 + L / kHop;
  long sN, sC, sM, sT;
  if (layout == 0) { sT = 1; sM = F; sC = kMels * F; sN = C * kMels * F; }
  else { sM = 1; sC = kMels; sT = C * kMels; sN = F * C * kMels; }

  std::vector<float> lds(kLdsFloatsPerWave, 0.0f);
  std::vector<float> tab_lds(kTabFloats);
  for (int e = 0; e < kTabFloats; ++e) tab_lds[e] = table_value(tab, e);
  std::vector<cf> z(64 * kN2), m(64 * 16);
  std::vector<float> smp(64 * 48);
  std::vector<LaneAcc> acc(64);
  const long iters_per_row = (F + kFramesPerIter - 1) / kFramesPerIter;

  for (long row = 0; row < N * C; ++row) {
    const T* rowp = pcm + row * L;
    const long n = row / C, c = row % C;
    for (long itr = 0; itr < iters_per_row; ++itr) {
      const long tf = itr * kFramesPerIter;
      const bool interior = (tf >= 1) && (static_cast<long>(kHop) * (tf + kFramesPerIter) <= L);
      for (int lane = 0; lane < 64; ++lane) {
        const long fa = tf + 2 * (lane >> 5);
        float(&s)[48] = *reinterpret_cast<float(*)[48]>(&smp[lane * 48]);
        if (interior) load_samples<T, true>(lane, rowp, L, fa, s);
        else load_samples<T, false>(lane, rowp, L, fa, s);
      }
      for (int lane = 0; lane < 64; ++lane) {
        LaneConsts consts;
        load_lane_consts(lane, tab_lds.data(), consts);
        phase_a(lane, *reinterpret_cast<float(*)[48]>(&smp[lane * 48]), consts, lds.data());
      }
      for (int lane = 0; lane < 64; ++lane)
        phase_b(lane, lds.data(), *reinterpret_cast<cf(*)[kN2]>(&z[lane * kN2]));
      for (int lane = 0; lane < 64; ++lane)
        phase_b_store(lane, lds.data(), *reinterpret_cast<cf(*)[kN2]>(&z[lane * kN2]));
      for (int lane = 0; lane < 64; ++lane)
        phase_c_load(lane, lds.data(), *reinterpret_cast<cf(*)[16]>(&m[lane * 16]));
      for (int lane = 0; lane < 64; ++lane) {
        float* pp[16];
        power_row_pointers(lane, lds.data(), t.pos.data(), pp);
        phase_c_store(lane, pp, *reinterpret_cast<cf(*)[kN2]>(&z[lane * kN2]), *reinterpret_cast<cf(*)[16]>(&m[lane * 16]));
      }
      for (int lane = 0; lane < 64; ++lane)
        phase_d_accumulate(lane, lds.data(), tab_lds.data(), t.pos[lane], acc[lane]);
      for (int lane = 0; lane < 64; ++lane) {
        float db[kFramesPerIter];
        float below[kFramesPerIter];
        for (int s = 0; s < kFramesPerIter; ++s) below[s] = lane > 0 ? acc[lane - 1].ab[s].y : 0.0f;
        phase_d_finish(acc[lane], below, db);
        float* outp = out + n * sN + c * sC + lane * sM;
        for (int s = 0; s < kFramesPerIter; ++s)
          if (tf + s < F) outp[(tf + s) * sT] = db[s];
      }
    }
  }
  return 0;
}

// The fused FOA pass (csrc/logmel.hip logmel_iv_kernel): four "wavefronts" = channels W, X, Y, Z of one clip in lock step,
// their tiles contiguous as in the kernel's LDS, W's spectrum buffer behind them.  Every LDS index the per-lane functions
// form is kept honest by guard cells either side of the block (writes) and by the comparison with the oracle (reads).
template <typename T>
int run_iv(const T* pcm, int64_t N, int64_t L, float* out, const float* fb_opt) {
  if (L <= kNfft / 2) return -1;
  HostTables t;
  hann_window(t.window);
  stage_twiddles(t.twiddle);
  if (fb_opt)
    t.fb.assign(fb_opt, fb_opt + static_cast<size_t>(kBins) * kMels);
  else
    default_mel_filterbank(t.fb);
  if (build_sparse_mel(t.fb, t.b0, t.wd, t.wu)) return -4;
  if (place_power_rows(t.fb, t.b0, t.pos)) return -5;
  const LogmelTables tab = t.view();
  const long F = 1 + L / kHop;
  const long total_ch = 2 * kIvChannels - 1;
  const long sM = 1, sC = kMels, sT = total_ch * kMels, sN = F * total_ch * kMels;
  constexpr int kGuard = 64;
  const float kGuardValue = -12345.0f;
  std::vector<float> block(kGuard + kIvChannels * kLdsFloatsPerWave + kIvSpecFloats + kGuard, 0.0f);
  for (int i = 0; i < kGuard; ++i) block[i] = block[block.size() - 1 - i] = kGuardValue;
  float* tiles = block.data() + kGuard;
  float* wspec = tiles + kIvChannels * kLdsFloatsPerWave;
  std::vector<float> tab_lds(kTabFloats);
  for (int e = 0; e < kTabFloats; ++e) tab_lds[e] = table_value(tab, e);
  std::vector<cf> z(kIvChannels * 64 * kN2), m(kIvChannels * 64 * 16), xa(kIvChannels * 64 * 16), xb(kIvChannels * 64 * 16);
  std::vector<float> smp(64 * 48), ia(kIvChannels * 64 * 16), ib(kIvChannels * 64 * 16);
  std::vector<LaneAcc> acc(64);
  const long iters_per_row = (F + kFramesPerIter - 1) / kFramesPerIter;
#define AT(vec, w, lane, n) (*reinterpret_cast<decltype(vec)::value_type(*)[n]>(&vec[((w) * 64 + (lane)) * (n)]))
  for (long clip = 0; clip < N; ++clip) {
    for (long itr = 0; itr < iters_per_row; ++itr) {
      const long tf = itr * kFramesPerIter;
      const bool interior = (tf >= 1) && (static_cast<long>(kHop) * (tf + kFramesPerIter) <= L);
      for (int w = 0; w < kIvChannels; ++w) {
        float* lds = tiles + w * kLdsFloatsPerWave;
        const T* rowp = pcm + (clip * kIvChannels + w) * L;
        for (int lane = 0; lane < 64; ++lane) {
          float(&s)[48] = *reinterpret_cast<float(*)[48]>(&smp[lane * 48]);
          const long fa = tf + 2 * (lane >> 5);
          if (interior) load_samples<T, true>(lane, rowp, L, fa, s);
          else load_samples<T, false>(lane, rowp, L, fa, s);
        }
        for (int lane = 0; lane < 64; ++lane) {
          LaneConsts consts;
          load_lane_consts(lane, tab_lds.data(), consts);
          phase_a(lane, *reinterpret_cast<float(*)[48]>(&smp[lane * 48]), consts, lds);
        }
        for (int lane = 0; lane < 64; ++lane) phase_b(lane, lds, AT(z, w, lane, kN2));
        for (int lane = 0; lane < 64; ++lane) phase_b_store(lane, lds, AT(z, w, lane, kN2));
        for (int lane = 0; lane < 64; ++lane) phase_c_load(lane, lds, AT(m, w, lane, 16));
        for (int lane = 0; lane < 64; ++lane) {
          float* pp[16];
          power_row_pointers(lane, lds, t.pos.data(), pp);
          phase_c_unpack(lane, pp, AT(z, w, lane, kN2), AT(m, w, lane, 16), AT(xa, w, lane, 16), AT(xb, w, lane, 16));
        }
        if (w == 0)
          for (int lane = 0; lane < 64; ++lane) iv_publish(lane, wspec, AT(xa, 0, lane, 16), AT(xb, 0, lane, 16));
      }
      // ---- workgroup barrier 1
      for (int w = 1; w < kIvChannels; ++w) {
        float* lds = tiles + w * kLdsFloatsPerWave;
        for (int lane = 0; lane < 64; ++lane) {
          float* pp[16];
          power_row_pointers(lane, lds, t.pos.data(), pp);
          iv_compute(lane, w, wspec, pp, AT(xa, w, lane, 16), AT(xb, w, lane, 16), 1e-8f, AT(ia, w, lane, 16), AT(ib, w, lane, 16));
        }
      }
      for (int w = 0; w < kIvChannels; ++w) {
        float* lds = tiles + w * kLdsFloatsPerWave;
        for (int lane = 0; lane < 64; ++lane) phase_d_accumulate(lane, lds, tab_lds.data(), t.pos[lane], acc[lane]);
        for (int lane = 0; lane < 64; ++lane) {
          float db[kFramesPerIter], below[kFramesPerIter];
          for (int s = 0; s < kFramesPerIter; ++s) below[s] = lane > 0 ? acc[lane - 1].ab[s].y : 0.0f;
          phase_d_finish(acc[lane], below, db);
          float* outp = out + clip * sN + w * sC + lane * sM;
          for (int s = 0; s < kFramesPerIter; ++s)
            if (tf + s < F) outp[(tf + s) * sT] = db[s];
        }
      }
      // ---- workgroup barrier 2
      for (int w = 1; w < kIvChannels; ++w) {
        float* lds = tiles + w * kLdsFloatsPerWave;
        for (int lane = 0; lane < 64; ++lane) {
          float* pp[16];
          power_row_pointers(lane, lds, t.pos.data(), pp);
          iv_store_rows(lane, pp, AT(ia, w, lane, 16), AT(ib, w, lane, 16));
        }
        for (int lane = 0; lane < 64; ++lane) phase_d_accumulate(lane, lds, tab_lds.data(), t.pos[lane], acc[lane]);
        for (int lane = 0; lane < 64; ++lane) {
          float* outp = out + clip * sN + (kIvChannels - 1 + w) * sC + lane * sM;
          for (int s = 0; s < kFramesPerIter; ++s)
            if (tf + s < F) outp[(tf + s) * sT] = acc[lane].ab[s].x + (lane > 0 ? acc[lane - 1].ab[s].y : 0.0f);
        }
      }
    }
  }
#undef AT
  for (int i = 0; i < kGuard; ++i)
    if (block[i] != kGuardValue || block[block.size() - 1 - i] != kGuardValue) return -9;     // an LDS index left the block
  return 0;
}

}  // namespace

extern "C" {
int emu_logmel_f32(const float* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, const float* fb) {
  return run<float>(pcm, N, C, L, out, layout, fb);
}
int emu_logmel_i16(const int16_t* pcm, int64_t N, int64_t C, int64_t L, float* out, int layout, const float* fb) {
  return run<int16_t>(pcm, N, C, L, out, layout, fb);
}
// pcm [N][4][L] -> out [N][F][7][64]: log-mel of W, X, Y, Z and the three intensity-vector channels (fused FOA pass)
int emu_logmel_iv_f32(const float* pcm, int64_t N, int64_t L, float* out, const float* fb) {
  return run_iv<float>(pcm, N, L, out, fb);
}
int emu_logmel_iv_i16(const int16_t* pcm, int64_t N, int64_t L, float* out, const float* fb) {
  return run_iv<int16_t>(pcm, N, L, out, fb);
}
}
