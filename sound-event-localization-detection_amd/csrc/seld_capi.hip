// libseld_hip.so -- library state, constant tables, error reporting.  C ABI: include/seld_hip.h
#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "seld_common.h"
#include "logmel_tables.h"

namespace seld {

namespace {
thread_local std::string g_last_error;
std::mutex g_mutex;
std::map<int, DeviceState> g_states;
}  // namespace

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

DeviceState* current_state() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("hipGetDevice failed (no HIP device visible)");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = g_states.find(dev);
  if (it == g_states.end() || !it->second.ready) {
    set_error("seld_init(device) has not been called for the current device");
    return nullptr;
  }
  return &it->second;
}

static int upload_mel(DeviceState& st, const std::vector<float>& fb) {
  std::vector<int> b0;
  std::vector<float> wd, wu;
  const char* why = build_sparse_mel(fb, b0, wd, wu);
  if (why) return fail(kErrUnsupported, why);
  std::vector<int> pos;
  why = place_power_rows(fb, b0, pos);
  if (why) return fail(kErrUnsupported, why);
  SELD_HIP_TRY(hipMemcpy(st.mel_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
  SELD_HIP_TRY(hipMemcpy(st.mel_b0, b0.data(), b0.size() * sizeof(int), hipMemcpyHostToDevice));
  SELD_HIP_TRY(hipMemcpy(st.mel_wd, wd.data(), wd.size() * sizeof(float), hipMemcpyHostToDevice));
  SELD_HIP_TRY(hipMemcpy(st.mel_wu, wu.data(), wu.size() * sizeof(float), hipMemcpyHostToDevice));
  SELD_HIP_TRY(hipMemcpy(st.mel_fb, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  return kOk;
}

}  // namespace seld

extern "C" {

const char* seld_last_error(void) { return seld::g_last_error.c_str(); }

int seld_version(void) { return 100; }

int seld_init(int device) {
  using namespace seld;
  SELD_HIP_TRY(hipSetDevice(device));
  std::lock_guard<std::mutex> lock(g_mutex);
  DeviceState& st = g_states[device];
  if (st.ready) return kOk;
  st.device = device;
  hipDeviceProp_t prop;
  SELD_HIP_TRY(hipGetDeviceProperties(&prop, device));
  st.num_cus = prop.multiProcessorCount;

  std::vector<float> window, twiddle;
  hann_window(window);
  stage_twiddles(twiddle);
  SELD_HIP_TRY(hipMalloc(&st.window, window.size() * sizeof(float)));
  SELD_HIP_TRY(hipMalloc(&st.twiddle, twiddle.size() * sizeof(float)));
  SELD_HIP_TRY(hipMalloc(&st.mel_b0, kMels * sizeof(int)));
  SELD_HIP_TRY(hipMalloc(&st.mel_pos, kMelPosInts * sizeof(int)));
  SELD_HIP_TRY(hipMalloc(&st.mel_wd, kMels * kMelMaxCnt * sizeof(float)));
  SELD_HIP_TRY(hipMalloc(&st.mel_wu, kMels * kMelMaxCnt * sizeof(float)));
  SELD_HIP_TRY(hipMalloc(&st.mel_fb, static_cast<size_t>(kBins) * kMels * sizeof(float)));
  SELD_HIP_TRY(hipMemcpy(st.window, window.data(), window.size() * sizeof(float), hipMemcpyHostToDevice));
  SELD_HIP_TRY(hipMemcpy(st.twiddle, twiddle.data(), twiddle.size() * sizeof(float), hipMemcpyHostToDevice));
  std::vector<float> fb;
  default_mel_filterbank(fb);
  const int rc = upload_mel(st, fb);
  if (rc != kOk) return rc;
  const int rc_gcc = build_gcc_table(&st);
  if (rc_gcc != kOk) return rc_gcc;
  SELD_HIP_TRY(hipStreamCreateWithFlags(&st.side_stream, hipStreamNonBlocking));
  SELD_HIP_TRY(hipEventCreateWithFlags(&st.fork_event, hipEventDisableTiming));
  SELD_HIP_TRY(hipEventCreateWithFlags(&st.join_event, hipEventDisableTiming));
  st.ready = true;
  return kOk;
}

int seld_shutdown(void) {
  using namespace seld;
  std::lock_guard<std::mutex> lock(g_mutex);
  for (auto& kv : g_states) {
    DeviceState& st = kv.second;
    if (!st.ready) continue;
    (void)hipSetDevice(st.device);
    (void)hipFree(st.window);
    (void)hipFree(st.twiddle);
    (void)hipFree(st.mel_b0);
    (void)hipFree(st.mel_pos);
    (void)hipFree(st.mel_wd);
    (void)hipFree(st.mel_wu);
    (void)hipFree(st.mel_fb);
    if (st.gcc_table) (void)hipFree(st.gcc_table);
    (void)hipEventDestroy(st.fork_event);
    (void)hipEventDestroy(st.join_event);
    (void)hipStreamDestroy(st.side_stream);
    st = DeviceState();
  }
  g_states.clear();
  return kOk;
}

int seld_set_mel_filterbank(const float* fb_host) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!fb_host) return fail(kErrInvalidArgument, "seld_set_mel_filterbank: null table");
  std::vector<float> fb(fb_host, fb_host + static_cast<size_t>(kBins) * kMels);
  return upload_mel(*st, fb);
}

int seld_set_window(const float* window_host) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (!window_host) return fail(kErrInvalidArgument, "seld_set_window: null table");
  SELD_HIP_TRY(hipMemcpy(st->window, window_host, kNfft * sizeof(float), hipMemcpyHostToDevice));
  return kOk;
}

// Host-side copies of the constant tables (no GPU needed): lets CPU tests check the C++
// default filterbank / sparse decomposition against the oracle.
int seld_default_tables(float* window960, float* fb481x64, int* mel_b0_64, float* mel_wd_24x64,
                        float* mel_wu_24x64) {
  using namespace seld;
  std::vector<float> fb;
  default_mel_filterbank(fb);
  if (window960) {
    std::vector<float> w;
    hann_window(w);
    memcpy(window960, w.data(), w.size() * sizeof(float));
  }
  if (fb481x64) memcpy(fb481x64, fb.data(), fb.size() * sizeof(float));
  std::vector<int> b0;
  std::vector<float> wd, wu;
  const char* why = build_sparse_mel(fb, b0, wd, wu);
  if (why) return fail(kErrUnsupported, why);
  if (mel_b0_64) memcpy(mel_b0_64, b0.data(), b0.size() * sizeof(int));
  if (mel_wd_24x64) memcpy(mel_wd_24x64, wd.data(), wd.size() * sizeof(float));
  if (mel_wu_24x64) memcpy(mel_wu_24x64, wu.data(), wu.size() * sizeof(float));
  return kOk;
}

}  // extern "C"
