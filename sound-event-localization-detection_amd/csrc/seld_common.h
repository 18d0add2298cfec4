// Shared host-side plumbing for libseld_hip.so (error reporting, device state).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "logmel_core.h"

namespace seld {

enum : int {
  kOk = 0,
  kErrInvalidArgument = -1,
  kErrHip = -2,
  kErrNotInitialised = -3,
  kErrUnsupported = -4,
};

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define SELD_HIP_TRY(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return ::seld::fail(::seld::kErrHip, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct DeviceState {
  bool ready = false;
  int device = -1;
  int num_cus = 0;
  // log-mel tables (device memory)
  float* window = nullptr;
  float* twiddle = nullptr;
  int* mel_b0 = nullptr;
  float* mel_wd = nullptr;
  float* mel_wu = nullptr;
  int* mel_pos = nullptr;       // power-row placement (logmel_tables.h, place_power_rows)
  float* mel_fb = nullptr;      // dense [481][64] copy (used by the intensity-vector / debug paths)
  void* gcc_table = nullptr;    // fp16 cosine / sine B fragments of the matrix-core GCC-PHAT (spatial.hip), built on first use
  // side stream + fork/join events: the tiny edge kernel of the log-mel path overlaps the main kernel
  hipStream_t side_stream = nullptr;
  hipEvent_t fork_event = nullptr, join_event = nullptr;
  // kernels that need more dynamic LDS than the default limit: hipFuncSetAttribute is per DEVICE (and is not a stream
  // operation: done once so that launches stay graph-capturable); one bit per kernel family, see need_lds()
  unsigned lds_attr_done = 0;
  LogmelTables tables() const { return LogmelTables{window, twiddle, mel_b0, mel_wd, mel_wu, mel_pos}; }
};

enum LdsAttrBit : unsigned {
  kAttrGruForward = 1u << 0,
  kAttrGruBackward = 1u << 1,
  kAttrStftF32 = 1u << 2,
  kAttrStftI16 = 1u << 3,
  kAttrLogmelF32 = 1u << 4,
  kAttrLogmelI16 = 1u << 5,
  kAttrSpatial = 1u << 6,
  kAttrLogmelSpecF32 = 1u << 7,
  kAttrLogmelSpecI16 = 1u << 8,
  kAttrGccMfma = 1u << 9,
  kAttrLogmelPhasorF32 = 1u << 10,
  kAttrLogmelPhasorI16 = 1u << 11,
  kAttrGccMfmaQ15 = 1u << 12,
  kAttrLogmelIvF32 = 1u << 13,
  kAttrLogmelIvI16 = 1u << 14,
};

// true when `bit` still has to be set up on this device (the caller then sets its attributes and calls lds_attr_set)
inline bool need_lds(const DeviceState* st, unsigned bit) { return (st->lds_attr_done & bit) == 0; }
inline void lds_attr_set(DeviceState* st, unsigned bit) { st->lds_attr_done |= bit; }

// Returns the state of the CURRENT hip device, or nullptr (and sets the error) if seld_init
// has not been called for it.
DeviceState* current_state();

// spatial.hip: the constant fp16 cosine / sine table of the matrix-core GCC-PHAT kernel (96 KB, built by seld_init)
int build_gcc_table(DeviceState* st);

}  // namespace seld
