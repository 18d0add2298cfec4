// Label rasteriser and window gather for gfx950 -- integer / index work, bit-exact.
//
// Replaces the pure-Python loops of the reference:
//   dataset.py:60-119  metadata_to_labels   (iterrows + T x 648 background fill)
//   utils.py:77-90     polar_to_grid
//   dataset.py:267-317 _create_windows      (slice / pad of the concatenated timeline)
//
// Labels are kept COMPACT on the device: one uint16 per (frame, grid cell) whose bit c says
// "class c active in this cell" (the reference's labels[t, cell, c] = 1.0, dataset.py:110).
// The background one-hot (dataset.py:114-117) is implied by mask == 0 and materialised only
// by seld_labels_expand (or consumed directly by the fused loss): 2 B per cell instead of 56 B.
#include "seld_common.h"

namespace seld {

constexpr int kFramesPerMeta = 5;   // dataset.py:69-71: 100 ms metadata frame / 20 ms label frame

// utils.py:77-90 in float64, same operation order: normalise, scale, clip, truncate.
__device__ __forceinline__ int polar_cell(int az, int el, int I, int J) {
  const double phi_norm = (static_cast<double>(az) + 180.0) / 360.0;
  const double theta_norm = (static_cast<double>(el) + 90.0) / 180.0;
  double jf = phi_norm * static_cast<double>(J);
  double ifl = theta_norm * static_cast<double>(I);
  jf = jf < 0.0 ? 0.0 : (jf > static_cast<double>(J - 1) ? static_cast<double>(J - 1) : jf);
  ifl = ifl < 0.0 ? 0.0 : (ifl > static_cast<double>(I - 1) ? static_cast<double>(I - 1) : ifl);
  return static_cast<int>(ifl) * J + static_cast<int>(jf);
}

// One thread per (event row, sub-frame).  uint16 cells are OR-ed through the aligned 32-bit word.
__global__ void rasterise_kernel(const int32_t* __restrict__ ev, long R, long T, int I, int J,
                                 unsigned int* __restrict__ mask_words) {
  const long gid = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= R * kFramesPerMeta) return;
  const long r = gid / kFramesPerMeta;
  const int sub = static_cast<int>(gid - r * kFramesPerMeta);
  const int32_t* row = ev + r * 5;
  const long meta_frame = row[0];
  const int cls = row[1];
  if (meta_frame < 0 || cls < 0 || cls > 15) return;
  const long t = meta_frame * kFramesPerMeta + sub;         // dataset.py:100-103
  if (t >= T) return;                                       // rows past the end: silently dropped
  const long cell = polar_cell(row[3], row[4], I, J);
  const long idx = t * (static_cast<long>(I) * J) + cell;   // uint16 index
  const unsigned int bit = (1u << cls) << ((idx & 1) * 16);
  atomicOr(mask_words + (idx >> 1), bit);
}

// mask -> dense float32 [n][M]: bit c -> 1.0; background (class M-1) = 1.0 where mask == 0.
// Each thread writes one float4 (n*M is a multiple of 4 because the grid has an even cell count).
__global__ void expand_kernel(const uint16_t* __restrict__ mask, long n_cells, int M, float4* __restrict__ dense) {
  const long total4 = n_cells * M / 4;
  for (long q = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; q < total4;
       q += static_cast<long>(gridDim.x) * blockDim.x) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long idx = q * 4 + e;
      const long cell = idx / M;
      const int c = static_cast<int>(idx - cell * M);
      const unsigned m = mask[cell];
      const bool on = ((m >> c) & 1u) || (c == M - 1 && m == 0u);
      v[e] = on ? 1.0f : 0.0f;
    }
    dense[q] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// Row gather: dst[b][w][:] = src[starts[b] + w][:] for starts[b]+w < total_rows, else zeros.
// One 16-byte chunk per thread; rows are multiples of 16 bytes (1024 B spec rows, 1296 B mask rows).
__global__ void gather_rows_kernel(const uint4* __restrict__ src, long total_rows, long row_chunks,
                                   const int64_t* __restrict__ starts, long B, long window,
                                   uint4* __restrict__ dst) {
  const long total = B * window * row_chunks;
  for (long q = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x; q < total;
       q += static_cast<long>(gridDim.x) * blockDim.x) {
    const long chunk = q % row_chunks;
    const long rw = q / row_chunks;
    const long w = rw % window;
    const long b = rw / window;
    const long srow = starts[b] + w;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (srow >= 0 && srow < total_rows) v = src[srow * row_chunks + chunk];
    dst[q] = v;
  }
}

static unsigned grid_for(long work_items, int block, int num_cus) {
  long blocks = (work_items + block - 1) / block;
  const long cap = static_cast<long>(num_cus) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

}  // namespace seld

extern "C" {

int seld_labels_rasterise(const int32_t* events, int64_t R, int64_t T, int I, int J, uint16_t* mask, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (T < 0 || R < 0 || I <= 0 || J <= 0) return fail(kErrInvalidArgument, "seld_labels_rasterise: bad extents");
  if ((static_cast<long>(I) * J) % 2 != 0)
    return fail(kErrUnsupported, "seld_labels_rasterise: I*J must be even (uint16 cells are OR-ed as 32-bit words)");
  if (T == 0) return kOk;
  if (!mask || (R > 0 && !events)) return fail(kErrInvalidArgument, "seld_labels_rasterise: null pointer");
  SELD_HIP_TRY(hipMemsetAsync(mask, 0, static_cast<size_t>(T) * I * J * sizeof(uint16_t), stream));
  if (R == 0) return kOk;
  const long threads = R * kFramesPerMeta;
  const unsigned blocks = static_cast<unsigned>((threads + 255) / 256);
  hipLaunchKernelGGL(rasterise_kernel, dim3(blocks), dim3(256), 0, stream, events, static_cast<long>(R),
                     static_cast<long>(T), I, J, reinterpret_cast<unsigned int*>(mask));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_labels_expand(const uint16_t* mask, int64_t n_cells, int num_classes, float* dense, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (n_cells < 0 || num_classes < 1 || num_classes > 16)
    return fail(kErrInvalidArgument, "seld_labels_expand: bad extents");
  if (n_cells == 0) return kOk;
  if (!mask || !dense) return fail(kErrInvalidArgument, "seld_labels_expand: null pointer");
  if ((n_cells * num_classes) % 4 != 0)
    return fail(kErrUnsupported, "seld_labels_expand: n_cells*num_classes must be a multiple of 4");
  const unsigned blocks = grid_for(n_cells * num_classes / 4, 256, st->num_cus);
  hipLaunchKernelGGL(expand_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), mask,
                     static_cast<long>(n_cells), num_classes, reinterpret_cast<float4*>(dense));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_window_gather(const void* src, int64_t total_rows, int64_t row_bytes, const int64_t* starts, int64_t B,
                       int64_t window, void* dst, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (total_rows < 0 || row_bytes <= 0 || B < 0 || window <= 0)
    return fail(kErrInvalidArgument, "seld_window_gather: bad extents");
  if (row_bytes % 16 != 0) return fail(kErrUnsupported, "seld_window_gather: row_bytes must be a multiple of 16");
  if (B == 0) return kOk;
  if (!src || !starts || !dst) return fail(kErrInvalidArgument, "seld_window_gather: null pointer");
  const long chunks = row_bytes / 16;
  const unsigned blocks = grid_for(B * window * chunks, 256, st->num_cus);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     static_cast<const uint4*>(src), static_cast<long>(total_rows), chunks, starts,
                     static_cast<long>(B), static_cast<long>(window), static_cast<uint4*>(dst));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
