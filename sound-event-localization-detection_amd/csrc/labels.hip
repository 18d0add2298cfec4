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

// Gaussian-region label augmentation (smrl_seld_gaussian.py:397-534): every event paints the class bit into ALL
// grid cells whose centre lies inside the +-2 sigma box around (azimuth + az_noise, elevation + el_noise), with
// azimuth wrap-around and elevation clipped to [-90, 90].  The per-source noise is drawn on the host (one
// (az, el) normal pair per unique (class, source), :426-437) and arrives as the box centre of every row.
// One thread per (row, cell); float64 and the reference's comparison order, so the painted set is bit-exact.
__global__ void rasterise_box_kernel(const int32_t* __restrict__ ev, const double* __restrict__ centre, long R, long T,
                                     int I, int J, double two_sigma_az, double two_sigma_el,
                                     unsigned int* __restrict__ mask_words) {
  const long cells = static_cast<long>(I) * J;
  const long gid = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= R * cells) return;
  const long r = gid / cells;
  const int cell = static_cast<int>(gid - r * cells);
  const int32_t* row = ev + r * 5;
  const long meta_frame = row[0];
  const int cls = row[1];
  if (meta_frame < 0 || cls < 0 || cls > 15) return;
  const long start = meta_frame * kFramesPerMeta;
  long end = start + kFramesPerMeta;
  if (end > T) end = T;
  if (start >= end) return;
  const int gi = cell / J, gj = cell - gi * J;
  const double centre_az = centre[2 * r], centre_el = centre[2 * r + 1];
  double el_min = centre_el - two_sigma_el, el_max = centre_el + two_sigma_el;
  el_min = el_min > -90.0 ? el_min : -90.0;                 // max(elevation_min, -90)
  el_max = el_max < 90.0 ? el_max : 90.0;                   // min(elevation_max, 90)
  // __dmul_rn / __dadd_rn: separately rounded like CPython's float ops (no fma contraction)
  const double cell_el = __dadd_rn(-90.0, __dmul_rn(gi + 0.5, 180.0 / I));
  const double cell_az = __dadd_rn(-180.0, __dmul_rn(gj + 0.5, 360.0 / J));
  double diff = cell_az - centre_az;                        // normalize_azimuth_diff (:498-505)
  while (diff > 180.0) diff -= 360.0;
  while (diff < -180.0) diff += 360.0;
  const bool az_ok = fabs(diff) <= two_sigma_az;
  const bool el_ok = el_min <= cell_el && cell_el <= el_max;
  if (!(az_ok && el_ok)) return;
  for (long t = start; t < end; ++t) {
    const long idx = t * cells + cell;
    atomicOr(mask_words + (idx >> 1), (1u << cls) << ((idx & 1) * 16));
  }
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

int seld_labels_rasterise_box(const int32_t* events, const double* centres, int64_t R, int64_t T, int I, int J,
                              double sigma_az, double sigma_el, uint16_t* mask, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (T < 0 || R < 0 || I <= 0 || J <= 0) return fail(kErrInvalidArgument, "seld_labels_rasterise_box: bad extents");
  if ((static_cast<long>(I) * J) % 2 != 0)
    return fail(kErrUnsupported, "seld_labels_rasterise_box: I*J must be even");
  if (T == 0) return kOk;
  if (!mask || (R > 0 && (!events || !centres))) return fail(kErrInvalidArgument, "seld_labels_rasterise_box: null pointer");
  SELD_HIP_TRY(hipMemsetAsync(mask, 0, static_cast<size_t>(T) * I * J * sizeof(uint16_t), stream));
  if (R == 0) return kOk;
  const long threads = R * I * J;
  hipLaunchKernelGGL(rasterise_box_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream,
                     events, centres, static_cast<long>(R), static_cast<long>(T), I, J, 2.0 * sigma_az, 2.0 * sigma_el,
                     reinterpret_cast<unsigned int*>(mask));
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
