// The three-term SMR-SELD loss of smrl_seld_gaussian.py:946-1072 (= loss.py:43-54, 56-146 applied to probabilities),
// value and gradient in ONE pass over the logits, for gfx950:
//
//   total = w_class * MSE(softmax(z), y) + w_aiur * AIUR + w_cl * CL
//
//   MSE  : mean over all (cell, class) of (p - y)^2                                    smrl_seld_gaussian.py:959-962
//   AIUR : 1 - mean over frames of |pred events AND true events| / |pred OR true|, events = argmax != background,
//          frames without any event on either side count as IoU 1; argmax based, no gradient       :964-997
//   CL   : sum over frames WITH events and cells of  pred_nonbg * y_at  / (frames_with_events * I * J + 1e-10),
//          pred_nonbg = 1 - p_background, y' = 1 on background cells and -N_bac / (N_non + 1e-10) on event cells,
//          y_at = y' + mean over the 8 circular neighbours of (y'_nb - y') = the MEAN OF THE 8 NEIGHBOURS' y'  :999-1055
//
// Two kernels.  `cl_prepare_kernel` reads only the labels (one block per frame): event flags, N_non / N_bac, the
// attention map y_at (zero on frames without events), the frame's true-event count and the number of frames with events.
// `smr_loss_kernel` is the softmax-MSE kernel of loss.hip (one cell per thread, 256-cell tiles staged through LDS) with
// two additions per cell: the CL term and its gradient through the softmax,
//     d CL / d z_k = c * p_k * p_bg (k != bg),   d CL / d z_bg = -c * p_bg * (1 - p_bg),   c = w_cl * y_at / denom,
// and the predicted-event flag, counted per frame with one integer atomic per (block, frame) -- integer sums are
// order independent, so the result is deterministic; the value sums use fixed-order double partials like loss.hip.
// HBM-bound like loss.hip: + 4 B per cell of attention map.
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

namespace {

constexpr int kL3Block = 256;
constexpr int kL3M = 14;
constexpr int kL3MaxCells = 1024;          // grid cells per frame the prepare kernel holds in LDS (18 x 36 = 648)

struct Loss3Args {
  const void* logits;          // [n_cells][14] fp32 / bf16
  const uint16_t* mask;        // [n_cells] or null
  const float* dense;          // [n_cells][14] or null
  const float* att;            // [n_cells] attention map (0 on frames without events)
  const int* frames_with_events;   // device scalar
  int* frame_counts;           // [frames][2]: predicted events, intersection with true events (zeroed by prepare)
  int n_cells, cells_per_frame;
  float mse_grad_scale;        // w_class * 2 / (n_cells * 14), or 0 with grad == null
  float w_cl;
  double* partials;            // [2][blocks]: squared error, CL numerator
  void* grad;                  // like logits, or null
};

__device__ __forceinline__ unsigned short f2bf(float f) { return __bfloat16_as_ushort(__float2bfloat16(f)); }

template <bool kMask>
__global__ __launch_bounds__(kL3Block) void cl_prepare_kernel(const uint16_t* __restrict__ mask,
                                                              const float* __restrict__ dense, int rows, int cols,
                                                              float* __restrict__ att, int* __restrict__ true_count,
                                                              int* __restrict__ frame_counts,
                                                              int* __restrict__ frames_with_events) {
  __shared__ float yprime[kL3MaxCells];
  __shared__ int counts[3];                 // event cells (CL), background cells (CL), argmax events (AIUR)
  const int frame = blockIdx.x, cells = rows * cols, tid = threadIdx.x;
  if (tid < 3) counts[tid] = 0;
  __syncthreads();
  int n_non = 0, n_bac = 0, n_arg = 0;
  for (int cell = tid; cell < cells; cell += kL3Block) {
    float nonbg;
    bool arg_event;
    if (kMask) {
      const unsigned m = mask[static_cast<long>(frame) * cells + cell];
      nonbg = static_cast<float>(__popc(m & 0x1fffu));          // sum of the 13 event classes of the multi-hot row
      arg_event = (m & 0x1fffu) != 0u;                          // first maximum of the row is an event class
    } else {
      const float* y = dense + (static_cast<long>(frame) * cells + cell) * kL3M;
      nonbg = 0.0f;
      int best = 0;
      float top = y[0];
#pragma unroll
      for (int c = 0; c < kL3M; ++c) {
        const float v = y[c];
        if (c < kL3M - 1) nonbg += v;
        if (v > top) { top = v; best = c; }
      }
      arg_event = best != kL3M - 1;
    }
    const bool event = nonbg > 0.01f;
    n_non += event ? 1 : 0;
    n_bac += nonbg < 0.01f ? 1 : 0;
    n_arg += arg_event ? 1 : 0;
    yprime[cell] = event ? -1.0f : 1.0f;                        // event cells get their value after the counts
  }
  atomicAdd(&counts[0], n_non);
  atomicAdd(&counts[1], n_bac);
  atomicAdd(&counts[2], n_arg);
  __syncthreads();
  const int non = counts[0];
  const float ratio = -(static_cast<float>(counts[1]) / (static_cast<float>(non) + 1e-10f));
  for (int cell = tid; cell < cells; cell += kL3Block)
    if (yprime[cell] < 0.0f) yprime[cell] = ratio;
  __syncthreads();
  for (int cell = tid; cell < cells; cell += kL3Block) {
    const int i = cell / cols, j = cell - i * cols;
    float sum = 0.0f;
    // the reference's order of the eight additions (di outer, dj inner), each term (neighbour - centre)
    const float centre = yprime[cell];
#pragma unroll
    for (int di = -1; di <= 1; ++di)
#pragma unroll
      for (int dj = -1; dj <= 1; ++dj) {
        if (di == 0 && dj == 0) continue;
        const int ii = (i + di + rows) % rows, jj = (j + dj + cols) % cols;
        sum += yprime[ii * cols + jj] - centre;
      }
    att[static_cast<long>(frame) * cells + cell] = non > 0 ? centre + sum / 8.0f : 0.0f;
  }
  if (tid == 0) {
    true_count[frame] = counts[2];
    frame_counts[2 * frame] = 0;
    frame_counts[2 * frame + 1] = 0;
    if (non > 0) atomicAdd(frames_with_events, 1);
  }
}

template <bool kBf16, bool kMask, bool kGrad>
__global__ __launch_bounds__(kL3Block) void smr_loss_kernel(const Loss3Args a) {
  __shared__ __attribute__((aligned(16))) float tile[kL3Block * kL3M];
  __shared__ double wave_sums[2][kL3Block / 64];
  __shared__ int seg_counts[2][2];           // [frame segment of the tile][pred, inter]
  const int tid = threadIdx.x;
  double local_sq = 0.0, local_cl = 0.0;
  const int n_tiles = (a.n_cells + kL3Block - 1) / kL3Block;
  const float cl_scale = a.w_cl / (static_cast<float>(*a.frames_with_events) * static_cast<float>(a.cells_per_frame) + 1e-10f);

  for (int tileno = blockIdx.x; tileno < n_tiles; tileno += gridDim.x) {
    const int cell0 = tileno * kL3Block;
    const int cells_here = min(a.n_cells - cell0, kL3Block);
    const int elems = cells_here * kL3M;
    if (tid < 4) seg_counts[tid >> 1][tid & 1] = 0;
    if (kBf16) {
      const unsigned short* src = static_cast<const unsigned short*>(a.logits) + static_cast<long>(cell0) * kL3M;
      for (int i = tid * 2; i < elems; i += kL3Block * 2) {     // elems is even (14 classes)
        const unsigned v = *reinterpret_cast<const unsigned*>(src + i);
        tile[i] = __uint_as_float(v << 16);
        tile[i + 1] = __uint_as_float(v & 0xffff0000u);
      }
    } else {
      const float* src = static_cast<const float*>(a.logits) + static_cast<long>(cell0) * kL3M;
      for (int i = tid; i < elems; i += kL3Block) tile[i] = src[i];
    }
    __syncthreads();

    const int first_frame = cell0 / a.cells_per_frame;
    float g[kL3M];
    if (tid < cells_here) {
      const int cell = cell0 + tid;
      float z[kL3M], y[kL3M];
#pragma unroll
      for (int c = 0; c < kL3M; c += 2) {
        const float2 v = *reinterpret_cast<const float2*>(tile + tid * kL3M + c);
        z[c] = v.x;
        z[c + 1] = v.y;
      }
      bool true_event;
      if (kMask) {
        const unsigned m = a.mask[cell];
#pragma unroll
        for (int c = 0; c < kL3M; ++c) y[c] = ((m >> c) & 1u) ? 1.0f : 0.0f;
        if (m == 0u) y[kL3M - 1] = 1.0f;
        true_event = (m & 0x1fffu) != 0u;
      } else {
        const float* yp = a.dense + static_cast<long>(cell) * kL3M;
        int best = 0;
#pragma unroll
        for (int c = 0; c < kL3M; ++c) {
          y[c] = yp[c];
          if (y[c] > y[best]) best = c;
        }
        true_event = best != kL3M - 1;
      }
      float zmax = z[0];
#pragma unroll
      for (int c = 1; c < kL3M; ++c) zmax = fmaxf(zmax, z[c]);
      float denom = 0.0f;
#pragma unroll
      for (int c = 0; c < kL3M; ++c) {
        z[c] = __expf(z[c] - zmax);
        denom += z[c];
      }
      const float inv = 1.0f / denom;
      float sq = 0.0f, dot = 0.0f;
      int best = 0;
      float top = -1.0f;
#pragma unroll
      for (int c = 0; c < kL3M; ++c) {
        const float p = z[c] * inv;
        if (p > top) { top = p; best = c; }                      // first maximum, like torch.argmax on the probabilities
        const float d = p - y[c];
        sq = fmaf(d, d, sq);
        dot = fmaf(p, d, dot);
        z[c] = p;
        y[c] = d;
      }
      local_sq += static_cast<double>(sq);
      const float p_bg = z[kL3M - 1];
      const float w = a.att[cell];
      local_cl += static_cast<double>((1.0f - p_bg) * w);
      if (best != kL3M - 1) {                                    // a predicted event: per-frame counts for the AIUR
        const int frame = cell / a.cells_per_frame;
        if (a.cells_per_frame >= kL3Block) {                     // a 256-cell tile then spans <= 2 frames: count in LDS
          atomicAdd(&seg_counts[frame - first_frame][0], 1);
          if (true_event) atomicAdd(&seg_counts[frame - first_frame][1], 1);
        } else {                                                 // toy grids: straight to the frame's counters
          atomicAdd(&a.frame_counts[2 * frame], 1);
          if (true_event) atomicAdd(&a.frame_counts[2 * frame + 1], 1);
        }
      }
      if (kGrad) {
        const float c_cl = cl_scale * w * p_bg;
#pragma unroll
        for (int c = 0; c < kL3M - 1; ++c) g[c] = fmaf(a.mse_grad_scale * z[c], y[c] - dot, c_cl * z[c]);
        g[kL3M - 1] = fmaf(a.mse_grad_scale * p_bg, y[kL3M - 1] - dot, -c_cl * (1.0f - p_bg));
      }
    }
    __syncthreads();                                             // logits consumed, segment counts complete
    if (tid < 4) {
      const int v = seg_counts[tid >> 1][tid & 1];
      const int frame = first_frame + (tid >> 1);
      if (v != 0 && static_cast<long>(frame) * a.cells_per_frame < a.n_cells) atomicAdd(&a.frame_counts[2 * frame + (tid & 1)], v);
    }
    if (kGrad) {
      if (tid < cells_here) {
#pragma unroll
        for (int c = 0; c < kL3M; c += 2) *reinterpret_cast<float2*>(tile + tid * kL3M + c) = make_float2(g[c], g[c + 1]);
      }
      __syncthreads();
      if (kBf16) {
        unsigned short* dst = static_cast<unsigned short*>(a.grad) + static_cast<long>(cell0) * kL3M;
        for (int i = tid * 2; i < elems; i += kL3Block * 2)
          *reinterpret_cast<unsigned*>(dst + i) = f2bf(tile[i]) | (static_cast<unsigned>(f2bf(tile[i + 1])) << 16);
      } else {
        float* dst = static_cast<float*>(a.grad) + static_cast<long>(cell0) * kL3M;
        for (int i = tid; i < elems; i += kL3Block) dst[i] = tile[i];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    local_sq += __shfl_down(local_sq, off, 64);
    local_cl += __shfl_down(local_cl, off, 64);
  }
  if ((tid & 63) == 0) {
    wave_sums[0][tid >> 6] = local_sq;
    wave_sums[1][tid >> 6] = local_cl;
  }
  __syncthreads();
  if (tid < 2) {
    double s = 0.0;
    for (int w = 0; w < kL3Block / 64; ++w) s += wave_sums[tid][w];
    a.partials[tid * gridDim.x + blockIdx.x] = s;
  }
}

// out = (total, mse, aiur, cl)
__global__ __launch_bounds__(256) void smr_loss_finish_kernel(const double* __restrict__ partials, int blocks,
                                                              const int* __restrict__ frame_counts,
                                                              const int* __restrict__ true_count, int frames,
                                                              const int* __restrict__ frames_with_events,
                                                              int cells_per_frame, double inv_elements, float w_class,
                                                              float w_aiur, float w_cl, float* __restrict__ out) {
  __shared__ double part[3][256];
  double sq = 0.0, cl = 0.0, iou = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) {
    sq += partials[i];
    cl += partials[blocks + i];
  }
  for (int f = threadIdx.x; f < frames; f += 256) {
    const float inter = static_cast<float>(frame_counts[2 * f + 1]);
    const float uni = static_cast<float>(frame_counts[2 * f]) + static_cast<float>(true_count[f]) - inter;
    iou += uni > 0.0f ? static_cast<double>(inter / (uni + 1e-8f)) : 1.0;       // fp32 quotient like the reference
  }
  part[0][threadIdx.x] = sq;
  part[1][threadIdx.x] = cl;
  part[2][threadIdx.x] = iou;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (static_cast<int>(threadIdx.x) < off)
      for (int k = 0; k < 3; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mse = static_cast<float>(part[0][0] * inv_elements);
    const float aiur = 1.0f - static_cast<float>(part[2][0] / frames);
    const float denom = static_cast<float>(*frames_with_events) * static_cast<float>(cells_per_frame) + 1e-10f;
    const float clv = static_cast<float>(part[1][0]) / denom;
    out[0] = w_class * mse + w_aiur * aiur + w_cl * clv;
    out[1] = mse;
    out[2] = aiur;
    out[3] = clv;
  }
}

}  // namespace

}  // namespace seld

extern "C" {

int64_t seld_smr_loss_workspace_bytes(int64_t frames, int64_t cells_per_frame) {
  if (frames <= 0 || cells_per_frame <= 0) return 0;
  // attention map (float per cell) | true counts (int per frame) | frame counts (2 ints per frame) | frames with events
  // (1 int, padded) | partial sums (2 x 4096 doubles)
  return frames * cells_per_frame * 4 + frames * 4 + frames * 8 + 16 + 2 * 4096 * 8 + 64;
}

int seld_smr_loss(const void* logits, int logits_is_bf16, const uint16_t* mask, const float* dense_labels,
                  int64_t frames, int rows, int cols, int num_classes, float w_class, float w_aiur, float w_cl,
                  float* loss_out4, void* grad, void* workspace, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (num_classes != kL3M) return fail(kErrUnsupported, "seld_smr_loss: built for 14 classes (config.py:40)");
  const int64_t cells = static_cast<int64_t>(rows) * cols;
  if (frames <= 0 || rows < 3 || cols < 3 || cells > kL3MaxCells)
    return fail(kErrInvalidArgument, "seld_smr_loss: frames > 0, a grid of at least 3 x 3 and at most 1024 cells");
  const int64_t n_cells = frames * cells;
  if (n_cells > (2147483647LL - 4096) / num_classes)
    return fail(kErrUnsupported, "seld_smr_loss: cells * classes must stay below 2^31 (32-bit tile indexing)");
  if (!logits || !loss_out4 || !workspace) return fail(kErrInvalidArgument, "seld_smr_loss: null pointer");
  if ((mask == nullptr) == (dense_labels == nullptr))
    return fail(kErrInvalidArgument, "seld_smr_loss: pass exactly one of mask / dense_labels");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  float* att = reinterpret_cast<float*>(ws);
  int* true_count = reinterpret_cast<int*>(ws + n_cells * 4);
  int* frame_counts = true_count + frames;
  int* n_event_frames = frame_counts + 2 * frames;
  size_t off = static_cast<size_t>(n_cells) * 4 + static_cast<size_t>(frames) * 12 + 16;
  off = (off + 63) / 64 * 64;
  double* partials = reinterpret_cast<double*>(ws + off);
  SELD_HIP_TRY(hipMemsetAsync(n_event_frames, 0, sizeof(int), stream));
  if (mask) hipLaunchKernelGGL(cl_prepare_kernel<true>, dim3(static_cast<unsigned>(frames)), dim3(kL3Block), 0, stream,
                               mask, dense_labels, rows, cols, att, true_count, frame_counts, n_event_frames);
  else hipLaunchKernelGGL(cl_prepare_kernel<false>, dim3(static_cast<unsigned>(frames)), dim3(kL3Block), 0, stream, mask,
                          dense_labels, rows, cols, att, true_count, frame_counts, n_event_frames);
  long tiles = (n_cells + kL3Block - 1) / kL3Block;
  long blocks = static_cast<long>(st->num_cus) * 8;
  if (blocks > tiles) blocks = tiles;
  if (blocks > 4096) blocks = 4096;
  Loss3Args a{logits, mask, dense_labels, att, n_event_frames, frame_counts, static_cast<int>(n_cells),
              static_cast<int>(cells), grad ? w_class * 2.0f / (static_cast<float>(n_cells) * kL3M) : 0.0f, w_cl,
              partials, grad};
  const bool bf = logits_is_bf16 != 0, mk = mask != nullptr, gr = grad != nullptr;
#define SELD_L3(B, K, G) \
  if (bf == B && mk == K && gr == G) \
    hipLaunchKernelGGL((smr_loss_kernel<B, K, G>), dim3(static_cast<unsigned>(blocks)), dim3(kL3Block), 0, stream, a)
  SELD_L3(false, false, false); SELD_L3(false, false, true); SELD_L3(false, true, false); SELD_L3(false, true, true);
  SELD_L3(true, false, false);  SELD_L3(true, false, true);  SELD_L3(true, true, false);  SELD_L3(true, true, true);
#undef SELD_L3
  hipLaunchKernelGGL(smr_loss_finish_kernel, dim3(1), dim3(256), 0, stream, partials, static_cast<int>(blocks),
                     frame_counts, true_count, static_cast<int>(frames), n_event_frames, static_cast<int>(cells),
                     1.0 / (static_cast<double>(n_cells) * kL3M), w_class, w_aiur, w_cl, loss_out4);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
