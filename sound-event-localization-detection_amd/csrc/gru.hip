// Persistent bidirectional GRU recurrence for gfx950 (forward and backward).
//
// Replaces the recurrent part of nn.GRU(2048 -> 256, 2 layers, bidirectional) at
// model_crnn.py:65-72 / :119.  The input projection  gi = x W_ih^T + b_ih  for all time steps is
// one large GEMM done by the host (hipBLASLt, MFMA); what remains is 250 strictly sequential
// steps per direction of
//     gh = h W_hh^T + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ; n = tanh(gi_n + r*gh_n)
//     h' = (1 - z) n + z h
// which a library executes as ~2 tiny kernels per step.  Here ONE workgroup per (direction,
// 8-sequence batch tile) runs all T steps with W_hh resident on the CU:
//   * 8 wavefronts; wavefront w owns hidden units [32w, 32w+32) for all three gates;
//   * W_hh (bf16, 393 KB per direction) does not fit one place: the r and z gate rows live in
//     VGPRs as MFMA A-fragments (128 registers per lane), the n gate rows in LDS (128 KB) laid
//     out fragment-major so every ds_read_b128 is a linear conflict-free 1 KB read;
//   * h_{t-1} (bf16) is exchanged through a double-buffered 16 x 256 LDS tile (row pitch 528 B:
//     conflict-free B-fragment reads), one barrier per step;
//   * v_mfma_f32_16x16x32_bf16: per step 48 MFMAs per wavefront, fp32 accumulate, fp32 gates.
//
// WHY ONLY 8 SEQUENCES PER 16-COLUMN MFMA TILE.  With the weights resident, a step is bound by what ONE CU
// can do besides the MFMAs: its vector-memory path sustains only ~10-15 B/clk (MI355X_MICROARCH.md: "~10
// B/cyc/CU"), and the gate math (6 transcendentals per element) issues on the same 4 SIMDs.  Both scale
// with sequences per CU while the MFMA time (96 per SIMD, ~1.5k cycles) does not -- so the batch is spread over
// TWICE the CUs: columns 8..15 of every MFMA are padding (zero h rows), and after the MFMAs lane (q, c >= 8)
// takes over the second 16-unit tile of lane (q, c - 8) with one DPP row-shift per value.  Every lane then
// owns ONE (sequence, 4 units) group: half the gate instructions and half the bytes per CU per step, all 64
// lanes active in every load and store.  Measured (B = 32, T = 250): see DESIGN.md section 5.4.
//
// The backward kernel mirrors it: dgh (bf16) goes through LDS as the B operand, W_hh^T fragments
// are register / LDS resident, dh is carried in registers; it emits the per-step gate gradients
// from which the host forms dW_ih, dW_hh, dx with three large GEMMs.
#include <hip/hip_bf16.h>

#include "seld_common.h"

namespace seld {

constexpr int kH = 256;            // hidden size (config.py:45 CRNN_RNN_HIDDEN)
constexpr int kG = 3 * kH;         // gate rows r | z | n
constexpr int kRows = 16;          // MFMA N (columns); rows of the LDS exchange tiles
constexpr int kSeqs = 8;           // sequences per workgroup (columns 0..7; 8..15 are padding)
constexpr int kGruThreads = 512;   // 8 wavefronts
constexpr int kHPitch = kH + 8;    // bf16 elements per h row in LDS (528 B)
constexpr int kDghPitch = kG + 8;  // bf16 elements per dgh row in LDS (1552 B)
constexpr int kWnBytes = 8 * 2 * 8 * 64 * 16;   // [wave][tile][kstep][lane] x 16 B = 131072

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(__hip_bfloat16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ __hip_bfloat16 from_float<__hip_bfloat16>(float v) { return __float2bfloat16(v); }

// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the gates are fp32 but not bit-critical
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(x) = 1 - 2 / (1 + e^{2x}): saturates correctly (e = inf -> 1, e = 0 -> -1) without a clamp
__device__ __forceinline__ float tanh_f(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f); }

// Lanes 8..15 of every 16-lane row take `hi` from the lane 8 to their left (row_shr:8), lanes 0..7 keep `lo`.
// bank_mask 0b1100 enables the write for banks 2, 3 (lanes 8..15 of the row) only.
__device__ __forceinline__ float take_second_tile(float lo, float hi) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lo), __float_as_int(hi), 0x118, 0xf, 0xc, false));
}

// 4 consecutive hidden units of one sequence, as stored in global memory
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float4 type; };
template <> struct Vec4<__hip_bfloat16> { typedef uint2 type; };

__device__ __forceinline__ void unpack4(const float4& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void unpack4(const uint2& v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x << 16);
  f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16);
  f[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(lo))) |
         (static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(hi))) << 16);
}
__device__ __forceinline__ void pack4(const float (&f)[4], float4& v) { v = make_float4(f[0], f[1], f[2], f[3]); }
__device__ __forceinline__ void pack4(const float (&f)[4], uint2& v) { v = make_uint2(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])); }

// Saved activations (r, z, n, gh_n, h) for the backward pass: fp32 in the fp32 build; fp16 in the bf16 build
// (all five are O(1) values -- r, z in (0,1), n, h in (-1,1) -- so fp16's 11-bit significand keeps them 8x
// finer than bf16 would, at half of fp32's bytes).
template <typename T> struct SavedVec;
template <> struct SavedVec<float> { typedef float4 type; };
template <> struct SavedVec<__hip_bfloat16> { typedef uint2 type; };
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const f16x2 hv = __builtin_convertvector(v, f16x2);       // round to nearest even
  return __builtin_bit_cast(unsigned, hv);
}
__device__ __forceinline__ void pack_saved(const float (&f)[4], float4& v) { v = make_float4(f[0], f[1], f[2], f[3]); }
__device__ __forceinline__ void pack_saved(const float (&f)[4], uint2& v) {
  v = make_uint2(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]));
}
__device__ __forceinline__ void unpack_saved(const float4& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
__device__ __forceinline__ void unpack_saved(const uint2& v, float (&f)[4]) {
  const f16x2 a = __builtin_bit_cast(f16x2, v.x), b = __builtin_bit_cast(f16x2, v.y);
  f[0] = static_cast<float>(a[0]);
  f[1] = static_cast<float>(a[1]);
  f[2] = static_cast<float>(b[0]);
  f[3] = static_cast<float>(b[1]);
}

// Two 4-unit groups of one lane stored side by side (one 16-byte access in the bf16 build): the CU's store path is
// ISSUE-bound (MI355X_MICROARCH.md, "store-ISSUE-bound ... 8x dwordx4 halves it"), so per step the saved
// activations go out as 2 stores (r|z, n|gh_n) and the gate gradients as 2 (da_r|da_z, da_n|da_n*r).
template <typename V> struct Pair { V a, b; };
template <> struct alignas(16) Pair<uint2> { uint2 a, b; };
template <> struct alignas(16) Pair<float4> { float4 a, b; };

// ---- private "tile" layout of every per-step tensor the kernels stream (gi, saved gates, dy, dg) --------
// Lane (q = lane>>4, c = lane&15) of wavefront w owns sequence (c & 7) of its 8-sequence tile and the 4 units
// 32w + 16(c>>3) + 4q .. +3.  Storing the 4-unit group of lane `lane` at
//     group(tile, t, dir, w, slot, lane) = (((((tile*T + t)*2 + dir)*8 + w)*NS + slot)*64 + lane)
// (x4 elements) makes every load / store instruction of a wavefront ONE contiguous 512 B / 1 KB run.
// In the natural [B][T][...][H] layout the same instruction touches 8 rows x 32..64 B: measured, the
// CU's store path then takes ~1.6 us per step for the stores of a step -- more than all the arithmetic.
// The host converts gi / dy into this layout and dg back with one permute each (tens of MB, microseconds).
__device__ __forceinline__ long tile_group(long tile, long T, long t, int dir, int w, int ns, int slot, int lane) {
  return (((((tile * T + t) * 2 + dir) * 8 + w) * ns + slot) * 64 + lane);
}

struct GruFwdArgs {
  const void* gi;        // tile layout, NS = 3 (r | z | n), dtype T; r/z already include b_hh
  const __hip_bfloat16* w_hh;   // [2][3H][H]
  const float* b_hn;     // [2][H]    recurrent bias of the n gate
  void* y;               // [tiles*8][T][2H]  natural layout (what the next layer's GEMM reads), dtype T
  void* saved;           // tile layout, NS = 2 pairs (r|z, n|gh_n) of SavedVec<T> groups (nullptr: inference)
  long tiles, T;
};

// The MFMA computes gh^T = W_hh h^T: M = gate rows (units), N = sequence columns; with D[m = 4q+i][n = c] a lane
// holds 4 consecutive units of one column.  The step body has NO divergent control flow (batch padded to whole
// tiles by the host; kSave compile time): the compiler counts outstanding loads / stores exactly and never
// drains the queue.
template <typename T, bool kSave>
__device__ __forceinline__ void gru_forward_steps(const GruFwdArgs& a, bf16x8* wn_lds, __hip_bfloat16* hbuf,
                                                  const bf16x8 (&wr)[2][8], const bf16x8 (&wz)[2][8]) {
  typedef typename Vec4<T>::type V4;
  typedef typename SavedVec<T>::type SV;
  constexpr bool kLdsY = sizeof(T) == 2;   // bf16: y rows are written from the LDS h tile (512-B runs)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int seq = c & 7, s_own = c >> 3;
  const int unit0 = 32 * wave + 16 * s_own + 4 * q;       // the 4 units this lane post-processes
  const int dir = blockIdx.y;
  const long tile = blockIdx.x;
  const float* bh = a.b_hn + dir * kH;
  const V4* gi = static_cast<const V4*>(a.gi);
  T* y = static_cast<T*>(a.y);
  typedef Pair<SV> SP;
  SP* saved = static_cast<SP*>(a.saved);
  const long b = tile * kSeqs + seq;       // this lane's sequence (natural-layout row)

  float bias_n[4], h_prev[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bias_n[i] = bh[unit0 + i];
    h_prev[i] = 0.0f;
  }

  auto time_of = [&](long step) { return dir == 0 ? step : a.T - 1 - step; };
  auto load_gi = [&](long step, V4 (&g)[3]) {
    // one base pointer per step + compile-time offsets (gate*64 groups): a single address register
    const V4* p = gi + tile_group(tile, a.T, time_of(step), dir, wave, 3, 0, lane);
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) g[gate] = p[gate * 64];
  };

  // `g` holds this step's operands on entry; they are unpacked at once and the SAME registers then receive
  // the next step's operands (a whole step of cover), so no second operand set is needed.
  auto step = [&](long t, V4 (&g)[3]) {
    const long tt = time_of(t);
    const int cur = static_cast<int>(t & 1), nxt = cur ^ 1;
    SP* const save_base = kSave ? saved + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane) : nullptr;
    float gir[4], giz[4], gin[4];
    unpack4(g[0], gir);                    // gi already holds b_ih + b_hh for the r and z gates
    unpack4(g[1], giz);
    unpack4(g[2], gin);
    // operands of the next step: unconditional (clamped at the last step) and pinned here -- the scheduler
    // otherwise sinks the loads below this step's stores, and the in-order vmcnt then makes the next step
    // wait for those stores' round trip.
    __builtin_amdgcn_sched_barrier(0);
    load_gi(t + 1 < a.T ? t + 1 : a.T - 1, g);
    __builtin_amdgcn_sched_barrier(0);
    // ---- gh^T = W_hh h^T : B fragments (k, n = column c) of h_{t-1} and the n-gate A fragments from LDS,
    // read one k-step ahead of the MFMAs that consume them (the LDS latency hides under 6 MFMAs)
    f32x4 acc_r[2], acc_z[2], acc_n[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_r[s][i] = acc_z[s][i] = acc_n[s][i] = 0.0f;
    const __hip_bfloat16* hrow = hbuf + (cur * kRows + c) * kHPitch + 8 * q;
    const bf16x8* wnp = wn_lds + wave * 2 * 8 * 64 + lane;
    bf16x8 hfrag = *reinterpret_cast<const bf16x8*>(hrow);
    bf16x8 wn0 = wnp[0], wn1 = wnp[8 * 64];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      bf16x8 hfrag_n = hfrag, wn0_n = wn0, wn1_n = wn1;
      if (kk + 1 < 8) {
        hfrag_n = *reinterpret_cast<const bf16x8*>(hrow + 32 * (kk + 1));
        wn0_n = wnp[(kk + 1) * 64];
        wn1_n = wnp[(8 + kk + 1) * 64];
      }
      acc_r[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[0][kk], hfrag, acc_r[0], 0, 0, 0);
      acc_z[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wz[0][kk], hfrag, acc_z[0], 0, 0, 0);
      acc_n[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn0, hfrag, acc_n[0], 0, 0, 0);
      acc_r[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[1][kk], hfrag, acc_r[1], 0, 0, 0);
      acc_z[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wz[1][kk], hfrag, acc_z[1], 0, 0, 0);
      acc_n[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wn1, hfrag, acc_n[1], 0, 0, 0);
      hfrag = hfrag_n;
      wn0 = wn0_n;
      wn1 = wn1_n;
    }
    // ---- gates for this lane's (sequence, 4 units): columns 0..7 keep unit tile 0, columns 8..15 take unit
    // tile 1 of the column 8 to their left
    float rr[4], zz[4], nn[4], gg[4], hh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ghr = take_second_tile(acc_r[0][i], acc_r[1][i]);
      const float ghz = take_second_tile(acc_z[0][i], acc_z[1][i]);
      gg[i] = take_second_tile(acc_n[0][i], acc_n[1][i]) + bias_n[i];
      rr[i] = sigmoid_f(gir[i] + ghr);
      zz[i] = sigmoid_f(giz[i] + ghz);
      nn[i] = tanh_f(fmaf(rr[i], gg[i], gin[i]));
      hh[i] = fmaf(zz[i], h_prev[i] - nn[i], nn[i]);
      h_prev[i] = hh[i];
    }
    uint2 hb;
    pack4(hh, hb);
    *reinterpret_cast<uint2*>(hbuf + (nxt * kRows + seq) * kHPitch + unit0) = hb;
    if (!kLdsY) {
      V4 yv;
      pack4(hh, yv);
      *reinterpret_cast<V4*>(y + (b * a.T + tt) * (2 * kH) + dir * kH + unit0) = yv;
    }
    if (kSave) {
      SP v;                                // h itself is not saved: the backward pass reads h_{t-1} from y
      pack_saved(rr, v.a);
      pack_saved(zz, v.b);
      save_base[0] = v;
      pack_saved(nn, v.a);
      pack_saved(gg, v.b);
      save_base[64] = v;
    }
    __syncthreads();
    if (kLdsY) {
      // y[b][tt][dir*H .. +H) is a 512-B run: wavefront w writes row w of the fresh h tile
      const uint2 v = *reinterpret_cast<const uint2*>(hbuf + (nxt * kRows + wave) * kHPitch + lane * 4);
      *reinterpret_cast<uint2*>(y + ((tile * kSeqs + wave) * a.T + tt) * (2 * kH) + dir * kH + lane * 4) = v;
    }
  };

  V4 g[3];
  load_gi(0, g);
  // The first step is peeled so that the loop is ENTERED in the same memory-queue state as the back edge
  // leaves it; otherwise the compiler merges the two states conservatively and every step waits for the
  // previous step's stores.
  step(0, g);
#pragma unroll 1
  for (long t = 1; t < a.T; ++t) step(t, g);
}

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_forward_kernel(GruFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);
  __hip_bfloat16* hbuf = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);   // [2][16][kHPitch]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const __hip_bfloat16* w = a.w_hh + static_cast<long>(blockIdx.y) * kG * kH;

  // ---- resident weights (A operand: row = unit l&15 of the tile, k = 8(l>>4)+j): r,z -> VGPRs, n -> LDS
  bf16x8 wr[2][8], wz[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * wave + 16 * s + c;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int k0 = 32 * kk + 8 * q;
      wr[s][kk] = *reinterpret_cast<const bf16x8*>(w + static_cast<long>(unit) * kH + k0);
      wz[s][kk] = *reinterpret_cast<const bf16x8*>(w + static_cast<long>(kH + unit) * kH + k0);
      wn_lds[((wave * 2 + s) * 8 + kk) * 64 + lane] =
          *reinterpret_cast<const bf16x8*>(w + static_cast<long>(2 * kH + unit) * kH + k0);
    }
  }
#ifdef SELD_GRU_SKEW
  if (wave < 4) __builtin_amdgcn_s_setprio(SELD_GRU_SKEW);   // experiment: de-phase the two waves of a SIMD
#endif
  // rows 8..15 of both h tiles are never written again: the padding columns of every MFMA stay zero
  for (int i = tid; i < 2 * kRows * kHPitch; i += kGruThreads) hbuf[i] = __float2bfloat16(0.0f);
  __syncthreads();
  if (a.saved) gru_forward_steps<T, true>(a, wn_lds, hbuf, wr, wz);
  else gru_forward_steps<T, false>(a, wn_lds, hbuf, wr, wz);
}

struct GruBwdArgs {
  const void* dy;        // tile layout, NS = 1, dtype T
  const void* saved;     // tile layout, NS = 2 pairs (r|z, n|gh_n) of SavedVec<T> groups
  const void* y;         // [tiles*8][T][2H] the forward output (h_t), natural layout, dtype T
  const __hip_bfloat16* w_hh_t;   // [2][H][3H]   W_hh transposed per direction
  void* dg;              // tile layout, NS = 2 pairs (da_r|da_z, da_n|da_n*r), dtype T
  float* dbias;          // [tiles][2][4][H]  per-tile sums over (sequence, t) of the four dg slots (fp32)
  long tiles, T;
};

template <typename T> struct GruStepIn {
  Pair<typename SavedVec<T>::type> rz, ng;
  typename Vec4<T>::type hp, d;
};

// dh_prev^T = W_hh^T dgh^T : M = hidden units, N = sequence columns; same lane ownership as the forward kernel.
template <typename T>
__device__ __forceinline__ void gru_backward_steps(const GruBwdArgs& a, bf16x8* wn_lds, __hip_bfloat16* dgh,
                                                   const bf16x8 (&wrz)[2][16]) {
  typedef typename Vec4<T>::type V4;
  typedef typename SavedVec<T>::type SV;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int seq = c & 7, s_own = c >> 3;
  const int unit0 = 32 * wave + 16 * s_own + 4 * q;
  const int dir = blockIdx.y;
  const long tile = blockIdx.x;
  const V4* dy = static_cast<const V4*>(a.dy);
  typedef Pair<SV> SP;
  typedef Pair<V4> DP;
  const SP* saved = static_cast<const SP*>(a.saved);
  const T* y = static_cast<const T*>(a.y);
  DP* dg = static_cast<DP*>(a.dg);
  const long b = tile * kSeqs + seq;

  float dh[4], bsum[4][4];           // bsum: running sums of da_r, da_z, da_n, da_n*r (the bias gradients)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dh[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) bsum[k][i] = 0.0f;
  }

  auto time_of = [&](long step) { return dir == 0 ? step : a.T - 1 - step; };
  auto load_step = [&](long step, GruStepIn<T>& in) {
    const long tt = time_of(step);
    const long tprev = time_of(step > 0 ? step - 1 : 0);      // h_{t-1} of the forward recurrence (unused at step 0)
    const SP* sp = saved + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane);
    in.rz = sp[0];
    in.ng = sp[64];
    in.hp = *reinterpret_cast<const V4*>(y + (b * a.T + tprev) * (2 * kH) + dir * kH + unit0);
    in.d = dy[tile_group(tile, a.T, tt, dir, wave, 1, 0, lane)];
  };

  // `in` holds this step's operands on entry; they are unpacked at once and the same registers then receive
  // the operands of the next (earlier-in-time) step.
  auto step = [&](long t, GruStepIn<T>& in) {
    const long tt = time_of(t);
    float r[4], z[4], n[4], g[4], hp[4], d[4];
    unpack_saved(in.rz.a, r);
    unpack_saved(in.rz.b, z);
    unpack_saved(in.ng.a, n);
    unpack_saved(in.ng.b, g);
    unpack4(in.hp, hp);
    unpack4(in.d, d);
    __builtin_amdgcn_sched_barrier(0);
    load_step(t > 0 ? t - 1 : 0, in);              // unconditional, clamped; pinned ahead of this step's stores
    __builtin_amdgcn_sched_barrier(0);
    float keep[4], da_r[4], da_z[4], da_n[4], dghn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float hprev = t > 0 ? hp[i] : 0.0f;
      const float dtot = d[i] + dh[i];
      const float dn = dtot * (1.0f - z[i]);
      const float dz = dtot * (hprev - n[i]);
      da_n[i] = dn * (1.0f - n[i] * n[i]);
      da_z[i] = dz * z[i] * (1.0f - z[i]);
      da_r[i] = da_n[i] * g[i] * r[i] * (1.0f - r[i]);
      dghn[i] = da_n[i] * r[i];
      keep[i] = dtot * z[i];
      bsum[0][i] += da_r[i];
      bsum[1][i] += da_z[i];
      bsum[2][i] += da_n[i];
      bsum[3][i] += dghn[i];
    }
    DP* const gp = dg + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane);
    DP v;
    pack4(da_r, v.a);
    pack4(da_z, v.b);
    gp[0] = v;
    pack4(da_n, v.a);
    pack4(dghn, v.b);
    gp[64] = v;
    __hip_bfloat16* drow = dgh + seq * kDghPitch + unit0;
    uint2 pk;
    pack4(da_r, pk);
    *reinterpret_cast<uint2*>(drow) = pk;
    pack4(da_z, pk);
    *reinterpret_cast<uint2*>(drow + kH) = pk;
    pack4(dghn, pk);
    *reinterpret_cast<uint2*>(drow + 2 * kH) = pk;
    __syncthreads();

    // two independent accumulator chains per unit tile (even / odd k-steps): with a single chain per tile the
    // 24 dependent MFMAs of a step run at the MFMA latency, not at its issue rate
    f32x4 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[s][0][i] = acc[s][1][i] = 0.0f;
    const __hip_bfloat16* brow = dgh + c * kDghPitch + 8 * q;
    const bf16x8* wnp = wn_lds + wave * 2 * 8 * 64 + lane;
    // LDS fragments are read two k-steps ahead of the MFMAs that consume them
    bf16x8 d0 = *reinterpret_cast<const bf16x8*>(brow), d1 = *reinterpret_cast<const bf16x8*>(brow + 32);
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
      bf16x8 d2 = d1;
      if (kk + 2 < 24) d2 = *reinterpret_cast<const bf16x8*>(brow + 32 * (kk + 2));
      if (kk < 16) {
        acc[0][kk & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wrz[0][kk < 16 ? kk : 0], d0, acc[0][kk & 1], 0, 0, 0);
        acc[1][kk & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wrz[1][kk < 16 ? kk : 0], d0, acc[1][kk & 1], 0, 0, 0);
      } else {
        const bf16x8 w0 = wnp[(kk - 16) * 64], w1 = wnp[(8 + kk - 16) * 64];
        acc[0][kk & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, d0, acc[0][kk & 1], 0, 0, 0);
        acc[1][kk & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, d0, acc[1][kk & 1], 0, 0, 0);
      }
      d0 = d1;
      d1 = d2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dh[i] = keep[i] + take_second_tile(acc[0][0][i] + acc[0][1][i], acc[1][0][i] + acc[1][1][i]);
    __syncthreads();
  };

  GruStepIn<T> in;
  load_step(a.T - 1, in);
  // reverse of the forward processing order; first step peeled (see the forward kernel)
  step(a.T - 1, in);
#pragma unroll 1
  for (long t = a.T - 2; t >= 0; --t) step(t, in);

  // bias gradients: add the 8 sequences of the tile (lanes that differ in c & 7), one float4 per slot and lane group
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[k][i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      bsum[k][i] = v;
    }
  if (seq == 0) {
    float* out = a.dbias + ((tile * 2 + dir) * 4) * kH + unit0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(out + k * kH) = make_float4(bsum[k][0], bsum[k][1], bsum[k][2], bsum[k][3]);
  }
}

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_backward_kernel(GruBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);                               // k-range of the n gate
  __hip_bfloat16* dgh = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);       // [16][kDghPitch]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const __hip_bfloat16* wt = a.w_hh_t + static_cast<long>(blockIdx.y) * kH * kG;  // [H][3H]

  bf16x8 wrz[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int unit = 32 * wave + 16 * s + c;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      wrz[s][kk] = *reinterpret_cast<const bf16x8*>(wt + static_cast<long>(unit) * kG + 32 * kk + 8 * q);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      wn_lds[((wave * 2 + s) * 8 + kk) * 64 + lane] =
          *reinterpret_cast<const bf16x8*>(wt + static_cast<long>(unit) * kG + 2 * kH + 32 * kk + 8 * q);
  }
#ifdef SELD_GRU_SKEW
  if (wave < 4) __builtin_amdgcn_s_setprio(SELD_GRU_SKEW);
#endif
  // rows 8..15 (padding columns of the MFMA) stay zero for the whole kernel
  for (int i = tid; i < kRows * kDghPitch; i += kGruThreads) dgh[i] = __float2bfloat16(0.0f);
  __syncthreads();
  gru_backward_steps<T>(a, wn_lds, dgh, wrz);
}

// ---- layout converters between the natural [B][T][2][NS][H] tensors of the host GEMMs and the tile layout ----
// One block per (tile, t, direction): 8 sequences x NS x 64 four-unit groups.  The tile side is accessed as one
// contiguous run per block; the natural side as 64-byte (bf16) / 128-byte (fp32) pieces of 512-B / 1-KB rows,
// every byte exactly once.  HBM-bound permutes (read + write of the tensor), replacing strided framework copies.
template <typename G>
__global__ __launch_bounds__(256) void gru_to_tile_kernel(const G* __restrict__ src, long B, long T, int ns,
                                                          G* __restrict__ dst) {
  const long blk = blockIdx.x;
  const int dir = static_cast<int>(blk & 1);
  const long t = (blk >> 1) % T, tile = (blk >> 1) / T;
  const int groups = 8 * ns * 64;
  G zero;
  __builtin_memset(&zero, 0, sizeof(G));
  for (int g = threadIdx.x; g < groups; g += 256) {
    const int lane = g & 63, slot = (g >> 6) % ns, w = (g >> 6) / ns;
    const int seq = lane & 7, s = (lane >> 3) & 1, q = lane >> 4;
    const long b = tile * kSeqs + seq;
    const int ug = 8 * w + 4 * s + q;                     // four-unit group along H: units 32w + 16s + 4q ..
    G v = zero;
    if (b < B) v = src[(((b * T + t) * 2 + dir) * ns + slot) * 64 + ug];
    dst[blk * groups + g] = v;
  }
}

template <typename G>
__global__ __launch_bounds__(256) void gru_from_pair_tile_kernel(const Pair<G>* __restrict__ src, long B, long T,
                                                                 G* __restrict__ dgi, G* __restrict__ dghn) {
  const long blk = blockIdx.x;
  const int dir = static_cast<int>(blk & 1);
  const long t = (blk >> 1) % T, tile = (blk >> 1) / T;
  for (int g = threadIdx.x; g < 8 * 2 * 64; g += 256) {
    const int lane = g & 63, ps = (g >> 6) & 1, w = g >> 7;
    const int seq = lane & 7, s = (lane >> 3) & 1, q = lane >> 4;
    const long b = tile * kSeqs + seq;
    if (b >= B) continue;
    const Pair<G> p = src[blk * (8 * 2 * 64) + g];
    const int ug = 8 * w + 4 * s + q;
    const long row = (b * T + t) * 2 + dir;
    if (ps == 0) {
      dgi[(row * 3 + 0) * 64 + ug] = p.a;                 // da_r
      dgi[(row * 3 + 1) * 64 + ug] = p.b;                 // da_z
    } else {
      dgi[(row * 3 + 2) * 64 + ug] = p.a;                 // da_n
      dghn[row * 64 + ug] = p.b;                          // da_n * r
    }
  }
}

}  // namespace seld

extern "C" {

int seld_gru_to_tile(const void* src, int elem_bytes, int64_t B, int64_t T, int ns, void* dst, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (B <= 0 || T <= 0 || ns <= 0) return fail(kErrInvalidArgument, "seld_gru_to_tile: B, T, ns must be positive");
  if (elem_bytes != 2 && elem_bytes != 4) return fail(kErrUnsupported, "seld_gru_to_tile: 2- or 4-byte elements");
  if (!src || !dst) return fail(kErrInvalidArgument, "seld_gru_to_tile: null pointer");
  const long tiles = (B + kSeqs - 1) / kSeqs;
  const dim3 grid(static_cast<unsigned>(tiles * T * 2));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (elem_bytes == 2) hipLaunchKernelGGL(gru_to_tile_kernel<uint2>, grid, dim3(256), 0, stream,
                                          static_cast<const uint2*>(src), static_cast<long>(B), static_cast<long>(T), ns,
                                          static_cast<uint2*>(dst));
  else hipLaunchKernelGGL(gru_to_tile_kernel<float4>, grid, dim3(256), 0, stream, static_cast<const float4*>(src),
                          static_cast<long>(B), static_cast<long>(T), ns, static_cast<float4*>(dst));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gru_from_pair_tile(const void* dg_tile, int elem_bytes, int64_t B, int64_t T, void* dgi, void* dghn,
                            void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (B <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_from_pair_tile: B and T must be positive");
  if (elem_bytes != 2 && elem_bytes != 4) return fail(kErrUnsupported, "seld_gru_from_pair_tile: 2- or 4-byte elements");
  if (!dg_tile || !dgi || !dghn) return fail(kErrInvalidArgument, "seld_gru_from_pair_tile: null pointer");
  const long tiles = (B + kSeqs - 1) / kSeqs;
  const dim3 grid(static_cast<unsigned>(tiles * T * 2));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (elem_bytes == 2) hipLaunchKernelGGL(gru_from_pair_tile_kernel<uint2>, grid, dim3(256), 0, stream,
                                          static_cast<const Pair<uint2>*>(dg_tile), static_cast<long>(B),
                                          static_cast<long>(T), static_cast<uint2*>(dgi), static_cast<uint2*>(dghn));
  else hipLaunchKernelGGL(gru_from_pair_tile_kernel<float4>, grid, dim3(256), 0, stream,
                          static_cast<const Pair<float4>*>(dg_tile), static_cast<long>(B), static_cast<long>(T),
                          static_cast<float4*>(dgi), static_cast<float4*>(dghn));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int64_t seld_gru_tile_rows(void) { return seld::kSeqs; }

int seld_gru_forward(const void* gi_tile, int is_bf16, const void* w_hh_bf16, const float* b_hn, int64_t tiles,
                     int64_t T, int64_t H, void* y, void* saved_tile, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (H != kH) return fail(kErrUnsupported, "seld_gru_forward: built for hidden size 256 (config.py:45)");
  if (tiles <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_forward: tiles and T must be positive");
  if (!gi_tile || !w_hh_bf16 || !b_hn || !y) return fail(kErrInvalidArgument, "seld_gru_forward: null pointer");
  GruFwdArgs a{gi_tile, static_cast<const __hip_bfloat16*>(w_hh_bf16), b_hn, y, saved_tile, tiles, T};
  const dim3 grid(static_cast<unsigned>(tiles), 2);
  const size_t lds = kWnBytes + 2 * kRows * kHPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  static bool attr_done = false;          // not a stream operation: do it once so launches stay graph-capturable
  if (!attr_done) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_done = true;
  }
  if (is_bf16) hipLaunchKernelGGL(gru_forward_kernel<__hip_bfloat16>, grid, dim3(kGruThreads), lds, stream, a);
  else hipLaunchKernelGGL(gru_forward_kernel<float>, grid, dim3(kGruThreads), lds, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gru_backward(const void* dy_tile, const void* saved_tile, const void* y, int is_bf16,
                      const void* w_hh_t_bf16, int64_t tiles, int64_t T, int64_t H, void* dg_tile, float* dbias,
                      void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (H != kH) return fail(kErrUnsupported, "seld_gru_backward: built for hidden size 256 (config.py:45)");
  if (tiles <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_backward: tiles and T must be positive");
  if (!dy_tile || !saved_tile || !y || !w_hh_t_bf16 || !dg_tile || !dbias)
    return fail(kErrInvalidArgument, "seld_gru_backward: null pointer");
  GruBwdArgs a{dy_tile, saved_tile, y, static_cast<const __hip_bfloat16*>(w_hh_t_bf16), dg_tile, dbias, tiles, T};
  const dim3 grid(static_cast<unsigned>(tiles), 2);
  const size_t lds = kWnBytes + kRows * kDghPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  static bool attr_done = false;
  if (!attr_done) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    attr_done = true;
  }
  if (is_bf16) hipLaunchKernelGGL(gru_backward_kernel<__hip_bfloat16>, grid, dim3(kGruThreads), lds, stream, a);
  else hipLaunchKernelGGL(gru_backward_kernel<float>, grid, dim3(kGruThreads), lds, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
