// Persistent bidirectional GRU recurrence for gfx950 (forward and backward).
//
// Replaces the recurrent part of nn.GRU(2048 -> 256, 2 layers, bidirectional) at
// model_crnn.py:65-72 / :119.  The input projection  gi = x W_ih^T + b_ih  for all time steps is
// one large GEMM done by the host (hipBLASLt, MFMA); what remains is 250 strictly sequential
// steps per direction of
//     gh = h W_hh^T + b_hh ;  r = s(gi_r + gh_r) ; z = s(gi_z + gh_z) ; n = tanh(gi_n + r*gh_n)
//     h' = (1 - z) n + z h
// which a library executes as ~2 tiny kernels per step.  Here ONE workgroup per (direction, kSeqs-sequence batch
// tile) runs all T steps with W_hh resident on the CU:
//   * 8 wavefronts; wavefront w owns hidden units [32w, 32w+32) for all three gates;
//   * W_hh (bf16, 393 KB per direction) does not fit one place: the r and z gate rows live in
//     VGPRs as MFMA A-fragments (128 registers per lane), the n gate rows in LDS (128 KB) laid
//     out fragment-major so every ds_read_b128 is a linear conflict-free 1 KB read;
//   * h_{t-1} (bf16) is exchanged through a double-buffered kSeqs x 256 LDS tile (row pitch 528 B:
//     conflict-free B-fragment reads), one barrier per step;
//   * v_mfma_f32_16x16x32_bf16: per step 48 MFMAs per wavefront, fp32 accumulate, fp32 gates.
//
// WHY ONLY kSeqs = 4 SEQUENCES PER 16-COLUMN MFMA TILE.  With the weights resident, a step is bound by what ONE CU
// can do besides the MFMAs: its vector-memory path sustains only ~10-15 B/clk (MI355X_MICROARCH.md: "~10
// B/cyc/CU", stores issue-bound), and the gate math (6 transcendentals per element) issues on the same 4 SIMDs.
// Both scale with sequences per CU while the MFMA time (96 per SIMD, ~1.5k cycles) does not -- so the batch is
// spread over MORE CUs: the other columns of every MFMA are padding (they repeat a valid column), and after the
// MFMAs the 16 / kSeqs lanes that share a sequence split the column's 2 x 4 accumulators among themselves with one
// DPP row shift per value.  Every lane then owns ONE (sequence, kU units) group: a fraction of the gate
// instructions and of the bytes per CU per step, all 64 lanes active in every load and store.  Measured
// (B = 32, T = 250, us per step forward / backward): 16 sequences 1.68 / 3.33, 8 sequences 1.28 / 1.57,
// 4 sequences 1.18 / 1.40.
//
// The backward kernel mirrors it: dgh (bf16) goes through LDS as the B operand, W_hh^T fragments
// are register / LDS resident, dh is carried in registers; it emits the per-step gate gradients
// from which the host forms dW_ih, dW_hh, dx with large GEMMs, and the bias gradients directly.
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "seld_common.h"

#ifndef SELD_GRU_SEQS
#define SELD_GRU_SEQS 4
#endif

namespace seld {

constexpr int kH = 256;            // hidden size (config.py:45 CRNN_RNN_HIDDEN)
constexpr int kG = 3 * kH;         // gate rows r | z | n
constexpr int kRows = 16;          // MFMA N (columns)
constexpr int kSeqs = SELD_GRU_SEQS;   // sequences per workgroup = valid MFMA columns (the rest are padding)
constexpr int kParts = kRows / kSeqs;  // lanes of a 16-lane row that share one sequence
constexpr int kU = 8 / kParts;         // hidden units a lane post-processes (of the 2 x 4 its column computes)
constexpr int kGruThreads = 512;   // 8 wavefronts
constexpr int kHPitch = kH + 8;    // bf16 elements per h row in LDS (528 B)
constexpr int kDghPitch = kG + 8;  // bf16 elements per dgh row in LDS (1552 B)
constexpr int kWnBytes = 8 * 2 * 8 * 64 * 16;   // [wave][tile][kstep][lane] x 16 B = 131072
static_assert(kSeqs == 8 || kSeqs == 4, "8 or 4 sequences per tile");

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: the gates are fp32 but not bit-critical
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(x) = 1 - 2 / (1 + e^{2x}): saturates correctly (e = inf -> 1, e = 0 -> -1) without a clamp
__device__ __forceinline__ float tanh_f(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)), 1.0f); }

// ---- which values of its column's 2 x 4 accumulators a lane post-processes -------------------------------------
// Column c of a 16-lane row holds sequence c % kSeqs; the kParts lanes that share a sequence split the 8 values
// (unit tile s = 0, 1; unit i = 0..3; flat index 4 s + i) of the ONE valid column (part 0's) among themselves:
// part p = c / kSeqs takes flat indices p kU .. p kU + kU - 1, fetched from the lane p kSeqs to its left with
// one DPP row shift (row_shr) per value; bank_mask enables the write only for the lanes of that part.
template <int kPart> __device__ __forceinline__ float shifted_part(float old, float src) {
  constexpr int ctrl = 0x110 + kPart * kSeqs;                                       // row_shr:(kPart * kSeqs)
  constexpr int banks = kSeqs == 8 ? 0xc : (1 << kPart);                            // banks = groups of 4 lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), ctrl, 0xf, banks, false));
}
// v[s][i]: this column's accumulators; returns the kU values this lane owns
__device__ __forceinline__ void own_values(const f32x4 (&v)[2], float (&out)[kU]) {
#pragma unroll
  for (int j = 0; j < kU; ++j) {
    float r = v[j >> 2][j & 3];                                                      // part 0: flat index j
    if (kParts >= 2) r = shifted_part<1>(r, v[(kU + j) >> 2][(kU + j) & 3]);
    if (kParts == 4) {
      r = shifted_part<2>(r, v[(2 * kU + j) >> 2][(2 * kU + j) & 3]);
      r = shifted_part<3>(r, v[(3 * kU + j) >> 2][(3 * kU + j) & 3]);
    }
    out[j] = r;
  }
}

// ---- storage of a lane's kU-unit group ---------------------------------------------------------------------------
template <int kBytes> struct Raw;
template <> struct alignas(4) Raw<4> { unsigned w[1]; };
template <> struct alignas(8) Raw<8> { unsigned w[2]; };
template <> struct alignas(16) Raw<16> { unsigned w[4]; };
template <> struct alignas(16) Raw<32> { unsigned w[8]; };

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(lo))) |
         (static_cast<unsigned>(__bfloat16_as_ushort(__float2bfloat16(hi))) << 16);
}
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));            // round to nearest even
}

struct AsF32 {       // fp32 build: data and saved activations
  static constexpr int kBytes = 4;
  template <int N> static __device__ __forceinline__ void enc(const float (&f)[N], unsigned* w) {
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = __float_as_uint(f[k]);
  }
  template <int N> static __device__ __forceinline__ void dec(const unsigned* w, float (&f)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) f[k] = __uint_as_float(w[k]);
  }
};
struct AsBF16 {      // bf16 build: streamed data (gi, dy, dg, y)
  static constexpr int kBytes = 2;
  template <int N> static __device__ __forceinline__ void enc(const float (&f)[N], unsigned* w) {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) w[k] = pack_bf16x2(f[2 * k], f[2 * k + 1]);
  }
  template <int N> static __device__ __forceinline__ void dec(const unsigned* w, float (&f)[N]) {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      f[2 * k] = __uint_as_float(w[k] << 16);
      f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
    }
  }
};
// Saved activations of the bf16 build: IEEE fp16.  r, z in (0,1), n in (-1,1), gh_n = O(1): fp16's 11-bit significand
// keeps them 8x finer than bf16 would at half of fp32's bytes (bytes per step per CU are what bounds the recurrence).
struct AsF16 {
  static constexpr int kBytes = 2;
  template <int N> static __device__ __forceinline__ void enc(const float (&f)[N], unsigned* w) {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) w[k] = pack_f16x2(f[2 * k], f[2 * k + 1]);
  }
  template <int N> static __device__ __forceinline__ void dec(const unsigned* w, float (&f)[N]) {
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      const f16x2 h = __builtin_bit_cast(f16x2, w[k]);
      f[2 * k] = static_cast<float>(h[0]);
      f[2 * k + 1] = static_cast<float>(h[1]);
    }
  }
};
template <typename T> struct Types;
template <> struct Types<float> { typedef AsF32 Data; typedef AsF32 Saved; };
template <> struct Types<__hip_bfloat16> { typedef AsBF16 Data; typedef AsF16 Saved; };

// one group / two groups side by side (one access): the CU's store path is ISSUE-bound (MI355X_MICROARCH.md,
// "store-ISSUE-bound ... 8x dwordx4 halves it"), so per step the saved activations go out as 2 stores
// (r|z, n|gh_n) and the gate gradients as 2 (da_r|da_z, da_n|da_n*r).
template <typename C> using Group = Raw<C::kBytes * kU>;
template <typename C> using PairOf = Raw<2 * C::kBytes * kU>;
template <typename C> __device__ __forceinline__ Group<C> enc1(const float (&a)[kU]) {
  Group<C> g;
  C::template enc<kU>(a, g.w);
  return g;
}
template <typename C> __device__ __forceinline__ void dec1(const Group<C>& g, float (&a)[kU]) { C::template dec<kU>(g.w, a); }
template <typename C> __device__ __forceinline__ PairOf<C> enc2(const float (&a)[kU], const float (&b)[kU]) {
  PairOf<C> p;
  C::template enc<kU>(a, p.w);
  C::template enc<kU>(b, p.w + C::kBytes * kU / 4);
  return p;
}
template <typename C> __device__ __forceinline__ void dec2(const PairOf<C>& p, float (&a)[kU], float (&b)[kU]) {
  C::template dec<kU>(p.w, a);
  C::template dec<kU>(p.w + C::kBytes * kU / 4, b);
}

// ---- private "tile" layout of every per-step tensor the kernels stream (gi, saved gates, dy, dg) --------
// Lane (q = lane>>4, c = lane&15) of wavefront w owns sequence c % kSeqs of its tile and the kU units starting at
//     unit0 = 32w + 4q + [flat index p kU decomposed as 16 s + i],   p = c / kSeqs.
// Storing the group of lane `lane` at
//     group(tile, t, dir, w, slot, lane) = (((((tile*T + t)*2 + dir)*8 + w)*NS + slot)*64 + lane)
// makes every load / store instruction of a wavefront ONE contiguous run.  In the natural [B][T][...][H] layout
// the same instruction touches kSeqs rows x 16..64 B: measured, the CU's store path then takes ~1.6 us per step
// for the stores of a step -- more than all the arithmetic.  seld_gru_to_tile / seld_gru_from_pair_tile convert.
__device__ __forceinline__ long tile_group(long tile, long T, long t, int dir, int w, int ns, int slot, int lane) {
  return (((((tile * T + t) * 2 + dir) * 8 + w) * ns + slot) * 64 + lane);
}
__device__ __forceinline__ int lane_unit0(int wave, int lane) {
  const int q = lane >> 4, part = (lane & 15) / kSeqs;
  const int flat = part * kU;                          // 4 s + i
  return 32 * wave + 16 * (flat >> 2) + 4 * q + (flat & 3);
}

struct GruFwdArgs {
  const void* gi;        // [B][T][2][3][H] natural layout (the input GEMM's output), dtype T; r/z already include b_hh
  const __hip_bfloat16* w_hh;   // [2][3H][H]
  const float* b_hn;     // [2][H]    recurrent bias of the n gate
  void* y;               // [tiles*kSeqs][T][2H]  natural layout (what the next layer's GEMM reads), dtype T
  void* saved;           // tile layout, NS = 2 pairs (r|z, n|gh_n), Saved encoding (nullptr: inference)
  long tiles, T, B;      // B sequences; the lanes of the last tile's padding sequences re-read sequence B-1
};

// The MFMA computes gh^T = W_hh h^T: M = gate rows (units), N = sequence columns; D[m = 4q+i][n = c].  The step
// body has NO divergent control flow (batch padded to whole tiles by the host; kSave compile time): the compiler
// counts outstanding loads / stores exactly and never drains the queue.
template <typename T, bool kSave>
__device__ __forceinline__ void gru_forward_steps(const GruFwdArgs& a, bf16x8* wn_lds, __hip_bfloat16* hbuf,
                                                  const bf16x8 (&wr)[2][8], const bf16x8 (&wz)[2][8]) {
  typedef typename Types<T>::Data D;
  typedef typename Types<T>::Saved S;
  constexpr bool kLdsY = sizeof(T) == 2;   // bf16: y rows are written from the LDS h tile (512-B runs)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int seq = c % kSeqs;
  const int unit0 = lane_unit0(wave, lane);               // the kU units this lane post-processes
  const int dir = blockIdx.y;
  const long tile = blockIdx.x;
  const float* bh = a.b_hn + dir * kH;
  const Group<D>* gi = static_cast<const Group<D>*>(a.gi);
  T* y = static_cast<T*>(a.y);
  PairOf<S>* saved = static_cast<PairOf<S>*>(a.saved);
  const long b = tile * kSeqs + seq;       // this lane's sequence (natural-layout row)

  float bias_n[kU], h_prev[kU];
#pragma unroll
  for (int i = 0; i < kU; ++i) {
    bias_n[i] = bh[unit0 + i];
    h_prev[i] = 0.0f;
  }

  auto time_of = [&](long step) { return dir == 0 ? step : a.T - 1 - step; };
  const long b_read = b < a.B ? b : a.B - 1;
  auto load_gi = [&](long step, Group<D> (&g)[3]) {
    // straight from the input GEMM's output (no layout permute): per wavefront load, kSeqs runs of 32 units (64 B in
    // bf16), one per sequence; one base pointer per step + compile-time offsets (gate * H): a single address register
    const Group<D>* p = gi + (((b_read * a.T + time_of(step)) * 2 + dir) * 3 * kH + unit0) / kU;
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) g[gate] = p[gate * (kH / kU)];
  };

  // `g` holds this step's operands on entry; they are unpacked at once and the same registers then receive the
  // operands of the step after the next one (two register sets used alternately: two steps, ~2.4 us, of cover for
  // every global load -- one step is not enough: HBM answers in about a step's time, later under load).
  auto step = [&](long t, Group<D> (&g)[3]) {
    const long tt = time_of(t);
    const int cur = static_cast<int>(t & 1), nxt = cur ^ 1;
    PairOf<S>* const save_base = kSave ? saved + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane) : nullptr;
    float gir[kU], giz[kU], gin[kU];
    dec1<D>(g[0], gir);                    // gi already holds b_ih + b_hh for the r and z gates
    dec1<D>(g[1], giz);
    dec1<D>(g[2], gin);
    // operands of the next step: unconditional (clamped at the last step) and pinned here -- the scheduler
    // otherwise sinks the loads below this step's stores, and the in-order vmcnt then makes the next step
    // wait for those stores' round trip.
    __builtin_amdgcn_sched_barrier(0);
    load_gi(t + 2 < a.T ? t + 2 : a.T - 1, g);
    __builtin_amdgcn_sched_barrier(0);
    // ---- gh^T = W_hh h^T : B fragments (k, n = column c) of h_{t-1} and the n-gate A fragments from LDS,
    // read one k-step ahead of the MFMAs that consume them (the LDS latency hides under 6 MFMAs).  The padding
    // columns read row c % kSeqs again: their results are never used.
    f32x4 acc_r[2], acc_z[2], acc_n[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_r[s][i] = acc_z[s][i] = acc_n[s][i] = 0.0f;
    const __hip_bfloat16* hrow = hbuf + (cur * kSeqs + seq) * kHPitch + 8 * q;
    const bf16x8* wnp = wn_lds + wave * 2 * 8 * 64 + lane;
    float ghr[kU], ghz[kU], gg[kU], rr[kU], zz[kU], nn[kU], hh[kU];
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
    // ---- gates for this lane's (sequence, kU units)
    own_values(acc_r, ghr);
    own_values(acc_z, ghz);
    own_values(acc_n, gg);
#pragma unroll
    for (int i = 0; i < kU; ++i) {
      gg[i] += bias_n[i];
      rr[i] = sigmoid_f(gir[i] + ghr[i]);
      zz[i] = sigmoid_f(giz[i] + ghz[i]);
    }
#pragma unroll
    for (int i = 0; i < kU; ++i) {
      nn[i] = tanh_f(fmaf(rr[i], gg[i], gin[i]));
      hh[i] = fmaf(zz[i], h_prev[i] - nn[i], nn[i]);
      h_prev[i] = hh[i];
    }
    *reinterpret_cast<Group<AsBF16>*>(hbuf + (nxt * kSeqs + seq) * kHPitch + unit0) = enc1<AsBF16>(hh);
    if (!kLdsY) *reinterpret_cast<Group<D>*>(y + (b * a.T + tt) * (2 * kH) + dir * kH + unit0) = enc1<D>(hh);
    if (kSave) {                           // h itself is not saved: the backward pass reads h_{t-1} from y
      save_base[0] = enc2<S>(rr, zz);
      save_base[64] = enc2<S>(nn, gg);
    }
    __syncthreads();
    if (kLdsY) {
      // y[b][tt][dir*H .. +H) is a 512-B run: the 8 wavefronts write the kSeqs rows of the fresh h tile, one row
      // (kSeqs = 8) or half a row (kSeqs = 4) each, as one contiguous run per wavefront
      constexpr int kPer = 4 * kSeqs / 8;                    // bf16 elements per lane
      const int row = wave * kSeqs / 8, col = (wave % (8 / kSeqs)) * (64 * kPer) + lane * kPer;
      typedef Raw<2 * kPer> Piece;
      const Piece v = *reinterpret_cast<const Piece*>(hbuf + (nxt * kSeqs + row) * kHPitch + col);
      *reinterpret_cast<Piece*>(y + ((tile * kSeqs + row) * a.T + tt) * (2 * kH) + dir * kH + col) = v;
    }
  };

  Group<D> g_a[3], g_b[3];
  load_gi(0, g_a);
  load_gi(a.T > 1 ? 1 : 0, g_b);
  // The first step is peeled so that the loop is ENTERED in the same memory-queue state as the back edge
  // leaves it; otherwise the compiler merges the two states conservatively and every step waits for the
  // previous step's stores.
  step(0, g_a);
  long t = 1;
#pragma unroll 1
  for (; t + 1 < a.T; t += 2) {
    step(t, g_b);
    step(t + 1, g_a);
  }
  if (t < a.T) step(t, g_b);
}

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_forward_kernel(GruFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);
  __hip_bfloat16* hbuf = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);   // [2][kSeqs][kHPitch]
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
  for (int i = tid; i < 2 * kSeqs * kHPitch; i += kGruThreads) hbuf[i] = __float2bfloat16(0.0f);   // h_0 = 0
  __syncthreads();
  if (a.saved) gru_forward_steps<T, true>(a, wn_lds, hbuf, wr, wz);
  else gru_forward_steps<T, false>(a, wn_lds, hbuf, wr, wz);
}

struct GruBwdArgs {
  const void* dy;        // tile layout, NS = 1, dtype T
  const void* saved;     // tile layout, NS = 2 pairs (r|z, n|gh_n), Saved encoding
  const void* y;         // [tiles*kSeqs][T][2H] the forward output (h_t), natural layout, dtype T
  const __hip_bfloat16* w_hh_t;   // [2][H][3H]   W_hh transposed per direction
  void* dg;              // tile layout, NS = 2 pairs (da_r|da_z, da_n|da_n*r), dtype T
  float* dbias;          // [tiles][2][4][H]  per-tile sums over (sequence, t) of the four dg slots (fp32)
  long tiles, T;
};

template <typename T> struct GruStepIn {
  PairOf<typename Types<T>::Saved> rz, ng;
  Group<typename Types<T>::Data> hp, d;
};

// dh_prev^T = W_hh^T dgh^T : M = hidden units, N = sequence columns; same lane ownership as the forward kernel.
template <typename T>
__device__ __forceinline__ void gru_backward_steps(const GruBwdArgs& a, bf16x8* wn_lds, __hip_bfloat16* dgh,
                                                   const bf16x8 (&wrz)[2][16]) {
  typedef typename Types<T>::Data D;
  typedef typename Types<T>::Saved S;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int seq = c % kSeqs;
  const int unit0 = lane_unit0(wave, lane);
  const int dir = blockIdx.y;
  const long tile = blockIdx.x;
  const Group<D>* dy = static_cast<const Group<D>*>(a.dy);
  const PairOf<S>* saved = static_cast<const PairOf<S>*>(a.saved);
  const T* y = static_cast<const T*>(a.y);
  PairOf<D>* dg = static_cast<PairOf<D>*>(a.dg);
  const long b = tile * kSeqs + seq;

  float dh[kU], bsum[4][kU];         // bsum: running sums of da_r, da_z, da_n, da_n*r (the bias gradients)
#pragma unroll
  for (int i = 0; i < kU; ++i) {
    dh[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) bsum[k][i] = 0.0f;
  }

  auto time_of = [&](long step) { return dir == 0 ? step : a.T - 1 - step; };
  auto load_step = [&](long step, GruStepIn<T>& in) {
    const long tt = time_of(step);
    const long tprev = time_of(step > 0 ? step - 1 : 0);      // h_{t-1} of the forward recurrence (unused at step 0)
    const PairOf<S>* sp = saved + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane);
    in.rz = sp[0];
    in.ng = sp[64];
    in.hp = *reinterpret_cast<const Group<D>*>(y + (b * a.T + tprev) * (2 * kH) + dir * kH + unit0);
    in.d = dy[tile_group(tile, a.T, tt, dir, wave, 1, 0, lane)];
  };

  // `in` holds this step's operands on entry; they are unpacked at once and the same registers then receive the
  // operands of the step after the next one (two register sets, used alternately): every global load has two whole
  // steps (~3 us) to arrive, so the recurrence keeps its pace when weight-gradient GEMMs run beside it on the other
  // CUs (seld_overlap.py) and the memory system answers late.
  auto step = [&](long t, GruStepIn<T>& in) {
    const long tt = time_of(t);
    float r[kU], z[kU], n[kU], g[kU], hp[kU], d[kU];
    dec2<S>(in.rz, r, z);
    dec2<S>(in.ng, n, g);
    dec1<D>(in.hp, hp);
    dec1<D>(in.d, d);
    __builtin_amdgcn_sched_barrier(0);
    load_step(t > 1 ? t - 2 : 0, in);              // unconditional, clamped; pinned ahead of this step's stores
    __builtin_amdgcn_sched_barrier(0);
    float keep[kU], da_r[kU], da_z[kU], da_n[kU], dghn[kU];
#pragma unroll
    for (int i = 0; i < kU; ++i) {
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
    PairOf<D>* const gp = dg + tile_group(tile, a.T, tt, dir, wave, 2, 0, lane);
    gp[0] = enc2<D>(da_r, da_z);
    gp[64] = enc2<D>(da_n, dghn);
    // dgh tile: kSeqs rows, double buffered by step parity.  The padding columns of the MFMA read row c % kSeqs
    // again (their results are never used), so no zero rows are needed, and with two buffers the step needs ONE
    // barrier: a wavefront may start writing step t+1's tile while a slower one still reads step t's.
    __hip_bfloat16* const dgh_cur = dgh + (t & 1) * (kSeqs * kDghPitch);
    __hip_bfloat16* drow = dgh_cur + seq * kDghPitch + unit0;
    *reinterpret_cast<Group<AsBF16>*>(drow) = enc1<AsBF16>(da_r);
    *reinterpret_cast<Group<AsBF16>*>(drow + kH) = enc1<AsBF16>(da_z);
    *reinterpret_cast<Group<AsBF16>*>(drow + 2 * kH) = enc1<AsBF16>(dghn);
    __syncthreads();

    // two independent accumulator chains per unit tile (even / odd k-steps): with a single chain per tile the
    // 24 dependent MFMAs of a step run at the MFMA latency, not at its issue rate
    f32x4 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[s][0][i] = acc[s][1][i] = 0.0f;
    const __hip_bfloat16* brow = dgh_cur + seq * kDghPitch + 8 * q;
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
    f32x4 total[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) total[s] = acc[s][0] + acc[s][1];
    float mine[kU];
    own_values(total, mine);
#pragma unroll
    for (int i = 0; i < kU; ++i) dh[i] = keep[i] + mine[i];
  };

  GruStepIn<T> in_a, in_b;
  load_step(a.T - 1, in_a);
  load_step(a.T > 1 ? a.T - 2 : 0, in_b);
  // reverse of the forward processing order; first step peeled (see the forward kernel)
  step(a.T - 1, in_a);
  long t = a.T - 2;
#pragma unroll 1
  for (; t >= 1; t -= 2) {
    step(t, in_b);
    step(t - 1, in_a);
  }
  if (t == 0) step(0, in_b);

  // bias gradients: add the kSeqs sequences of the tile (lanes that differ in c % kSeqs), one group per slot
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < kU; ++i) {
      float v = bsum[k][i];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      if (kSeqs == 8) v += __shfl_xor(v, 4);
      bsum[k][i] = v;
    }
  if (seq == 0) {
    float* out = a.dbias + ((tile * 2 + dir) * 4) * kH + unit0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < kU; ++i) out[k * kH + i] = bsum[k][i];
  }
}

template <typename T>
__global__ __launch_bounds__(kGruThreads, 2) void gru_backward_kernel(GruBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* wn_lds = reinterpret_cast<bf16x8*>(smem);                               // k-range of the n gate
  __hip_bfloat16* dgh = reinterpret_cast<__hip_bfloat16*>(smem + kWnBytes);       // [2][kSeqs][kDghPitch]
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
  __syncthreads();
  gru_backward_steps<T>(a, wn_lds, dgh, wrz);
}

// ---- layout converters between the natural [B][T][2][NS][H] tensors of the host GEMMs and the tile layout ----
// One block per (tile, t, direction): kSeqs sequences x NS x (256 / kU) groups.  The tile side is accessed as one
// contiguous run per block; the natural side as short pieces of 512-B / 1-KB rows, every byte exactly once.
// HBM-bound permutes (read + write of the tensor), replacing strided framework copies.
__device__ __forceinline__ void lane_coords(int w, int lane, int& seq, int& ug) {
  seq = (lane & 15) % kSeqs;
  ug = lane_unit0(w, lane) / kU;                           // kU-unit group along H
}

template <int kGroupBytes>
__global__ __launch_bounds__(256) void gru_to_tile_kernel(const Raw<kGroupBytes>* __restrict__ src, long B, long T,
                                                          int ns, Raw<kGroupBytes>* __restrict__ dst) {
  const long blk = blockIdx.x;
  const int dir = static_cast<int>(blk & 1);
  const long t = (blk >> 1) % T, tile = (blk >> 1) / T;
  const int groups = 8 * ns * 64;
  Raw<kGroupBytes> zero;
  __builtin_memset(&zero, 0, sizeof(zero));
  for (int g = threadIdx.x; g < groups; g += 256) {
    const int lane = g & 63, slot = (g >> 6) % ns, w = (g >> 6) / ns;
    int seq, ug;
    lane_coords(w, lane, seq, ug);
    const long b = tile * kSeqs + seq;
    Raw<kGroupBytes> v = zero;
    if (b < B) v = src[(((b * T + t) * 2 + dir) * ns + slot) * (kH / kU) + ug];
    dst[blk * groups + g] = v;
  }
}

template <int kGroupBytes>
__global__ __launch_bounds__(256) void gru_from_pair_tile_kernel(const Raw<2 * kGroupBytes>* __restrict__ src, long B,
                                                                 long T, Raw<kGroupBytes>* __restrict__ dgi,
                                                                 Raw<kGroupBytes>* __restrict__ dghn) {
  const long blk = blockIdx.x;
  const int dir = static_cast<int>(blk & 1);
  const long t = (blk >> 1) % T, tile = (blk >> 1) / T;
  constexpr int kWords = kGroupBytes / 4;
  constexpr int kRow = kH / kU;                            // groups per H
  for (int g = threadIdx.x; g < 8 * 2 * 64; g += 256) {
    const int lane = g & 63, ps = (g >> 6) & 1, w = g >> 7;
    int seq, ug;
    lane_coords(w, lane, seq, ug);
    const long b = tile * kSeqs + seq;
    if (b >= B) continue;
    const Raw<2 * kGroupBytes> p = src[blk * (8 * 2 * 64) + g];
    Raw<kGroupBytes> lo, hi;
#pragma unroll
    for (int k = 0; k < kWords; ++k) {
      lo.w[k] = p.w[k];
      hi.w[k] = p.w[kWords + k];
    }
    const long row = (b * T + t) * 2 + dir;
    if (ps == 0) {
      dgi[(row * 3 + 0) * kRow + ug] = lo;                 // da_r
      dgi[(row * 3 + 1) * kRow + ug] = hi;                 // da_z
    } else {
      dgi[(row * 3 + 2) * kRow + ug] = lo;                 // da_n
      dghn[row * kRow + ug] = hi;                          // da_n * r
    }
  }
}

// h_{t-1} of the forward recurrence for every (sequence, t, direction): y shifted by one step in each direction's
// own time order, zero at the first step.  out[b][t][0] = y[b][t-1][0], out[b][t][1] = y[b][t+1][1].  One 16-byte
// piece per thread; replaces four strided framework copies and two fills per layer and iteration.
__global__ __launch_bounds__(256) void gru_previous_state_kernel(const uint4* __restrict__ y, long rows, long T,
                                                                 int pieces, uint4* __restrict__ out) {
  const long i = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;      // piece index in [rows][2][pieces]
  if (i >= rows * 2 * pieces) return;
  const long row = i / (2 * pieces);
  const int dir = static_cast<int>((i / pieces) & 1);
  const long t = row % T;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (dir == 0 ? t > 0 : t + 1 < T) v = y[dir == 0 ? i - 2 * pieces : i + 2 * pieces];
  out[i] = v;
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
  if (elem_bytes == 2) hipLaunchKernelGGL(gru_to_tile_kernel<2 * kU>, grid, dim3(256), 0, stream,
                                          static_cast<const Raw<2 * kU>*>(src), static_cast<long>(B),
                                          static_cast<long>(T), ns, static_cast<Raw<2 * kU>*>(dst));
  else hipLaunchKernelGGL(gru_to_tile_kernel<4 * kU>, grid, dim3(256), 0, stream, static_cast<const Raw<4 * kU>*>(src),
                          static_cast<long>(B), static_cast<long>(T), ns, static_cast<Raw<4 * kU>*>(dst));
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
  if (elem_bytes == 2) hipLaunchKernelGGL(gru_from_pair_tile_kernel<2 * kU>, grid, dim3(256), 0, stream,
                                          static_cast<const Raw<4 * kU>*>(dg_tile), static_cast<long>(B),
                                          static_cast<long>(T), static_cast<Raw<2 * kU>*>(dgi),
                                          static_cast<Raw<2 * kU>*>(dghn));
  else hipLaunchKernelGGL(gru_from_pair_tile_kernel<4 * kU>, grid, dim3(256), 0, stream,
                          static_cast<const Raw<8 * kU>*>(dg_tile), static_cast<long>(B), static_cast<long>(T),
                          static_cast<Raw<4 * kU>*>(dgi), static_cast<Raw<4 * kU>*>(dghn));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int seld_gru_previous_state(const void* y, int elem_bytes, int64_t B, int64_t T, void* h_prev, void* stream_) {
  using namespace seld;
  if (!current_state()) return kErrNotInitialised;
  if (B <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_previous_state: B and T must be positive");
  if (elem_bytes != 2 && elem_bytes != 4) return fail(kErrUnsupported, "seld_gru_previous_state: 2- or 4-byte elements");
  if (!y || !h_prev) return fail(kErrInvalidArgument, "seld_gru_previous_state: null pointer");
  const int pieces = kH * elem_bytes / 16;
  const long total = static_cast<long>(B) * T * 2 * pieces;
  hipLaunchKernelGGL(gru_previous_state_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), static_cast<const uint4*>(y), static_cast<long>(B) * T,
                     static_cast<long>(T), pieces, static_cast<uint4*>(h_prev));
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

int64_t seld_gru_tile_rows(void) { return seld::kSeqs; }

int seld_gru_forward(const void* gi, int is_bf16, const void* w_hh_bf16, const float* b_hn, int64_t B,
                     int64_t T, int64_t H, void* y, void* saved_tile, void* stream_) {
  using namespace seld;
  DeviceState* st = current_state();
  if (!st) return kErrNotInitialised;
  if (H != kH) return fail(kErrUnsupported, "seld_gru_forward: built for hidden size 256 (config.py:45)");
  if (B <= 0 || T <= 0) return fail(kErrInvalidArgument, "seld_gru_forward: B and T must be positive");
  if (!gi || !w_hh_bf16 || !b_hn || !y) return fail(kErrInvalidArgument, "seld_gru_forward: null pointer");
  const long tiles = (B + kSeqs - 1) / kSeqs;
  GruFwdArgs a{gi, static_cast<const __hip_bfloat16*>(w_hh_bf16), b_hn, y, saved_tile, tiles, T, B};
  const dim3 grid(static_cast<unsigned>(tiles), 2);
  const size_t lds = kWnBytes + 2 * kSeqs * kHPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (need_lds(st, kAttrGruForward)) {     // once per device (seld_common.h)
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_forward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    lds_attr_set(st, kAttrGruForward);
  }
  // (Round 2 measured two variants against this kernel and dropped them: all of W_hh in registers with four wavefronts,
  // 36 % slower -- one wavefront per SIMD has nothing to run in the shadow of its own MFMAs; gate-major MFMA order, no
  // change.  DESIGN.md 5.4.)
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
  const size_t lds = kWnBytes + 2 * kSeqs * kDghPitch * sizeof(__hip_bfloat16);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (need_lds(st, kAttrGruBackward)) {
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<__hip_bfloat16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    SELD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gru_backward_kernel<float>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    lds_attr_set(st, kAttrGruBackward);
  }
  if (is_bf16) hipLaunchKernelGGL(gru_backward_kernel<__hip_bfloat16>, grid, dim3(kGruThreads), lds, stream, a);
  else hipLaunchKernelGGL(gru_backward_kernel<float>, grid, dim3(kGruThreads), lds, stream, a);
  SELD_HIP_TRY(hipGetLastError());
  return kOk;
}

}  // extern "C"
