// A complex fp32 value as ONE register pair, and the five packed operations the log-mel kernel is written in.
//
// gfx950 has packed fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: both halves of an even-aligned
// 64-bit register pair per lane in one issue slot; op_sel / op_sel_hi pick which half of each source feeds each half of
// the result, neg_lo / neg_hi negate per half).  A complex value kept as (re, im) in such a pair makes every butterfly
// one instruction and multiplication by +-i free (a swap is an op_sel).  On the device `cf` is clang's 2-vector, which
// the AMDGPU back end keeps in a pair and for which it folds swaps, splats and negations into the modifiers; the host
// build (the CPU lane emulator of tests/emu/, g++) is a plain struct with the SAME arithmetic written out with fmaf.
#pragma once

#include <math.h>

#ifndef SELD_HD
#if defined(__HIPCC__)
#define SELD_HD __host__ __device__ __forceinline__
#else
#define SELD_HD inline
#endif
#endif

namespace seld {

#if defined(__HIP_DEVICE_COMPILE__)

typedef float cf __attribute__((ext_vector_type(2)));

SELD_HD cf cf_make(float x, float y) { cf r = {x, y}; return r; }
SELD_HD cf cf_add(cf a, cf b) { return a + b; }
SELD_HD cf cf_sub(cf a, cf b) { return a - b; }
SELD_HD cf cf_scale(cf a, float s) { return a * s; }
SELD_HD cf cf_mul(cf a, cf b) { return a * b; }                                         // per half
SELD_HD cf cf_fma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }      // per half
// (a.x s + b.x, a.y s + b.y)
SELD_HD cf cf_fma_splat(cf a, float s, cf b) { return __builtin_elementwise_fma(a, cf_make(s, s), b); }
// (a.y cx + b.x, a.x cy + b.y): the swapped source is an op_sel, not an instruction
SELD_HD cf cf_fma_swap(cf a, float cx, float cy, cf b) {
  return __builtin_elementwise_fma(__builtin_shufflevector(a, a, 1, 0), cf_make(cx, cy), b);
}
SELD_HD cf cf_mul_swap(cf a, float cx, float cy) { return __builtin_shufflevector(a, a, 1, 0) * cf_make(cx, cy); }

#else

struct cf { float x, y; };

SELD_HD cf cf_make(float x, float y) { cf r = {x, y}; return r; }
SELD_HD cf cf_add(cf a, cf b) { return cf_make(a.x + b.x, a.y + b.y); }
SELD_HD cf cf_sub(cf a, cf b) { return cf_make(a.x - b.x, a.y - b.y); }
SELD_HD cf cf_scale(cf a, float s) { return cf_make(a.x * s, a.y * s); }
SELD_HD cf cf_mul(cf a, cf b) { return cf_make(a.x * b.x, a.y * b.y); }
SELD_HD cf cf_fma(cf a, cf b, cf c) { return cf_make(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
SELD_HD cf cf_fma_splat(cf a, float s, cf b) { return cf_make(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y)); }
SELD_HD cf cf_fma_swap(cf a, float cx, float cy, cf b) { return cf_make(fmaf(a.y, cx, b.x), fmaf(a.x, cy, b.y)); }
SELD_HD cf cf_mul_swap(cf a, float cx, float cy) { return cf_make(a.y * cx, a.x * cy); }

#endif

// complex product a * c
SELD_HD cf cf_cmul(cf a, cf c) {
  const cf t = cf_scale(a, c.x);
  return cf_fma_swap(a, -c.y, c.y, t);
}

}  // namespace seld
