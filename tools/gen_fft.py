#!/usr/bin/env python3
"""Generate straight-line in-register DFT routines for the 960-point STFT.

The log-mel kernel factors N = 960 = 32 x 30 (one half-wavefront per transform):
stage A is a 32-point DFT held in the registers of one lane, stage B a 30-point DFT.
This script emits both as fully unrolled ``__host__ __device__`` C++ (static indices only,
so every element stays in a VGPR -- no scratch) into
``sound-event-localization-detection_amd/csrc/fft_gen.h``.

  * 32 = 2^5        : radix-2 decimation-in-time recursion.
  * 30 = 2 x 3 x 5  : Good-Thomas prime-factor mapping (no twiddles at all) over hand
                      written 2/3/5-point butterflies.

Every value is a COMPLEX number held as one (re, im) register pair (``cf``, seld_complex.h) and every emitted operation is
one packed-fp32 instruction on gfx950 (``v_pk_add_f32`` / ``v_pk_mul_f32`` / ``v_pk_fma_f32``: two flops per lane and
issue slot).  Round 1 emitted scalar code over separate re[] / im[] arrays and left the pairing to the compiler's SLP
vectoriser: it found the packed instructions (494 of them in the two DFT stages) but had to assemble and take apart
the register pairs around them with 338 ``v_mov_b32`` -- 40 % of the stage's issue slots.  Multiplications by
+-i and -1 cost nothing: they are carried as a pending rotation of the value and folded into the consumer
(a + (-i) b  =  fma(swap(b), (1, -1), a): the swap is the instruction's op_sel modifier, the signs a constant pair).

Forward transform, e^{-2 pi i nk/N}.  Output in natural order.
Run:  python tools/gen_fft.py   (idempotent; the header is committed)
"""
from __future__ import annotations

import cmath
import math
from pathlib import Path

OUT = Path(__file__).resolve().parent.parent / "sound-event-localization-detection_amd" / "csrc" / "fft_gen.h"

ROT = {0: 1 + 0j, 1: -1j, 2: -1 + 0j, 3: 1j}     # pending rotation index r: value = (-i)^r * name


def lit(v: float) -> str:
    r = repr(float(v))
    return r + "f"


class Emitter:
    """Three-address straight-line code over complex values (name, rot): the value is (-i)^rot * name."""

    def __init__(self):
        self.lines = []
        self.count = 0
        self.ops = 0

    def new(self, expr: str, ops: int = 1) -> str:
        name = f"t{self.count}"
        self.count += 1
        self.ops += ops
        self.lines.append(f"  const cf {name} = {expr};")
        return name

    # a + (-i)^k b  as one packed instruction
    def _add_rot(self, na, nb, k):
        k %= 4
        if k == 0:
            return self.new(f"cf_add({na}, {nb})")
        if k == 2:
            return self.new(f"cf_sub({na}, {nb})")
        if k == 1:      # a - i b = (a.x + b.y, a.y - b.x)
            return self.new(f"cf_fma_swap({nb}, 1.0f, -1.0f, {na})")
        return self.new(f"cf_fma_swap({nb}, -1.0f, 1.0f, {na})")      # a + i b

    def add(self, a, b):
        (na, ra), (nb, rb) = a, b
        return (self._add_rot(na, nb, rb - ra), ra % 4)

    def sub(self, a, b):
        (na, ra), (nb, rb) = a, b
        return (self._add_rot(na, nb, rb - ra + 2), ra % 4)

    def neg_i(self, a):
        return (a[0], (a[1] + 1) % 4)

    def pos_i(self, a):
        return (a[0], (a[1] + 3) % 4)

    def scale(self, a, s):        # real scalar
        return (self.new(f"cf_scale({a[0]}, {lit(s)})"), a[1])

    def axpy(self, s, a, b):      # s*a + b (complex a, b; real s)
        (na, ra), (nb, rb) = a, b
        k = (ra - rb) % 4         # value = (-i)^rb (nb + s (-i)^k na)
        if k == 0:
            return (self.new(f"cf_fma_splat({na}, {lit(s)}, {nb})"), rb)
        if k == 2:
            return (self.new(f"cf_fma_splat({na}, {lit(-s)}, {nb})"), rb)
        if k == 1:                # s (-i) a = s (a.y, -a.x)
            return (self.new(f"cf_fma_swap({na}, {lit(s)}, {lit(-s)}, {nb})"), rb)
        return (self.new(f"cf_fma_swap({na}, {lit(-s)}, {lit(s)}, {nb})"), rb)

    def mul_const(self, a, w: complex):
        na, ra = a
        eps = 1e-12
        for k, r in ROT.items():
            if abs(w - r) < eps:
                return (na, (ra + k) % 4)
        # (x + i y)(c + i d) = (x c - y d, y c + x d) = fma(swap(a), (-d, d), a * c)
        t = self.new(f"cf_scale({na}, {lit(w.real)})")
        return (self.new(f"cf_fma_swap({na}, {lit(-w.imag)}, {lit(w.imag)}, {t})"), ra)

    def plain(self, a):
        """Materialise a pending rotation (only the outputs of a routine need it)."""
        na, ra = a
        ra %= 4
        if ra == 0:
            return na
        if ra == 2:
            return self.new(f"cf_scale({na}, -1.0f)")
        if ra == 1:               # (-i)(x, y) = (y, -x)
            return self.new(f"cf_mul_swap({na}, 1.0f, -1.0f)")
        return self.new(f"cf_mul_swap({na}, -1.0f, 1.0f)")

    # -- butterflies -----------------------------------------------------------------
    def dft2(self, x):
        return [self.add(x[0], x[1]), self.sub(x[0], x[1])]

    def dft3(self, x):
        s = math.sin(2 * math.pi / 3)
        t1 = self.add(x[1], x[2])
        y0 = self.add(x[0], t1)
        m1 = self.axpy(-0.5, t1, x[0])
        d = self.sub(x[1], x[2])
        m2 = self.neg_i(self.scale(d, s))           # -i*s*(x1-x2)
        return [y0, self.add(m1, m2), self.sub(m1, m2)]

    def dft5(self, x):
        c1, c2 = math.cos(2 * math.pi / 5), math.cos(4 * math.pi / 5)
        s1, s2 = math.sin(2 * math.pi / 5), math.sin(4 * math.pi / 5)
        a = self.add(x[1], x[4])
        b = self.sub(x[1], x[4])
        c = self.add(x[2], x[3])
        d = self.sub(x[2], x[3])
        y0 = self.add(x[0], self.add(a, c))
        p1 = self.axpy(c2, c, self.axpy(c1, a, x[0]))
        p2 = self.axpy(c1, c, self.axpy(c2, a, x[0]))
        q1 = self.axpy(s2, d, self.scale(b, s1))     # s1 b + s2 d
        q2 = self.axpy(-s1, d, self.scale(b, s2))    # s2 b - s1 d
        iq1 = self.pos_i(q1)
        iq2 = self.pos_i(q2)
        return [y0, self.sub(p1, iq1), self.sub(p2, iq2), self.add(p2, iq2), self.add(p1, iq1)]

    # -- composite -------------------------------------------------------------------
    def dft(self, x):
        n = len(x)
        if n == 1:
            return list(x)
        if n == 2:
            return self.dft2(x)
        if n == 3:
            return self.dft3(x)
        if n == 5:
            return self.dft5(x)
        for n1 in (2, 3, 5):
            if n % n1 == 0:
                n2 = n // n1
                if math.gcd(n1, n2) == 1:
                    return self.pfa(x, n1, n2)
        if n % 2 == 0:
            return self.cooley_tukey(x, 2, n // 2)
        raise ValueError(n)

    def cooley_tukey(self, x, n1, n2):
        """n = n2*a + b (a<n1, b<n2), k = k1 + n1*k2:
        X[k1+n1 k2] = sum_b W_n^{b k1} W_n2^{b k2} sum_a x[n2 a + b] W_n1^{a k1}."""
        n = n1 * n2
        inner = [self.dft([x[n2 * a + b] for a in range(n1)]) for b in range(n2)]   # [b][k1]
        out = [None] * n
        for k1 in range(n1):
            col = [self.mul_const(inner[b][k1], cmath.exp(-2j * math.pi * b * k1 / n)) for b in range(n2)]
            y = self.dft(col)
            for k2 in range(n2):
                out[k1 + n1 * k2] = y[k2]
        return out

    def pfa(self, x, n1, n2):
        """Good-Thomas: n = (n2 a + n1 b) mod N ; k = (k1 n2 (n2^-1 mod n1) + k2 n1 (n1^-1 mod n2)) mod N."""
        n = n1 * n2
        inv2 = pow(n2, -1, n1)
        inv1 = pow(n1, -1, n2)
        grid = [[x[(n2 * a + n1 * b) % n] for b in range(n2)] for a in range(n1)]
        cols = [self.dft([grid[a][b] for a in range(n1)]) for b in range(n2)]       # [b][k1]
        out = [None] * n
        for k1 in range(n1):
            y = self.dft([cols[b][k1] for b in range(n2)])                          # [k2]
            for k2 in range(n2):
                out[(k1 * n2 * inv2 + k2 * n1 * inv1) % n] = y[k2]
        return out


def gen_function(n: int) -> tuple[str, int]:
    e = Emitter()
    xin = [(f"z[{i}]", 0) for i in range(n)]
    y = e.dft(xin)
    names = [e.plain(v) for v in y]
    body = "\n".join(e.lines)
    stores = "\n".join(f"  z[{k}] = {names[k]};" for k in range(n))
    src = (f"// {n}-point forward DFT, in place, natural order in and out ({e.ops} packed operations)\n"
           f"SELD_HD void dft{n}(cf (&z)[{n}]) {{\n{body}\n{stores}\n}}\n")
    return src, e.ops


def main():
    parts = []
    for n in (32, 30):
        src, ops = gen_function(n)
        parts.append(src)
        print(f"dft{n}: {ops} packed operations")
    header = (
        "// GENERATED by tools/gen_fft.py -- do not edit by hand.\n"
        "// Straight-line in-register DFTs for the 960 = 32 x 30 point STFT of the log-mel kernel.\n"
        "#pragma once\n"
        "#include \"seld_complex.h\"\n\n"
        "namespace seld {\n\n" + "\n".join(parts) + "\n}  // namespace seld\n")
    OUT.write_text(header)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
