#!/usr/bin/env python3
"""Generate tests/golden/g9_exact_dot.npz: known-answer vectors for the canonical score  RN32(exact q.x).

Expected values come from Python rationals (fractions.Fraction: exact products, exact sum) rounded ONCE to fp32 with integer
arithmetic — no C, no numpy arithmetic, no GPU code involved — so they pin BOTH the C oracle (oracle/search_c.c) and the HIP
re-rank / exact-scan kernels (tests/test_gpu_search.py) to the mathematical definition.

Run:  python tests/golden/make_exact_dot_golden.py        (pure Python; ~1 minute)

Cases (inputs stored as fp32 bit patterns, ragged lengths via offsets):
  * random unit vectors, d in {1 .. 1024}, and the metric's d = 1024
  * wide exponent spreads (elements scaled by 2^[-60, 60]) and heavy cancellation (sum << sum of magnitudes)
  * exact midpoints between two adjacent floats (round-to-even decides), midpoint +- a tiny term (sticky bit decides)
  * results in the fp32 subnormal range, results that round to +-0, results that overflow to inf
"""
import os
import sys
from fractions import Fraction

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle.search_np import dot_fraction  # noqa: E402  (pure-Python Fraction arithmetic; the C library is not touched)

rng = np.random.Generator(np.random.PCG64(99))
cases = []


def add(q, x):
    q = np.asarray(q, np.float32); x = np.asarray(x, np.float32)
    assert q.shape == x.shape and np.isfinite(q).all() and np.isfinite(x).all()
    cases.append((q, x))


def unit(d):
    v = rng.standard_normal(d).astype(np.float32)
    return (v / np.linalg.norm(v)).astype(np.float32)


for d in (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 384, 500, 768, 1000, 1023, 1024):
    for _ in range(4):
        add(unit(d), unit(d))
for _ in range(40):                                                 # the metric's dimension
    add(unit(1024), unit(1024))
for _ in range(40):                                                 # wide exponent spread
    d = int(rng.integers(2, 200))
    add(rng.standard_normal(d).astype(np.float32) * np.float32(2.0) ** rng.integers(-60, 60, d).astype(np.float32),
        rng.standard_normal(d).astype(np.float32) * np.float32(2.0) ** rng.integers(-40, 40, d).astype(np.float32))
for _ in range(40):                                                 # cancellation: x = [a, -a] pattern plus a small rest
    d = int(rng.integers(2, 100)) * 2
    a = rng.standard_normal(d // 2).astype(np.float32) * np.float32(1e3)
    q = np.concatenate([a, a]); x = np.concatenate([np.ones(d // 2, np.float32), -np.ones(d // 2, np.float32)])
    x[: d // 2] += (rng.standard_normal(d // 2) * 1e-6).astype(np.float32)
    add(q, x)
one = np.float32(1.0)
for e in (-24, -25, -23):                                           # 1 + 2^-24 is the exact midpoint of 1 and 1 + 2^-23
    add([1.0, 2.0 ** e], [1.0, 1.0])
    add([1.0, 2.0 ** e, 2.0 ** -70], [1.0, 1.0, 1.0])               # just above the midpoint
    add([1.0, 2.0 ** e, -(2.0 ** -70)], [1.0, 1.0, 1.0])            # just below
    add([-1.0, -(2.0 ** e), 2.0 ** -70], [1.0, 1.0, 1.0])
add([1.0 + 2.0 ** -23, 2.0 ** -24], [1.0, 1.0])                     # midpoint with an odd lower neighbour: rounds up to even
add([3.0, 2.0 ** -23], [0.5, 1.0])
for _ in range(30):                                                 # random midpoints: (m + 1/2) * 2^e with a 24-bit m, split over terms
    m = int(rng.integers(2 ** 23, 2 ** 24)); e = int(rng.integers(-100, 80))
    hi = np.float32(m) * np.float32(2.0) ** np.float32(e)
    half = np.float32(2.0) ** np.float32(e - 1)
    if np.isfinite(hi) and hi != 0 and half != 0:
        add([hi, half], [1.0, 1.0])
        add([hi, half, half * np.float32(2.0 ** -40)], [1.0, 1.0, 1.0])
for _ in range(30):                                                 # subnormal results and results that vanish
    d = int(rng.integers(1, 20))
    add(rng.standard_normal(d).astype(np.float32) * np.float32(1e-22), rng.standard_normal(d).astype(np.float32) * np.float32(1e-22))
add([2.0 ** -75], [2.0 ** -75])                                     # 2^-150: the midpoint of 0 and the smallest subnormal -> +0
add([2.0 ** -75, 2.0 ** -100], [2.0 ** -75, 2.0 ** -100])           # just above it -> 2^-149
add([-(2.0 ** -75)], [2.0 ** -75])                                  # -> -0
add([2.0 ** -100], [2.0 ** -100])                                   # -> +0 (non-zero exact value)
add([1e-45, -1e-45], [1e-45, 1e-45])                                # exact zero from subnormal products -> +0
add([0.0, -0.0, 0.0], [1.0, 5.0, -2.0])                             # exact zero -> +0
add([3e38, 3e38], [1.0, 1.0])                                       # overflow -> +inf
add([-3e38, -3e38, 1.0], [1.0, 1.0, 1.0])                           # -> -inf
add([3.4028234e38, 2.0 ** 103], [1.0, 1.0])                         # FLT_MAX + half an ulp: the midpoint to 2^128 -> inf (round to even)
add([3.4028234e38, 2.0 ** 102], [1.0, 1.0])                         # FLT_MAX + a quarter ulp -> FLT_MAX
add([2.0 ** 100, 1.0, -(2.0 ** 100)], [2.0 ** 20, 2.0 ** -120, 2.0 ** 20])      # huge spread with total cancellation of the big terms

qs = np.concatenate([c[0] for c in cases]).view(np.uint32)
xs = np.concatenate([c[1] for c in cases]).view(np.uint32)
off = np.cumsum([0] + [len(c[0]) for c in cases]).astype(np.int64)
exp = np.array([dot_fraction(q, x) for q, x in cases], np.float32).view(np.uint32)
out = os.path.join(REPO, "tests", "golden", "g9_exact_dot.npz")
np.savez_compressed(out, q_bits=qs, x_bits=xs, offsets=off, expected_bits=exp)
print(f"{len(cases)} cases -> {out} ({os.path.getsize(out)} bytes)")
