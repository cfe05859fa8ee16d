#!/usr/bin/env python3
"""Derive the atan polynomial used by csrc/planner.hip: atan(t) = t + t*z*Q(z), z = t*t, t in [0, 1].

Q is the degree-N Chebyshev interpolant of g(z) = (atan(sqrt z)/sqrt z - 1)/z on [0, 1] computed in
80-bit long double, converted to the monomial basis and rounded to float64.  Prints the coefficients as
C hex-float literals and the observed error of the float64 Horner evaluation against long-double atan.
"""
import sys
import numpy as np

LD = np.longdouble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 22


def g(z):
    z = np.asarray(z, LD)
    t = np.sqrt(z)
    small = z < LD(1e-4)
    zs = np.where(small, z, LD(0.5))
    # series near 0 (avoids 0/0 and cancellation): -1/3 + z/5 - z^2/7 + z^3/9 - z^4/11
    ser = -LD(1) / 3 + zs * (LD(1) / 5 + zs * (-LD(1) / 7 + zs * (LD(1) / 9 - zs / 11)))
    tt = np.where(small, LD(1), t)
    zz = np.where(small, LD(1), z)
    full = (np.arctan(tt) / tt - 1) / zz
    return np.where(small, ser, full)


def cheb_to_mono(c):
    """Chebyshev series on w in [-1,1] -> monomial coefficients in z, w = 2z - 1 (long double)."""
    n = len(c)
    T = [np.zeros(n, LD) for _ in range(n)]      # T[k] = coefficients of T_k(w) in powers of w
    T[0][0] = 1
    if n > 1:
        T[1][1] = 1
    for k in range(2, n):
        T[k][1:] = 2 * T[k - 1][:-1]
        T[k] -= T[k - 2]
    pw = sum(c[k] * T[k] for k in range(n))       # polynomial in w
    # substitute w = 2z - 1
    out = np.zeros(n, LD)
    binom = [[LD(1)]]
    for i in range(1, n):
        row = [LD(1)] + [binom[-1][j - 1] + binom[-1][j] for j in range(1, i)] + [LD(1)]
        binom.append(row)
    for i in range(n):
        for j in range(i + 1):
            out[j] += pw[i] * binom[i][j] * (LD(2) ** j) * (LD(-1) ** (i - j))
    return out


def main():
    n = N + 1
    k = np.arange(n, dtype=LD)
    nodes = np.cos(np.pi * (k + LD(0.5)) / n)                  # Chebyshev nodes in w
    fz = g((nodes + 1) / 2)
    c = np.array([(2 / LD(n)) * np.sum(fz * np.cos(np.pi * j * (k + LD(0.5)) / n)) for j in range(n)], LD)
    c[0] /= 2
    mono = cheb_to_mono(c)
    q = mono.astype(np.float64)
    rng = np.random.RandomState(0)
    t = np.concatenate([rng.uniform(0, 1, 2_000_000), np.linspace(0, 1, 200_001), 10.0 ** rng.uniform(-12, 0, 500_000)])
    t = np.minimum(t, 1.0)
    z = t * t
    p = np.full_like(t, q[-1])
    for coef in q[-2::-1]:
        p = p * z + coef                      # float64 Horner (the kernel uses fma; this bounds it from above)
    a = t + (t * z) * p
    ref = np.arctan(t.astype(LD))
    err = np.abs(a.astype(LD) - ref)
    ulp = np.spacing(np.maximum(ref.astype(np.float64), 1e-300))
    print("degree", N, "max abs err %.3e" % float(err.max()), "max err in ulp %.3f" % float((err / ulp).max()))
    for i, v in enumerate(q):
        print("    %s,   // z^%d" % (float(v).hex(), i))


if __name__ == "__main__":
    main()
