"""Generates the polynomial coefficients used by the device math in voc_kernels.h (atan on
[-tan(pi/8), tan(pi/8)]), by Chebyshev-node interpolation in 60-digit arithmetic followed by a few
Remez exchange steps, and reports the error of the double-precision evaluation.
Run: python tools/gen_poly.py"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def fit(g, a, b, deg, iters=6):
    n = deg + 2
    # initial reference: Chebyshev extrema
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * i / (n - 1)) for i in range(n)][::-1]
    for _ in range(iters):
        A = mp.matrix(n, n)
        rhs = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for j in range(deg + 1):
                A[i, j] = x ** j
            A[i, deg + 1] = (-1) ** i
            rhs[i] = g(x)
        sol = mp.lu_solve(A, rhs)
        coef = [sol[j] for j in range(deg + 1)]
        err = lambda x: sum(c * x ** j for j, c in enumerate(coef)) - g(x)
        # new reference: local extrema of the error on a fine grid
        grid = [a + (b - a) * i / 4000 for i in range(4001)]
        ev = [err(x) for x in grid]
        ext = [grid[0]]
        for i in range(1, 4000):
            if (ev[i] - ev[i - 1]) * (ev[i + 1] - ev[i]) <= 0:
                ext.append(grid[i])
        ext.append(grid[-1])
        if len(ext) < n:
            break
        # keep n alternating extrema with largest magnitude
        while len(ext) > n:
            mags = [abs(err(x)) for x in ext]
            k = mags.index(min(mags))
            ext.pop(k)
        xs = ext
    return coef, max(abs(e) for e in ev)


def main():
    tmax = mp.tan(mp.pi / 8) * (1 + mp.mpf(2) ** -40)
    zmax = tmax ** 2

    def g(z):
        if z == 0:
            return mp.mpf(-1) / 3
        t = mp.sqrt(z)
        return (mp.atan(t) / t - 1) / z

    for deg in (9, 10, 11, 12):
        coef, e = fit(g, mp.mpf(0), zmax, deg)
        print(f"deg {deg}: max |poly - g| = {mp.nstr(e, 3)}  -> relative error of atan ~ {mp.nstr(e * zmax, 3)}")
        if deg == 10:
            best = coef
    print("atan: atan(t) = t + t*z*P(z), z = t*t, |t| <= tan(pi/8); P coefficients, low order first:")
    c64 = [float(c) for c in best]
    for c in c64:
        print(f"    {c!r},  // {c.hex()}")
    # double-precision check (Horner, no fma)
    rng = np.random.default_rng(0)
    t = rng.uniform(-float(tmax), float(tmax), 200000)
    z = t * t
    p = np.zeros_like(z)
    for c in c64[::-1]:
        p = p * z + c
    got = t + t * z * p
    ref = np.array([float(mp.atan(mp.mpf(v))) for v in t[:20000]])
    print("double evaluation: max abs err", np.abs(got[:20000] - ref).max(), "max rel", np.abs((got[:20000] - ref) / ref).max())


if __name__ == "__main__":
    main()
