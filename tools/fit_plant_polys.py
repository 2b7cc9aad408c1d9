"""Developer tool (CPU): coefficients of the range-specialised FP64 kernels of the plant integrator (lmpc_kernels.hip.h: plant_atan, plant_sin1) and their
measured accuracy.  Chebyshev interpolation in extended precision (numpy.longdouble, 64-bit mantissa), coefficients rounded to double, error measured in
double arithmetic with the kernel's own evaluation order (Estrin) against the extended-precision libm.

    python tools/fit_plant_polys.py            # prints the C arrays and the error table (profiles/r5_plant_polys.txt)
"""
import numpy as np

LD = np.longdouble


def cheb_fit(f, a, b, n):
    """degree n-1 interpolant of f on [a, b] at Chebyshev nodes, returned as monomial coefficients in w (extended precision)."""
    k = np.arange(n, dtype=LD)
    t = np.cos(LD(np.pi) * (k + LD(0.5)) / LD(n))
    w = (LD(a) + LD(b)) / 2 + (LD(b) - LD(a)) / 2 * t
    V = np.vander(w, n, increasing=True)
    # solve in extended precision by QR-free Gaussian elimination with partial pivoting (numpy.linalg has no longdouble path)
    A = np.hstack([V, f(w)[:, None]]).astype(LD)
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]
        A[i] = A[i] / A[i, i]
        for r in range(n):
            if r != i:
                A[r] = A[r] - A[r, i] * A[i]
    return A[:, n]


def estrin(c, w):
    """the kernel's evaluation order, in double"""
    c = [np.float64(v) for v in c]
    lvl = [np.asarray(x, np.float64) * np.ones_like(w) for x in c]
    p = w.copy()
    while len(lvl) > 1:
        nxt = []
        for i in range(0, len(lvl) - 1, 2):
            nxt.append(lvl[i] + lvl[i + 1] * p)          # (an FMA in the kernel: one rounding instead of two -- the measured error is an upper bound)
        if len(lvl) % 2:
            nxt.append(lvl[-1])
        lvl = nxt; p = p * p
    return lvl[0]


def report(name, c, f_exact, xs, form):
    x = xs.astype(np.float64)
    w = x * x
    got = form(x, estrin(c, w))
    ref = f_exact(xs.astype(LD))
    err = np.abs(got.astype(LD) - ref)
    rel = err / np.maximum(np.abs(ref), LD(1e-300))
    print("// %s: %d coefficients, max abs err %.2e, max rel err %.2e (%d points)" % (name, len(c), float(err.max()), float(rel[np.abs(ref) > 1e-8].max()), len(x)))
    print("static __device__ const double %s[%d] = {%s};" % (name, len(c), ", ".join("%.17e" % float(np.float64(v)) for v in c)))


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    # atan(z) = z P(z^2), |z| <= 1
    n_at = 23
    c = cheb_fit(lambda w: np.where(w > 0, np.arctan(np.sqrt(np.maximum(w, LD(1e-40)))) / np.sqrt(np.maximum(w, LD(1e-40))), LD(1)), 0.0, 1.0, n_at)
    xs = np.concatenate([rng.uniform(-1, 1, 400000), np.linspace(-1, 1, 20001), rng.uniform(-1e-3, 1e-3, 1000)])
    report("PLANT_ATAN_C", c, np.arctan, xs, lambda x, p: x * p)
    # sin(x) = x Q(x^2), |x| <= 1
    n_s = 10
    c = cheb_fit(lambda w: np.where(w > 0, np.sin(np.sqrt(np.maximum(w, LD(1e-40)))) / np.sqrt(np.maximum(w, LD(1e-40))), LD(1)), 0.0, 1.0, n_s)
    report("PLANT_SIN1_C", c, np.sin, xs, lambda x, p: x * p)
    # sin / cos on |r| <= pi / 4 after the two-constant Cody-Waite reduction: sin r = r + r^3 S(r^2), cos r = 1 - r^2 / 2 + r^4 C(r^2)
    a = (np.pi / 4) ** 2 * 1.02
    cs = cheb_fit(lambda w: np.where(w > 0, (np.sin(np.sqrt(np.maximum(w, LD(1e-40)))) / np.sqrt(np.maximum(w, LD(1e-40))) - 1) / np.maximum(w, LD(1e-40)), LD(-1) / 6), 0.0, a, 7)
    cc = cheb_fit(lambda w: np.where(w > 0, (np.cos(np.sqrt(np.maximum(w, LD(1e-40)))) - 1 + w / 2) / np.maximum(w, LD(1e-40)) ** 2, LD(1) / 24), 0.0, a, 7)
    xr = np.concatenate([rng.uniform(-np.pi / 4, np.pi / 4, 400000), np.linspace(-np.pi / 4, np.pi / 4, 20001)])
    report("PLANT_SINK_C", cs, np.sin, xr, lambda x, p: x + x * (x * x) * p)
    report("PLANT_COSK_C", cc, np.cos, xr, lambda x, p: (1.0 - 0.5 * (x * x)) + (x * x) * (x * x) * p)
