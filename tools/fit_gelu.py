"""Measures, in float32 arithmetic emulated with numpy, the error of csrc/mirx_common.h:gelu_tanh (the sigmoid form) against
float64, next to what 0.5 v (1 + tanh u) gives with a correctly rounded tanh; and fits / measures the branch-free erf that was
tried for gelu_erf and not kept (it was not faster than ocml's erff: mirx_common.h).
  python tools/fit_gelu.py          (CPU, numpy + scipy)"""
import numpy as np
from scipy.special import erf, erfc

f32 = np.float32


def fit():
    za = np.cos(np.pi * (np.arange(400) + 0.5) / 400) * 0.5 + 0.5            # Chebyshev nodes in (0, 1)
    a = np.polyfit(za * za, erf(za) / za - 1, 6)                             # erf(z) = z + z A(z^2)
    zb = np.cos(np.pi * (np.arange(600) + 0.5) / 600) * 1.5 + 2.5            # ... in (1, 4)
    b = np.polyfit(zb, -np.log(erfc(zb)), 8)                                 # erf(z) = 1 - exp(-B(z))
    return a.astype(f32), b.astype(f32)


def erf_poly(z, a, b):
    z = z.astype(f32)
    az = np.abs(z)
    u = (az * az).astype(f32)
    p = np.zeros_like(u) + a[0]
    for c in a[1:]:
        p = (p * u + c).astype(f32)
    ra = (az * p + az).astype(f32)
    zc = np.minimum(az, f32(4))
    q = np.zeros_like(zc) + b[0]
    for c in b[1:]:
        q = (q * zc + c).astype(f32)
    rb = (f32(1) - np.exp2((q * f32(-1.4426950408889634)).astype(f32)).astype(f32)).astype(f32)
    return np.copysign(np.where(az < 1, ra, rb), z)


def main():
    a, b = fit()
    print("A (highest power first):", ", ".join("%.9e" % c for c in a))
    print("B (highest power first):", ", ".join("%.9e" % c for c in b))
    t = np.linspace(-9, 9, 2000001)
    v = t.astype(f32)
    z = (v * f32(0.70710678118654752)).astype(f32)
    ref = 0.5 * t * (1 + erf(t / np.sqrt(2)))
    g = (f32(0.5) * v * (f32(1) + erf_poly(z, a, b))).astype(f32)
    g0 = (f32(0.5) * v * (f32(1) + erf(z.astype(np.float64)).astype(f32))).astype(f32)
    print("erf_poly vs float64 erf: max abs %.3e" % np.abs(erf_poly(z, a, b).astype(np.float64) - erf(z.astype(np.float64))).max())
    print("gelu_erf:  max abs %.3e   (with a correctly rounded erf: %.3e)" % (np.abs(g - ref).max(), np.abs(g0 - ref).max()))

    t = np.linspace(-12, 12, 2400001)
    v = t.astype(f32)
    k1 = f32(-2 * 0.7978845608028654 * 1.4426950408889634)
    k2 = f32(-2 * 0.7978845608028654 * 1.4426950408889634 * 0.044715)
    with np.errstate(over="ignore"):
        e = np.exp2((v * ((v * v).astype(f32) * k2 + k1).astype(f32)).astype(f32)).astype(f32)
    g = (v * (f32(1) / (f32(1) + e).astype(f32)).astype(f32)).astype(f32)
    ref = 0.5 * t * (1 + np.tanh(0.7978845608028654 * (t + 0.044715 * t ** 3)))
    u = (f32(0.7978845608028654) * (v + f32(0.044715) * v * v * v)).astype(f32)
    g0 = (f32(0.5) * v * (f32(1) + np.tanh(u.astype(np.float64)).astype(f32))).astype(f32)
    big = np.abs(ref) > 1e-3
    for name, x in (("gelu_tanh (sigmoid form)", g), ("0.5 v (1 + tanh u), correctly rounded tanh", g0)):
        err = np.abs(x - ref)
        print("%s: max abs %.3e, max rel where |value| > 1e-3 %.3e" % (name, err.max(), (err[big] / np.abs(ref[big])).max()))
    print("K1 = %r, K2 = %r" % (float(k1), float(k2)))


if __name__ == "__main__":
    main()
