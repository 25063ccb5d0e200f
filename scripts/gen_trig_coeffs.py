"""Coefficients of the library's own atan2 / asin / acos ("software trigonometry").

The reference calls f64::atan2 / asin / acos (ray_casting.rs:137-138, sphere.rs:42-43), i.e. the platform libm, whose
last-ulp behaviour differs between glibc and the device's ocml.  To make the image-texture and spherical-sky scenes
bit-exact between the oracle and the device, both evaluate the SAME algorithm with +, -, *, /, sqrt only:

  atan:  fdlibm's four-breakpoint argument reduction (7/16, 11/16, 19/16, 39/16), then
         atan(t) = t - t*z*A(z), z = t*t, A = polynomial below (approximates 1/3 - z/5 + z^2/7 - ... on [0, (7/16)^2])
  asin:  |x| < 0.5: x + x*z*S(z), z = x*x, S below (approximates (asin(x) - x)/x^3 on [0, 1/4]);
         otherwise through sqrt((1 - |x|)/2)

This script derives A and S (Chebyshev fits computed in 60-digit arithmetic, near-minimax) and prints them as C
initialisers for crucible_amd/csrc/softtrig.hpp and oracle/crucible_oracle.c, with the worst-case approximation error.
usage: python scripts/gen_trig_coeffs.py
"""
import mpmath as mp

mp.mp.dps = 60


def atan_f(z):
    z = mp.mpf(z)
    if z == 0:
        return mp.mpf(1) / 3
    t = mp.sqrt(z)
    return (t - mp.atan(t)) / (t * z)


def asin_f(z):
    z = mp.mpf(z)
    if z == 0:
        return mp.mpf(1) / 6
    x = mp.sqrt(z)
    return (mp.asin(x) - x) / (x * z)


def fit(f, hi, degree, name, fmt):
    poly = mp.chebyfit(f, [0, hi], degree + 1)          # highest power first
    coeffs = poly[::-1]
    worst = 0
    for k in range(2001):
        z = mp.mpf(hi) * k / 2000
        approx = sum(fmt(c) * z ** i for i, c in enumerate(coeffs))
        worst = max(worst, abs(approx - f(z)))
    print(f"/* {name}: degree {degree} on [0, {hi}], max |error| of the rounded polynomial {mp.nstr(worst, 3)} */")
    return coeffs


def as_f64(c):
    return mp.mpf(float(c))


def as_f32(c):
    import numpy as np
    return mp.mpf(float(np.float32(float(c))))


if __name__ == "__main__":
    import numpy as np
    a64 = fit(atan_f, mp.mpf(7) / 16 * mp.mpf(7) / 16, 11, "ATAN f64", as_f64)
    print("{" + ", ".join(float(c).hex() for c in a64) + "}")
    print("{" + ", ".join(repr(float(c)) for c in a64) + "}")
    a32 = fit(atan_f, mp.mpf(7) / 16 * mp.mpf(7) / 16, 5, "ATAN f32", as_f32)
    print("{" + ", ".join(float(np.float32(float(c))).hex() for c in a32) + "}")
    print("{" + ", ".join(repr(float(np.float32(float(c)))) + "f" for c in a32) + "}")
    s64 = fit(asin_f, mp.mpf(1) / 4, 15, "ASIN f64", as_f64)
    print("{" + ", ".join(float(c).hex() for c in s64) + "}")
    print("{" + ", ".join(repr(float(c)) for c in s64) + "}")
    s32 = fit(asin_f, mp.mpf(1) / 4, 6, "ASIN f32", as_f32)
    print("{" + ", ".join(float(np.float32(float(c))).hex() for c in s32) + "}")
    print("{" + ", ".join(repr(float(np.float32(float(c)))) + "f" for c in s32) + "}")
    for name, v in (("atan(0.5)", mp.atan(mp.mpf(1) / 2)), ("pi/4", mp.pi / 4), ("atan(1.5)", mp.atan(mp.mpf(3) / 2)), ("pi/2", mp.pi / 2),
                    ("pi", mp.pi)):
        hi = float(v)
        lo = float(v - mp.mpf(hi))
        hi32 = float(np.float32(float(v)))
        lo32 = float(np.float32(float(v - mp.mpf(hi32))))
        print(f"/* {name} */ f64 hi {hi.hex()} lo {lo.hex()} ({hi!r}, {lo!r});  f32 hi {hi32!r}f lo {lo32!r}f")
