"""dev: fit the two-branch fp32 erf used by the fp32 mode's GELU / GELU' (csrc/common.h gg_erff) and report its error against
scipy.special.erf with every operation rounded to fp32 (fma = one rounding)."""
import numpy as np
from scipy.special import erf, erfc
from numpy.polynomial import chebyshev as Ch, polynomial as Po

f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

C = 0.92
def cheb_fit(fn, lo, hi, deg, n=4000):
    k = np.arange(n); u = np.cos(np.pi * (k + 0.5) / n); x = 0.5 * (hi - lo) * u + 0.5 * (hi + lo)
    c = Ch.chebfit(u, fn(x), deg)
    p = Ch.cheb2poly(c)                       # polynomial in u
    # substitute u = (2x - (hi+lo)) / (hi-lo)
    a, b = 2.0 / (hi - lo), -(hi + lo) / (hi - lo)
    out = np.zeros(1)
    for i, ci in enumerate(p):
        out = Po.polyadd(out, ci * Po.polypow([b, a], i))
    return out
# small: erf(x) = x + x * P(s), s = x^2, P(s) ~ erf(x)/x - 1
ps = cheb_fit(lambda s: erf(np.sqrt(s)) / np.sqrt(s) - 1.0, 1e-12, C * C + 0.02, 6)
# large: erf(t) = 1 - exp(-t * R(t)), R(t) = -log(erfc(t)) / t
pl = cheb_fit(lambda t: -np.log(erfc(t)) / t, C - 0.02, 4.05, 7)
print("small", [float(f32(c)) for c in ps]); print("large", [float(f32(c)) for c in pl])

def erf32(x):
    x = x.astype(f32); t = np.abs(x); s = (x.astype(np.float64) ** 2).astype(f32)
    r = np.full_like(x, f32(ps[-1]))
    for c in ps[-2::-1]: r = fma(r, s, np.full_like(x, f32(c)))
    small = fma(r, x, x)
    tt = np.minimum(t, f32(4.0))
    q = np.full_like(x, f32(pl[-1]))
    for c in pl[-2::-1]: q = fma(q, tt, np.full_like(x, f32(c)))
    e = (-(q.astype(np.float64) * tt.astype(np.float64))).astype(f32)              # -t*R(t)
    ex = np.exp2((e.astype(np.float64) * 1.4426950408889634).astype(f32).astype(np.float64)).astype(f32)   # v_exp_f32(e * log2e)
    large = np.copysign((f32(1.0) - ex).astype(f32), x)
    return np.where(t > f32(C), large, small)
x = np.concatenate([np.linspace(-6, 6, 2000001), np.linspace(-1e-3, 1e-3, 20001)])
err = np.abs(erf32(x).astype(np.float64) - erf(x.astype(f32).astype(np.float64)))
print("max abs err", err.max(), "at", x[err.argmax()], " ulp of 1:", 2.0 ** -24)
rel = err / np.maximum(np.abs(erf(x.astype(f32).astype(np.float64))), 1e-30)
print("max rel err", rel.max(), "at", x[rel.argmax()])
