"""dev: fit Phi(x) = 0.5 erfc(-x/sqrt2) for the fp32 GELU / GELU' (csrc/common.h gg_phi_f32): h(t) = 0.5 erfc(t) = exp2(P(t)), t = |x|/sqrt2 clamped
to 4.2, P a polynomial (weighted fit: the error of h is h * ln2 * the error of P); Phi = x < 0 ? h : 1 - h.  Reports the error of Phi, of
GELU = x Phi and of GELU' = Phi + x phi against double precision with every step rounded to fp32."""
import numpy as np
from scipy.special import erfc
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
TMAX = 4.2
DEG = 9
t = 0.5 * TMAX * (np.cos(np.pi * (np.arange(20000) + 0.5) / 20000) + 1)
y = np.log2(0.5 * erfc(t))
w = 0.5 * erfc(t) + 1e-9                       # error of exp2(P) = h ln2 dP
coef = np.polynomial.polynomial.polyfit(t, y, DEG, w=w)
print("coef (low -> high):", ", ".join("%.9g" % f32(c) for c in coef))
def phi32(x):
    x = x.astype(f32)
    tt = np.minimum(np.abs(x) * f32(0.70710678118654752), f32(TMAX)).astype(f32)
    p = np.full_like(x, f32(coef[-1]))
    for c in coef[-2::-1]: p = fma(p, tt, np.full_like(x, f32(c)))
    h = np.exp2(p.astype(np.float64)).astype(f32)
    return np.where(x < 0, h, (f32(1.0) - h).astype(f32))
x = np.linspace(-8, 8, 4000001)
xs = x.astype(f32).astype(np.float64)
ref = 0.5 * erfc(-xs / np.sqrt(2.0))
got = phi32(x).astype(np.float64)
e = np.abs(got - ref)
print("Phi  max abs err %.3g at x=%.4f" % (e.max(), x[e.argmax()]))
eg = np.abs(xs * got - xs * ref)
print("GELU max abs err %.3g at x=%.4f ; max rel err for |gelu|>1e-3: %.3g" % (eg.max(), x[eg.argmax()], (eg / np.maximum(np.abs(xs * ref), 1e-30))[np.abs(xs * ref) > 1e-3].max()))
