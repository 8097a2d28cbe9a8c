"""Coefficients of the central part of dqnorm / orc_qnorm: Phi^-1(p) = y F(w), y = 2 p - 1, w = -log(1 - y^2) in [0, 6.25].
F(w) = sqrt(2) erfinv(y) / y is smooth there; this fits a degree-24 polynomial in s = w - 3.125 by Chebyshev interpolation at
60 digits and prints it with the error of the whole double-precision evaluation against mpmath."""
import mpmath as mp
import numpy as np

mp.mp.dps = 60
W, DEG = mp.mpf("6.25"), 24


def F(w):
    w = mp.mpf(w)
    if w == 0:
        return mp.sqrt(mp.pi / 2)
    y = mp.sqrt(1 - mp.e ** (-w))
    return mp.sqrt(2) * mp.erfinv(y) / y


n = DEG + 1
nodes = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
fx = [F(W / 2 + W / 2 * t) for t in nodes]
c = [2 / mp.mpf(n) * sum(fx[k] * mp.cos(mp.pi * j * (2 * k + 1) / (2 * n)) for k in range(n)) for j in range(n)]
c[0] /= 2
T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]          # Chebyshev polynomials as monomial coefficients
for j in range(2, n):
    new = [mp.mpf(0)] + [2 * x for x in T[-1]]
    for i, x in enumerate(T[-2]):
        new[i] -= x
    T.append(new)
mono = [mp.mpf(0)] * n
for j in range(n):
    for i, x in enumerate(T[j]):
        mono[i] += c[j] * x
coef = [float(mono[i] / (W / 2) ** i) for i in range(n)]      # ascending powers of s = w - 3.125
if __name__ == "__main__":
    for x in coef:
        print(repr(x))

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    p = np.concatenate([rng.random(200000), 0.5 + np.linspace(-0.4995, 0.4995, 20001), 0.5 + 10.0 ** rng.uniform(-17, -1, 20000)])
    y = 2 * (p - 0.5)
    w = -np.log((1 - y) * (1 + y))
    m = w < 6.25
    s = w - 3.125
    acc = np.full_like(s, coef[-1])
    for x in coef[-2::-1]:
        acc = acc * s + x
    res = y * acc
    err = [float(abs(mp.mpf(float(res[i])) / (mp.sqrt(2) * mp.erfinv(mp.mpf(float(y[i])))) - 1)) for i in np.where(m & (y != 0))[0][::7]]
    print(f"central fraction {m.mean():.5f}; max relative error {max(err):.3g} over {len(err)} points")
