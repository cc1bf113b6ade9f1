"""Coefficients of the degree-7 polynomial the candidate walk uses for exp on [-0.25, 0] (cvo_kernels.hip, exp7_*):
Chebyshev interpolation of exp(x) on x in [-1/4, 0] in 80-bit arithmetic, written in powers of x.
Prints the coefficients as C hex-float literals and the measured maximum relative error on a dense grid."""
import numpy as np

LD = np.longdouble
deg = 7
h = LD(1) / LD(8)
k = np.arange(deg + 1, dtype=LD)
nodes = h * np.cos((2 * k + 1) * LD(np.pi) / (2 * (deg + 1))) - h       # Chebyshev nodes on [-2h, 0]
# (np.pi is a double: the nodes are merely near the Chebyshev points, which is all interpolation needs)
V = np.vander(nodes, deg + 1, increasing=True).astype(LD)
f = np.exp(nodes)
# solve in long double by Gaussian elimination with partial pivoting
A = V.copy(); b = f.copy(); n = deg + 1
for i in range(n):
    p = i + int(np.argmax(np.abs(A[i:, i])))
    A[[i, p]] = A[[p, i]]; b[[i, p]] = b[[p, i]]
    for r in range(i + 1, n):
        m = A[r, i] / A[i, i]
        A[r, i:] -= m * A[i, i:]; b[r] -= m * b[i]
c = np.zeros(n, LD)
for i in range(n - 1, -1, -1):
    c[i] = (b[i] - (A[i, i + 1:] * c[i + 1:]).sum()) / A[i, i]
cd = c.astype(np.float64)
u = np.linspace(-2 * float(h), 0.0, 2_000_001).astype(LD)
p = np.zeros_like(u)
for ck in cd[::-1]:
    p = p * u + LD(ck)
err = np.abs(p / np.exp(u) - 1)
print("max relative error of the double-rounded coefficients (exact Horner):", float(err.max()))
# Horner in double with fma-free steps bounds the evaluation error from above (the device uses fma: one rounding per step)
ud = u.astype(np.float64); pd = np.zeros_like(ud)
for ck in cd[::-1]:
    pd = pd * ud + ck
print("max relative error evaluated in double:", float(np.abs(pd.astype(LD) / np.exp(u) - 1).max()))
for i, ck in enumerate(cd):
    print(f"c{i} = {float(ck).hex()}  /* {float(ck)!r} */")
