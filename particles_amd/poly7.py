"""Seventh-degree force-curve polynomial (reference: src/poly7.rs).

``coeffs[i]`` multiplies ``x**i`` (src/poly7.rs:60-68); the shader's field
names h,g,f,e,d,c,b,a are coeffs[0..8] (src/compute.wgsl:40-49).
"""
import numpy as np

POLY7_DTYPE = np.dtype([("coeffs", np.float32, (8,))])
assert POLY7_DTYPE.itemsize == 32


class Poly7:
    """Host-side mirror of ``Poly7`` (src/poly7.rs:6-75)."""

    COEFF_NAMES = ("h", "g", "f", "e", "d", "c", "b", "a")  # src/poly7.rs:17-19

    def __init__(self, coeffs=None):
        self.coeffs = np.zeros(8, dtype=np.float32) if coeffs is None else np.asarray(coeffs, dtype=np.float32).copy()
        if self.coeffs.shape != (8,):
            raise ValueError("Poly7 takes exactly 8 coefficients")

    @classmethod
    def zero(cls):  # src/poly7.rs:27-29
        return cls()

    @classmethod
    def const_val(cls, val):  # src/poly7.rs:31-35
        c = np.zeros(8, dtype=np.float32)
        c[0] = val
        return cls(c)

    def invert(self):  # src/poly7.rs:21-25
        self.coeffs = -self.coeffs

    def eval(self, x):
        """Host evaluation in the order of src/poly7.rs:60-68 (ascending
        powers, running product) — the plotting twin, not the shader's order."""
        res = np.float32(0.0)
        x_to_the_i = np.float32(1.0)
        x = np.float32(x)
        for c in self.coeffs:
            res = np.float32(res + x_to_the_i * c)
            x_to_the_i = np.float32(x_to_the_i * x)
        return res

    @classmethod
    def from_points(cls, points):
        """Interpolating polynomial through 8 (x, y) points: src/poly7.rs:37-58.

        Builds the 8x8 Vandermonde matrix in fp32, inverts it with the
        reference's pivot-free Gauss-Jordan elimination (src/poly7.rs:120-147)
        and multiplies by the y values.  Returns ``None`` when the inverse has
        a non-finite entry (repeated x), as the reference does.
        """
        pts = np.asarray(points, dtype=np.float32)
        if pts.shape != (8, 2):
            raise ValueError("from_points takes 8 (x, y) pairs")
        f32 = np.float32
        n = 8
        m = np.empty((n, n), dtype=np.float32)
        for i in range(n):
            for j in range(n):
                m[i, j] = f32(pts[i, 0]) ** j if j else f32(1.0)  # powi
        res = np.eye(n, dtype=np.float32)
        with np.errstate(all="ignore"):
            for i in range(n):
                for j in range(i + 1, n):
                    pivot = f32(-m[j, i] / m[i, i])
                    m[j, :] = m[j, :] + pivot * m[i, :]
                    res[j, :] = res[j, :] + pivot * res[i, :]
                div = f32(m[i, i])
                m[i, :] = m[i, :] / div
                res[i, :] = res[i, :] / div
            for i in range(n - 1, -1, -1):
                for j in range(i - 1, -1, -1):
                    pivot = f32(-m[j, i] / m[i, i])
                    m[j, :] = m[j, :] + pivot * m[i, :]
                    res[j, :] = res[j, :] + pivot * res[i, :]
            if not np.all(np.isfinite(res)):
                return None
            coeffs = np.zeros(n, dtype=np.float32)
            for r in range(n):
                acc = f32(0.0)
                for i in range(n):
                    acc = f32(acc + res[r, i] * pts[i, 1])
                coeffs[r] = acc
        return cls(coeffs)
