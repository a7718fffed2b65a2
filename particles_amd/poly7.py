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
        """Interpolating polynomial through 8 (x, y) points: src/poly7.rs:37-58
        (``pcl_poly7_from_points`` of the C ABI: the 8x8 Vandermonde matrix in
        fp32 with ``powi`` powers, the reference's Gauss-Jordan elimination
        without pivot search, src/poly7.rs:120-147, then the product with the y
        values).  Returns ``None`` where the reference does (a non-finite entry
        in the inverse, e.g. a repeated x)."""
        import ctypes as C
        from . import _lib
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if pts.shape != (8, 2):
            raise ValueError("from_points takes 8 (x, y) pairs")
        out = np.zeros(8, dtype=np.float32)
        rc = _lib.load().pcl_poly7_from_points(pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        if rc == _lib.PCL_ERR_SINGULAR:
            return None
        _lib.check(rc)
        return cls(out)
