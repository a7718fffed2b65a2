"""The 912-byte parameter block (reference: src/sim_params.rs:6-61).

Offsets: attraction_force @0 (25 x Poly7), particle_type_masses @800
(5 x MassWrap), force_grid_dimensions @880, delta_t @892, max_velocity @896,
bounding_volume_radius @900, cut_off_distance @904, distance_exponent @908
(mirrored by src/compute.wgsl:17-28).
"""
import numpy as np

from .grid import Bounds, Grid
from .particle import MASS_WRAP_DTYPE
from .poly7 import POLY7_DTYPE

SIM_PARAMS_DTYPE = np.dtype(
    [
        ("attraction_force", POLY7_DTYPE, (25,)),
        ("particle_type_masses", MASS_WRAP_DTYPE, (5,)),
        ("force_grid_dimensions", np.uint32, (3,)),
        ("delta_t", np.float32),
        ("max_velocity", np.float32),
        ("bounding_volume_radius", np.float32),
        ("cut_off_distance", np.float32),
        ("distance_exponent", np.float32),
    ]
)
assert SIM_PARAMS_DTYPE.itemsize == 912
assert SIM_PARAMS_DTYPE.fields["particle_type_masses"][1] == 800
assert SIM_PARAMS_DTYPE.fields["force_grid_dimensions"][1] == 880
assert SIM_PARAMS_DTYPE.fields["delta_t"][1] == 892
assert SIM_PARAMS_DTYPE.fields["distance_exponent"][1] == 908


class SimParams:
    """Mirror of ``SimParams`` with the defaults of src/sim_params.rs:20-31."""

    def __init__(self):
        self.raw = np.zeros((), dtype=SIM_PARAMS_DTYPE)
        self.raw["particle_type_masses"]["mass"] = 1.0
        self.raw["force_grid_dimensions"] = 10
        self.raw["delta_t"] = 0.0
        self.raw["max_velocity"] = 100.0
        self.raw["bounding_volume_radius"] = 10.0
        self.raw["cut_off_distance"] = 1.0
        self.raw["distance_exponent"] = 0.0

    # scalar fields -----------------------------------------------------
    def _get(self, name):
        return self.raw[name].item() if self.raw[name].shape == () else self.raw[name]

    delta_t = property(lambda s: s._get("delta_t"), lambda s, v: s.raw.__setitem__("delta_t", v))
    max_velocity = property(lambda s: s._get("max_velocity"), lambda s, v: s.raw.__setitem__("max_velocity", v))
    bounding_volume_radius = property(
        lambda s: s._get("bounding_volume_radius"), lambda s, v: s.raw.__setitem__("bounding_volume_radius", v))
    cut_off_distance = property(
        lambda s: s._get("cut_off_distance"), lambda s, v: s.raw.__setitem__("cut_off_distance", v))
    distance_exponent = property(
        lambda s: s._get("distance_exponent"), lambda s, v: s.raw.__setitem__("distance_exponent", v))

    @property
    def force_grid_dimensions(self):
        return tuple(int(v) for v in self.raw["force_grid_dimensions"])

    @force_grid_dimensions.setter
    def force_grid_dimensions(self, dims):
        self.raw["force_grid_dimensions"] = np.asarray(dims, dtype=np.uint32)

    @property
    def masses(self):
        return self.raw["particle_type_masses"]["mass"].copy()

    @masses.setter
    def masses(self, values):
        self.raw["particle_type_masses"]["mass"] = np.asarray(values, dtype=np.float32)

    def set_poly(self, ty_self, ty_other, coeffs):
        """Force curve felt by a particle of type ``ty_self`` from one of type
        ``ty_other``: table index ty_other + 5*ty_self (src/compute.wgsl:134)."""
        self.raw["attraction_force"]["coeffs"][ty_other + 5 * ty_self] = np.asarray(
            getattr(coeffs, "coeffs", coeffs), dtype=np.float32)

    @property
    def polys(self):
        return self.raw["attraction_force"]["coeffs"]

    def pair_term_inert(self):
        """True when the all-pairs loop provably adds exactly +0: no distance
        can pass both cut-offs, or every polynomial coefficient is +-0 while
        the masses are finite and pow(len, exponent) cannot overflow for len
        in [0.001, cut_off] (src/compute.wgsl:126-131,139; 0 * inf would be
        NaN).  The same rule, on double-precision logarithms, as the library."""
        import math
        cut = float(np.float32(self.raw["cut_off_distance"]))
        if cut < float(np.float32(0.001)):
            return True
        if np.any(self.raw["attraction_force"]["coeffs"] != 0.0):
            return False
        if not np.all(np.isfinite(self.raw["particle_type_masses"]["mass"])):
            return False
        e = float(np.float32(self.raw["distance_exponent"]))
        if e == 0.0:
            return True
        if not math.isfinite(e):
            return False
        hi = cut if math.isfinite(cut) else 3.4028234663852886e38
        reach = max(abs(math.log(0.001)), abs(math.log(hi)))
        return abs(e) * reach <= 69.0

    def copy(self):
        c = SimParams()
        c.raw = self.raw.copy()
        return c

    def tobytes(self):
        return self.raw.tobytes()

    # field constructors -----------------------------------------------
    def _bounds(self):
        bvr = np.float32(self.raw["bounding_volume_radius"])
        v = np.array([bvr, bvr, bvr], dtype=np.float32)
        return Bounds(-v, np.float32(2.0) * v)

    def new_force_grid_centered(self):  # src/sim_params.rs:33-45
        d = self.force_grid_dimensions
        return Grid.new_centered(d[0], d[1], d[2], self._bounds())

    def new_force_grid_zero(self):  # src/sim_params.rs:47-60
        d = self.force_grid_dimensions
        return Grid.new_uniform(d[0], d[1], d[2], self._bounds())
