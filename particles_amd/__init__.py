"""MI355X-native particle update behind the reference's ``Compute`` surface.

The compute backend is the in-tree HIP library ``libparticles_hip.so``
(gfx950); see include/particles_hip.h for the C ABI.
"""
from .particle import PARTICLE_DTYPE, MASS_WRAP_DTYPE, NUM_TYPES, empty_particles, lattice_particles, random_particles
from .poly7 import POLY7_DTYPE, Poly7
from .grid import Bounds, Grid
from .sim_params import SIM_PARAMS_DTYPE, SimParams
from ._lib import PclError, KERNEL_NONE, KERNEL_STREAM, KERNEL_PAIRS, PAIRS_AUTO, PAIRS_ALL, PAIRS_BINNED
from ._lib import (STEP_NONE, STEP_STREAM, STEP_PAIRS, STEP_STREAM_COUNT, STEP_STREAM_SCATTER, STEP_PAIRS_BINNED,
                   STEP_PAIRS_GATHERED, STEP_STREAM_FUSED)
from .compute import Compute
from .cursor import BRUSH_INPUT_DTYPE, Brush, DeviceBrush, EditMode
from .sharded import ShardedCompute, shard_range
from .group import Group

__all__ = [
    "PARTICLE_DTYPE", "MASS_WRAP_DTYPE", "NUM_TYPES", "empty_particles", "lattice_particles", "random_particles",
    "POLY7_DTYPE", "Poly7", "Bounds", "Grid", "SIM_PARAMS_DTYPE", "SimParams",
    "STEP_NONE", "STEP_STREAM", "STEP_PAIRS", "STEP_STREAM_COUNT", "STEP_STREAM_SCATTER", "STEP_PAIRS_BINNED",
    "STEP_PAIRS_GATHERED", "STEP_STREAM_FUSED",
    "PAIRS_AUTO", "PAIRS_ALL", "PAIRS_BINNED",
    "PclError", "KERNEL_NONE", "KERNEL_STREAM", "KERNEL_PAIRS", "Compute", "Group", "ShardedCompute", "shard_range", "Brush", "DeviceBrush", "EditMode", "BRUSH_INPUT_DTYPE",
]
