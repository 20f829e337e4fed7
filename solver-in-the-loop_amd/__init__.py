"""solver-in-the-loop_amd -- MI355X (gfx950) native engine for the solver-in-the-loop hot path.

Import name: `sol_amd` (see /sol_amd.py; the directory name carries a hyphen).
Host-side mirror of the reference's call surface for this path + ctypes binding of the
C-ABI library libsol_hip.so (include/sol_hip.h).  No CPU fallback anywhere.
"""
from . import _build, _lib  # noqa: F401
from ._lib import SolError, load, declared_symbols, lib_path  # noqa: F401
from .fluid import (box, Box, Sphere, Inflow, Obstacle, Gravity, Domain, OPEN, PERIODIC,  # noqa: F401
                    CenteredGrid, StaggeredGrid, Fluid, BurgersVelocitySMAC, unstack_staggered_tensor)
from . import ops, dist  # noqa: F401
from .karman import KarmanFlow, to_feature, to_staggered, lr_schedule, velocity_bc_masks  # noqa: F401
from .model import model_mars_moon, model_mercury, MarsMoon, Mercury, ConvNet  # noqa: F401
from .trainer import SolTrainer, SolRollout, GraphTrainer, make_trainer  # noqa: F401
from . import synthetic, scene, burgers  # noqa: F401
from .burgers import BurgersTest, BurgersTrainer, BurgersRollout, TFAdam  # noqa: F401
from . import karman3d, precond3d  # noqa: F401

__version__ = "0.1.0"


def build(force=False, verbose=False):
    """Compile libsol_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    return _build.build(force=force, verbose=verbose)
