"""slim_amd -- MI355X-native SLIM training engine.

``from slim_amd import SLIM, SLIMatrix`` is the drop-in for the reference's
``from SLIM import SLIM, SLIMatrix`` (python-package/SLIM/__init__.py).  The solver
is HIP on gfx950 behind the C ABI of include/slim.h; see DESIGN.md.
"""
name = "slim_amd"

from . import constants as config  # noqa: E402  (reference exposes SLIM.config)
from .interface import SLIM, SLIMatrix  # noqa: E402

__all__ = ["SLIM", "SLIMatrix", "config"]
