"""dynamicfusion_body_amd -- MI355X-native per-frame DynamicFusion hot path.

TSDF integration (depth -> volume, volume -> volume rigid / DQB-warped) and the warp-field
solve of nintendops/DynamicFusion_Body as hand-written HIP kernels for gfx950 behind the
reference's own Python call surface.  See DESIGN.md.
"""
from . import scene  # noqa: F401
from .fusion_dm import FusionDM  # noqa: F401
from .fusion import Fusion  # noqa: F401

# the reference selects its device plug-in by class name (test.py:158-161)
FusionDM_GPU = FusionDM

__all__ = ["Fusion", "FusionDM", "FusionDM_GPU", "scene"]
