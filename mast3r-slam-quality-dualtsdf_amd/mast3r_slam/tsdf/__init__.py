from .global_volume import TSDFVolume  # noqa: F401
from .tsdf_optimizer import TSDFPoseOptimizer  # noqa: F401
