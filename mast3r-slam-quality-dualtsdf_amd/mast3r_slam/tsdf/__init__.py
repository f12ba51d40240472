from .global_volume import TSDFVolume  # noqa: F401
from .tsdf_optimizer import TSDFPoseOptimizer  # noqa: F401
from .global_manager import TSDFGlobalIntegrator, TSDFGlobalManager  # noqa: F401
