"""MI355X-native registration hot path of PointcloudProcessing.

The classes keep the reference's names and call surface; the compute backend
is libpcp.so (hand-written gfx950 HIP kernels behind the C ABI of
include/pcp.h).
"""
from ._lib import LIB_PATH, PcpError, load  # noqa: F401
from .alignment import PointcloudAlignment  # noqa: F401
from .compute_shader import MODE_PLANE, MODE_POINT, ComputeShader, IcpBatch  # noqa: F401
from .icp_container import PointcloudIcpContainer  # noqa: F401
from .cell_grid import CellGrid  # noqa: F401
from .environment import EnvironmentConstructor  # noqa: F401
from .frame import DeviceFrame  # noqa: F401
from .oxts import OXTS, OxtsDataReader  # noqa: F401
from .voxelizer import Voxelizer  # noqa: F401
from .sequence import PinnedFrames, SequencePipeline  # noqa: F401
from .map_sink import MapSink  # noqa: F401
from .chain import OdometryChain, chain_odometry, chain_odometry_files  # noqa: F401
from .frame_files import FrameFileRuns, velodyne_files  # noqa: F401
