from .KM3D import KM3D  # noqa: F401
from .yolomono3d_detector import GroundAwareYolo3D, Yolo3D  # noqa: F401
from .yolostereo3d_detector import Stereo3D  # noqa: F401
