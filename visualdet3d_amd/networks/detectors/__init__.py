from .yolostereo3d_detector import Stereo3D  # noqa: F401
