from .utils.registry import *  # noqa: F401,F403
from . import backbones  # noqa: F401
from . import detectors  # noqa: F401
