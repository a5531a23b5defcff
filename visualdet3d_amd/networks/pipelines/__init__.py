from .testers import test_mono_detection, test_stereo_detection  # noqa: F401
from .evaluators import postprocess_batch, test_one  # noqa: F401
