from .config import EasyDict, cfg_from_file  # noqa: F401
