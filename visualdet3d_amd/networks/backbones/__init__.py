from .dla import dlanet  # noqa: F401
from .resnet import resnet  # noqa: F401
from ..utils.registry import BACKBONE_DICT


def build_backbone(cfg):
    """backbones/__init__.py:5-14 of the reference: ``name`` selects the registered factory (default 'resnet')."""
    temp_cfg = dict(cfg)
    name = temp_cfg.pop('name') if 'name' in temp_cfg else 'resnet'
    return BACKBONE_DICT[name](**temp_cfg)
