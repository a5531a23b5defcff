from .resnet import resnet  # noqa: F401
