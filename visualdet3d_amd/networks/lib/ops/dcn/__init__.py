from . import deform_conv_ext  # noqa: F401  (the reference's pybind module, name for name)
from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, deform_conv,  # noqa: F401
                          modulated_deform_conv)
