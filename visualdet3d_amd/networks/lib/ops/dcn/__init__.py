from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, deform_conv,  # noqa: F401
                          modulated_deform_conv)
