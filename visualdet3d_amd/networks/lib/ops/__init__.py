from .dcn.deform_conv import DeformConvPack, ModulatedDeformConvPack  # noqa: F401
