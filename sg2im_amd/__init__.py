"""sg2im_amd: the sg2im training hot path on MI355X (gfx950).

Hand-written HIP kernels (sg2im_amd/csrc, C ABI in include/sg2im_hip.h) behind the
reference's Python class API: ``Sg2ImModel``, ``PatchDiscriminator``,
``AcCropDiscriminator`` keep the reference's constructor arguments, forward signatures
and state_dict keys.  There is no CPU fallback: the modules need a GPU and the built
``libsg2im_hip.so`` (``python -m sg2im_amd.build``)."""

__version__ = '0.2.0'
