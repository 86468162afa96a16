"""heal_swin_amd -- MI355X-native HEAL-SWIN forward/backward hot path.

Host side (Python, mirroring the reference's `heal_swin.models_torch` interface) over the C ABI of
`libhealswin.so` (HIP kernels for gfx950, `include/healswin.h`).  PyTorch provides device memory,
streams, autograd bookkeeping and `torch.distributed`; the hot-path arithmetic runs in the library.

    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerSys, SwinHPTransformerConfig
    from heal_swin_amd.data_spec import DataSpec
"""
from . import _lib  # noqa: F401  (loads libhealswin.so; raises if it has not been built)

__version__ = "0.1.0"
