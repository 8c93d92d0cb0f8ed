"""pgtformer_amd — MI355X-native (gfx950) forward path of PGTFormer.

Public surface mirrors the reference repo's: `PGTFormer`, `TDCRQVAE3` (same constructor kwargs,
state-dict keys and forward signatures), an `ARCH_REGISTRY`, and a driver (`pgtformer_amd.driver`)
with the semantics of the reference's inference.py.  All compute runs in hand-written HIP kernels
(`pgtformer_amd/csrc`, C-ABI in include/pgt_hip.h); importing this package does not need a GPU, running
a model does (there is no CPU fallback).
"""
from .config import default_config, load_config  # noqa: F401
from .registry import ARCH_REGISTRY  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name in ("PGTFormer", "TDCRQVAE3"):
        from .archs.pgtformer_arch import PGTFormer
        from .archs.tdcrqvae3_arch import TDCRQVAE3
        return {"PGTFormer": PGTFormer, "TDCRQVAE3": TDCRQVAE3}[name]
    raise AttributeError(name)
