"""dropoutdecoding_amd — MI355X-native Dropout Decoding hot path (HIP kernels behind a C-ABI).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic of the
path runs in libdropdec.so (include/dropdec.h).  See DESIGN.md.
"""
from .config import settings  # noqa: F401

__all__ = ["settings"]
