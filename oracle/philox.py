"""Philox4x32-10 stream of torch's GPU default generator (oracle; test infrastructure only).

The reference draws its dropout uniforms with `torch.rand_like(epis_uncert)` (reference
models/llava.py:650, models/llavanext.py:797); with the model on a GPU that is the device
generator: Philox4x32-10 keyed by the `torch.manual_seed(seed)` value (models/llava.py:16-20),
with a 64-bit offset that every random kernel advances.  For one `rand_like` over n <= 524288
float32 elements (ATen's elementwise random kernel: 256 threads per block, 4 draws per thread per
trip, one trip) element i is

    x = philox4x32_10(counter = (offset/4 as 64 bit, subsequence i as 64 bit), key = seed)[0]
    u = 2**-32 + float32(x) * 2**-32          (one fused multiply-add; rocRAND rocrand_uniform.h:67)
    u = 0 if u == 1 else u                    (ATen maps the closed end back to `from`)

and the offset then moves by 4.  The algorithm is the published Philox4x32-10 (Salmon et al.,
SC'11; multipliers 0xD2511F53 / 0xCD9E8D57, Weyl constants 0x9E3779B9 / 0xBB67AE85); the counter
layout is the one cuRAND and rocRAND share (rocrand_philox4x32_10.h:199-233).  Pinned against
`torch.manual_seed(s); torch.rand(n, device="cuda")` on the MI355X by
tests/test_gpu_dropout_ops.py::test_philox_stream_is_torch_gpu_rand and by the vectors in
tests/golden/g8_philox.npz (made there by oracle/gen_golden_philox.py).
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Ten rounds over arrays of 32-bit counter words (held as uint64); returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for r in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        ka, kb = np.uint64((k0 + r * _W0) & 0xFFFFFFFF), np.uint64((k1 + r * _W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = (p1 >> _S32) ^ c1 ^ ka, p1 & _MASK, (p0 >> _S32) ^ c3 ^ kb, p0 & _MASK
    return c0, c1, c2, c3


def uniform_f32(x: np.ndarray) -> np.ndarray:
    """float32(x) * 2**-32 + 2**-32 as ONE fused multiply-add, then 1.0 -> 0.0."""
    xf = x.astype(np.uint32).astype(np.float32).astype(np.float64)     # the u32 -> f32 conversion rounds to nearest even
    c = float(np.float32(2.3283064e-10))
    u = (xf * c + c).astype(np.float32)                                # exact in f64 (24 x 24 bit product + one term), one rounding
    u[u == np.float32(1.0)] = np.float32(0.0)
    return u


class TorchGpuPhilox:
    """Restates `torch.manual_seed(seed)` + successive `torch.rand(n, device="cuda")` (n <= 524288)."""

    MAX_N = 524288   # 256 CUs x 8 blocks x 256 threads: beyond it ATen's kernel grid-strides and the mapping changes

    def __init__(self, seed: int, offset: int = 0):
        if offset % 4:
            raise ValueError("offset must be a multiple of 4")
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset)

    def raw(self, n: int) -> np.ndarray:
        if n > self.MAX_N:
            raise ValueError("n too large for the one-trip mapping")
        i = np.arange(n, dtype=np.uint64)
        o = self.offset // 4
        x, _, _, _ = philox4x32_10(np.full(n, o & 0xFFFFFFFF, dtype=np.uint64), np.full(n, o >> 32, dtype=np.uint64),
                                   i & _MASK, i >> _S32, self.seed & 0xFFFFFFFF, self.seed >> 32)
        self.offset += 4
        return x.astype(np.uint32)

    def rand_f32(self, n: int) -> np.ndarray:
        return uniform_f32(self.raw(n))
