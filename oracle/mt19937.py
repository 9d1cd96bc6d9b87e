"""mt19937 stream of torch's CPU default generator (oracle; test infrastructure only).

The reference draws its dropout uniforms with `torch.rand_like(epis_uncert)`
(reference models/llava.py:650, models/llavanext.py:797) from torch's *global default
generator*, seeded once at import time by `torch.manual_seed(seed)`
(models/llava.py:16-20, models/llavanext.py:18-21, models/instructblip.py:17-21).
On the CPU path that generator is a 32-bit mt19937 with the classic Knuth
initialisation; a float32 uniform is one 32-bit output `x` mapped to
`(x & 0xFFFFFF) * 2**-24`, and consecutive `rand` calls continue the same stream.
(Verified against `torch.manual_seed(s); torch.rand(n)` in tests/test_oracle_rng.py
and pinned by tests/golden/g6_rng.npz.)

The state layout (624 words + index) is the same one `dd_rng_mt19937_*` in
include/dropdec.h keeps in device memory.
"""
from __future__ import annotations

import numpy as np

N, M = 624, 397
_UPPER, _LOWER, _MATRIX_A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF


def seed_state(seed: int) -> np.ndarray:
    """625 uint32 words: mt[0..623] then the read index (624 = "twist before next draw")."""
    st = np.empty(N + 1, dtype=np.uint32)
    x = seed & 0xFFFFFFFF
    st[0] = x
    for j in range(1, N):
        x = (1812433253 * (x ^ (x >> 30)) + j) & 0xFFFFFFFF
        st[j] = x
    st[N] = N
    return st


def _twist(mt: np.ndarray) -> None:
    """Regenerate all 624 words in place (sequential dependency kept exactly)."""
    mt64 = mt.astype(np.uint64)
    for i in range(N):
        y = (mt64[i] & _UPPER) | (mt64[(i + 1) % N] & _LOWER)
        v = mt64[(i + M) % N] ^ (y >> np.uint64(1))
        if int(y) & 1:
            v ^= np.uint64(_MATRIX_A)
        mt64[i] = v
    mt[:] = mt64.astype(np.uint32)


def _temper(y: np.ndarray) -> np.ndarray:
    y = y.astype(np.uint64)
    y ^= y >> np.uint64(11)
    y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
    y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
    y ^= y >> np.uint64(18)
    return (y & np.uint64(0xFFFFFFFF)).astype(np.uint32)


class TorchCpuMT19937:
    """Restates `torch.manual_seed(seed)` + successive `torch.rand(n, dtype=float32)` on CPU."""

    def __init__(self, seed: int):
        self.state = seed_state(seed)

    def raw(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint32)
        mt, got = self.state[:N], 0
        while got < n:
            idx = int(self.state[N])
            if idx >= N:
                _twist(mt)
                idx = 0
            take = min(n - got, N - idx)
            out[got:got + take] = _temper(mt[idx:idx + take])
            self.state[N] = idx + take
            got += take
        return out

    def rand_f32(self, n: int) -> np.ndarray:
        """n float32 uniforms in [0,1): (x & 0xFFFFFF) * 2**-24, one draw per element."""
        return ((self.raw(n) & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
