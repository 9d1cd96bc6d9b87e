"""Host-side mirror of the reference's dropout-specific methods, backed by the HIP kernels.

Names and argument meaning follow kigb/DropoutDecoding `models/llava.py` so parity tests read like the
reference: calculate_vision_uncertainty (:710-756), get_topk_token_id (:428-441),
get_overlap_image_tokens (:443-482), get_image_attention_mask "epis" (:589-662) and select_by_vote (:22-36).
Tensors are torch CUDA(ROCm) tensors; torch only provides device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib

MASK_LLAVA_CUMULATIVE, MASK_NEXT_RESET, MASK_NEXT_NO_OVERLAP, MASK_IBLIP_QUANTILE = 0, 1, 2, 3
MASK_LLAVA_CUMULATIVE_NO_OVERLAP = 4      # models/llava.py:663-683 (dormant "epis_no_overlap"), cumulative call site
MASK_IBLIP_KL = 5                         # models/instructblip.py:464-485 (dormant "epis_kl"): keep flags from lowest_percent_kl_indices
RNG_INJECTED, RNG_MT19937 = 0, 1


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")


class TorchCpuCompatRNG:
    """mt19937 stream of torch's CPU default generator, kept on the device (dd_rng_*).

    `TorchCpuCompatRNG(seed).rand(n)` returns what `torch.manual_seed(seed); torch.rand(n)` returns on CPU,
    and successive calls continue the stream (reference models/llava.py:16-20, :650).
    """

    def __init__(self, seed: int, lib=None):
        self._lib = lib if lib is not None else _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.dd_rng_create(C.c_uint32(seed & 0xFFFFFFFF), C.byref(self._h)), "dd_rng_create")

    def manual_seed(self, seed: int) -> None:
        _lib.check(self._lib.dd_rng_seed(self._h, C.c_uint32(seed & 0xFFFFFFFF), _stream()), "dd_rng_seed")

    def rand(self, n: int, device="cuda") -> torch.Tensor:
        out = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(self._lib.dd_rng_uniform(self._h, out.data_ptr(), n, _stream()), "dd_rng_uniform")
        return out

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                self._lib.dd_rng_destroy(self._h)
        except Exception:
            pass


class TorchGpuCompatRNG(TorchCpuCompatRNG):
    """Philox stream of torch's GPU default generator, kept on the device (dd_rng_create_philox).

    `TorchGpuCompatRNG(seed).rand(n)` returns what `torch.manual_seed(seed); torch.rand(n, device="cuda")` returns
    (n <= 524288), and successive calls continue the stream: the draws the reference makes at models/llava.py:650
    when it runs on a GPU.  `offset` is the generator's philox offset to start from (a multiple of 4).
    """

    def __init__(self, seed: int, offset: int = 0, lib=None):
        self._lib = lib if lib is not None else _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.dd_rng_create_philox(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(offset),
                                                  C.byref(self._h)), "dd_rng_create_philox")

    def manual_seed(self, seed: int) -> None:
        if not 0 <= seed < 1 << 32:          # dd_rng_seed carries 32 bits; wider seeds go through the constructor
            raise ValueError("manual_seed on a Philox stream takes a 32-bit seed; build TorchGpuCompatRNG(seed) for wider ones")
        super().manual_seed(seed)


def calculate_vision_uncertainty(logits: torch.Tensor, topk: int = 0):
    """logits [1, L, V] (or [L, V]) fp32 on GPU -> dict with the reference's six keys; with topk>0 also
    returns (values, ids) like get_topk_token_id."""
    _need_cuda(logits, "logits")
    x = logits.reshape(-1, logits.shape[-1])
    if x.dtype != torch.float32:
        x = x.float()
    if x.stride(-1) != 1:
        x = x.contiguous()
    L, V = x.shape
    lib = _lib.load()
    dev = x.device
    var, epi, alea = (torch.empty(L, dtype=torch.float32, device=dev) for _ in range(3))
    sc = torch.empty(3, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.dd_uncertainty_workspace_bytes(L, V), dtype=torch.uint8, device=dev)
    vals = torch.empty(L, max(topk, 1), dtype=torch.float32, device=dev)
    ids = torch.empty(L, max(topk, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.dd_vision_uncertainty(x.data_ptr(), L, V, x.stride(0), var.data_ptr(), epi.data_ptr(),
                                         alea.data_ptr(), sc.data_ptr(), topk, vals.data_ptr() if topk else None,
                                         ids.data_ptr() if topk else None, ws.data_ptr(), ws.numel(), _stream()),
               "dd_vision_uncertainty")
    d = {"variance_per_token": var[None], "epis_uncert_per_token": epi[None], "alea_uncert_per_token": alea[None],
         "variance": sc[0:1], "epis_uncert": sc[1:2], "alea_uncert": sc[2:3]}
    if topk:
        return d, (vals[None], ids.long()[None])
    return d


def get_topk_token_id(image_logits: torch.Tensor, topk: int = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    _, tk = calculate_vision_uncertainty(image_logits, topk=topk)
    return tk


def get_overlap_image_tokens(step_logits: torch.Tensor, topk_ids: torch.Tensor, start_image_pos: int = 0):
    """step_logits [..., V] (last position of the un-masked pass), topk_ids [L, k] -> (indices + start, keep u8[L])."""
    _need_cuda(step_logits, "step_logits")
    x = step_logits.reshape(-1).float().contiguous()
    ids = topk_ids.reshape(-1, topk_ids.shape[-1]).to(torch.int32).contiguous()
    L, k = ids.shape
    keep = torch.empty(L, dtype=torch.uint8, device=x.device)
    am = torch.empty(1, dtype=torch.int32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.dd_overlap_keep(x.data_ptr(), x.numel(), ids.data_ptr(), L, k, keep.data_ptr(), am.data_ptr(),
                                   _stream()), "dd_overlap_keep")
    return torch.nonzero(keep).flatten() + start_image_pos, keep


def sample_masks(epi: torch.Tensor, mprobs: Sequence[float], keep: Optional[torch.Tensor], mode: int,
                 uniforms: Optional[torch.Tensor] = None, rng: Optional[TorchCpuCompatRNG] = None,
                 want_indices: bool = False):
    """All K members' drop flags for one step: returns (drop u8 [K, L], n_drop int32 [K][, idx int32 [K, L]])."""
    _need_cuda(epi, "epi")
    e = epi.reshape(-1).float().contiguous()
    L, K = e.numel(), len(mprobs)
    dev = e.device
    drop = torch.empty(K, L, dtype=torch.uint8, device=dev)
    nd = torch.empty(K, dtype=torch.int32, device=dev)
    idx = torch.empty(K, L, dtype=torch.int32, device=dev) if want_indices else None
    arr = (C.c_double * K)(*[float(p) for p in mprobs])
    kp = keep.to(torch.uint8).contiguous() if keep is not None else None
    un = uniforms.float().contiguous() if uniforms is not None else None
    rng_mode = RNG_INJECTED if un is not None or rng is None else RNG_MT19937
    lib = _lib.load()
    _lib.check(lib.dd_sample_masks(e.data_ptr(), L, arr, K, kp.data_ptr() if kp is not None else None, mode, rng_mode,
                                   un.data_ptr() if un is not None else None, rng.handle if rng is not None else None,
                                   drop.data_ptr(), nd.data_ptr(), idx.data_ptr() if idx is not None else None,
                                   _stream()), "dd_sample_masks")
    return (drop, nd, idx) if want_indices else (drop, nd)


def select_by_vote(argmax_ids: torch.Tensor) -> Tuple[int, int]:
    """argmax ids of the K members (int tensor on GPU) -> (winner index, majority id)."""
    _need_cuda(argmax_ids, "argmax_ids")
    ids = argmax_ids.to(torch.int32).contiguous()
    out = torch.empty(2, dtype=torch.int32, device=ids.device)
    _lib.check(_lib.load().dd_vote(ids.data_ptr(), ids.numel(), out.data_ptr(), _stream()), "dd_vote")
    w, t = out.tolist()
    return w, t


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x, "x")
    x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
    out = torch.empty(x2.shape[0], dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().dd_argmax_rows(x2.data_ptr(), x2.shape[0], x2.shape[1], x2.stride(0), out.data_ptr(),
                                          _stream()), "dd_argmax_rows")
    return out


def lowest_percent_kl_indices(image_logits: torch.Tensor, logits: torch.Tensor, percent: float = 0.1) -> torch.Tensor:
    """models/instructblip.py:559-578 (same function at llava.py:758): indices of the int(percent * N) visual tokens whose
    prefill distribution has the smallest KL from the step's distribution.  image_logits [1, N, V] (or [N, V]), logits [1, V]
    (or [V]) fp32 on the GPU -> int64 indices, ascending KL."""
    if percent != 0.1:
        raise ValueError("the C-ABI entry point implements the reference's default percent = 0.1")
    _need_cuda(image_logits, "image_logits")
    img = image_logits.reshape(-1, image_logits.shape[-1]).float().contiguous()
    st = logits.reshape(-1).float().contiguous()
    L, V = img.shape
    keep = torch.empty(L, dtype=torch.uint8, device=img.device)
    kl = torch.empty(L, dtype=torch.float32, device=img.device)
    _lib.check(_lib.load().dd_kl_keep(st.data_ptr(), img.data_ptr(), L, V, img.stride(0), keep.data_ptr(), kl.data_ptr(), _stream()),
               "dd_kl_keep")
    idx = torch.nonzero(keep).flatten()
    return idx[torch.argsort(kl[idx], stable=True)]
