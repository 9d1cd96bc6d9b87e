"""Reference-faithful Dropout-Decoding driver on CPU (oracle; test infrastructure only).

Restates the control flow of the reference's `forward()` overrides:
  LLaVA-1.5     models/llava.py:218-226, 254-314, 336-376
  LLaVA-NeXT    models/llavanext.py:490-600
  InstructBLIP  models/instructblip.py:59-165 (+ generate :588-697 for the span)
with the same cost structure — per decoded token `1 + K` *sequential* batch-1 LM
forwards, each on a *copy* of the whole KV cache (the reference's `copy.deepcopy`) —
so it doubles as the timed CPU baseline (`bench.py`, cpu_baseline.kind = "port").
Checked end-to-end against tests/golden/g5_*.npz (the reference's own forward run in
the build container through the compatibility shim in oracle/gen_golden.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dropout_ref as DR
from .lm_ref import KVCache, LMConfig, lm_hidden, lm_logits
from .mt19937 import TorchCpuMT19937
from .philox import TorchGpuPhilox

FAMILY_LLAVA = "llava-1.5"
FAMILY_NEXT = "llava-next"
FAMILY_IBLIP = "instructblip"

K_TOP = {FAMILY_LLAVA: 5, FAMILY_NEXT: 10, FAMILY_IBLIP: 10}     # llava.py:408, llavanext.py:652, instructblip.py:187


@dataclass
class StepRecord:
    base_argmax: int
    keep: np.ndarray                  # bool [L]
    uniforms: Optional[np.ndarray]    # fp32 [K, L] or None (InstructBLIP)
    drop: np.ndarray                  # bool [K, L]
    masked_numbers: List[int]
    member_argmax: List[int]
    winner: int
    token: int
    logits: np.ndarray                # fp32 [V] (winner's)
    base_logits: np.ndarray           # fp32 [V]
    phases: Dict[str, float] = field(default_factory=dict)
    # diagnostics for the parity tests (not part of the reference's state): top-1 minus top-2 of what each member's
    # argmax ran over (llava.py:27 / instructblip.py:125-137), relative to the largest magnitude — tells a legitimate
    # near-tie flip (inside the 1e-3 logits tolerance) from a wrong result
    member_margin: List[float] = field(default_factory=list)


class RefDecoder:
    def __init__(self, family: str, cfg: LMConfig, weights: Dict[str, torch.Tensor],
                 mprobs: Sequence[float], seed: int = 5217, use_random: bool = False,
                 dropout: bool = True, iblip_positions: str = "cache", mask_method: str = "epis",
                 use_avg: bool = False, first_step_ensemble: bool = False, rng_stream: str = "cpu"):
        self.family, self.cfg, self.w = family, cfg, weights
        self.mprobs = list(mprobs)
        # the generator torch.rand_like draws from (llava.py:650): mt19937 on CPU, Philox4x32-10 on a GPU
        self.rng = TorchGpuPhilox(seed) if rng_stream == "gpu" else TorchCpuMT19937(seed)
        self.use_random = use_random            # settings['use_random'][0], llavanext.py:547
        self.dropout = dropout                  # False = the `--original` path (stock greedy)
        self.iblip_positions = iblip_positions  # "cache" (transformers 5.x) | "mask" (4.44 cumsum rule), SURVEY Q2
        self.mask_method = mask_method          # "epis" | "epis_no_overlap" (dormant: llava.py:663-683)
        self.use_avg = use_avg                  # select_by_average instead of select_by_vote (dormant: llava.py:37-52)
        self.first_step_ensemble = first_step_ensemble   # the `# if True:` toggle at llava.py:336-337
        self.cache = KVCache()
        self.dtype = weights["lm_head.weight"].dtype
        self.leaked = None                      # InstructBLIP: last member's drop flags (Q2)

    # ---- prefill: reference llava.py:218-226, 285-314 -------------------------------
    def prefill(self, embeds: torch.Tensor, span_start: int, span_len: int) -> int:
        T0 = embeds.shape[0]
        self.span_start, self.L = span_start, span_len
        self.cache = KVCache()
        self.leaked = None
        hid = lm_hidden(self.cfg, self.w, embeds.to(self.dtype), torch.arange(T0), self.cache)
        logits = lm_logits(self.cfg, self.w, hid)                       # fp32 [T0, V]  (llava.py:294-305)
        img = logits[span_start:span_start + span_len]                  # llava.py:412-426
        self.topk_vals, self.topk_ids = DR.topk_tokens(img, K_TOP[self.family])     # llava.py:310
        self.uncert = DR.vision_uncertainty(img[None])                  # llava.py:311-314
        self.epi = self.uncert["epis_uncert_per_token"][0]
        self.image_logits = img
        self.prefill_logits = logits
        return int(torch.argmax(logits[-1]))                            # HF greedy on the prefill logits (Q9)

    def _mode(self) -> int:
        if self.family == FAMILY_LLAVA:
            return DR.MODE_LLAVA_CUMULATIVE_NO_OVERLAP if self.mask_method == "epis_no_overlap" else DR.MODE_LLAVA_CUMULATIVE
        if self.mask_method == "epis_kl":
            return DR.MODE_IBLIP_KL
        if self.mask_method == "epis_no_overlap":
            return DR.MODE_NEXT_NO_OVERLAP
        if self.family == FAMILY_NEXT:
            return DR.MODE_NEXT_NO_OVERLAP if self.use_random else DR.MODE_NEXT_RESET
        return DR.MODE_IBLIP_QUANTILE

    def _select(self, member_logits, member_hid):
        """-> (winner index, logits, ids).  Vote (llava.py:22-36 / instructblip.py:125-137) or mean (llava.py:37-52)."""
        if self.use_avg:
            ids = [int(torch.argmax(l)) for l in member_logits]
            mean = torch.from_numpy(torch.stack(member_logits).numpy().mean(axis=0))     # numpy fp32 mean, llava.py:47
            return 0, mean, ids
        if self.family == FAMILY_IBLIP:
            ids = [int(torch.argmax(h[-1])) for h in member_hid]
        else:
            ids = [int(torch.argmax(l)) for l in member_logits]
        win, _ = DR.vote(ids)
        return win, member_logits[win], ids

    def prefill_first_step(self, embeds: torch.Tensor, span_start: int, span_len: int) -> int:
        """prefill() followed by the ensemble on the FIRST token (llava.py:336 with `if True:`): each member re-runs the
        whole prompt from an empty cache with its zero columns; the winner's logits and cache continue."""
        self.prefill(embeds, span_start, span_len)
        T0 = embeds.shape[0]
        base_logits = self.prefill_logits[-1]
        keep = DR.overlap_keep(base_logits, self.topk_ids)
        K = len(self.mprobs)
        mode = self._mode()
        uniforms = None
        if mode != DR.MODE_IBLIP_QUANTILE:
            uniforms = torch.from_numpy(np.stack([self.rng.rand_f32(self.L) for _ in range(K)]))
        drop = DR.sample_masks(self.epi, self.mprobs, keep, mode, uniforms)
        member_logits, member_hid, member_cache = [], [], []
        for k in range(K):
            c = KVCache()
            km = torch.ones(T0, dtype=torch.long)
            km[span_start:span_start + span_len][drop[k]] = 0
            hid = lm_hidden(self.cfg, self.w, embeds.to(self.dtype), torch.arange(T0), c, km)
            member_hid.append(hid[-1:])
            member_logits.append(lm_logits(self.cfg, self.w, hid[-1:])[0])
            member_cache.append(c)
        win, logits, ids = self._select(member_logits, member_hid)
        self.cache = member_cache[win]
        self.first_record = StepRecord(int(torch.argmax(base_logits)), keep.numpy().copy(),
                                       None if uniforms is None else uniforms.numpy().copy(), drop.numpy().copy(),
                                       [int(d.sum()) for d in drop], ids, win, int(torch.argmax(logits)),
                                       logits.numpy().copy(), base_logits.numpy().copy(), {})
        return int(torch.argmax(logits))

    def embed(self, token: int) -> torch.Tensor:
        return self.w["model.embed_tokens.weight"][token][None].to(self.dtype)

    def _forward_one(self, x, pos, cache, key_mask):
        hid = lm_hidden(self.cfg, self.w, x, torch.tensor([pos]), cache, key_mask)
        return hid

    # ---- one decode step: reference llava.py:254-283, 292-305, 336-376 ---------------
    def step(self, token: int) -> StepRecord:
        import time
        ph = {"copy": 0.0, "lm": 0.0, "mask": 0.0, "vote": 0.0}
        x = self.embed(token)
        T = self.cache.length
        base_mask = torch.ones(T + 1, dtype=torch.long)                 # llava.py:266-282
        pos = T                                                          # llava.py:283 (sum(mask) - 1)
        if self.family == FAMILY_IBLIP and self.leaked is not None:     # Q2: caller's mask mutated in place
            base_mask[self.span_start:self.span_start + self.L][self.leaked] = 0
            if self.iblip_positions == "mask":
                pos = int(base_mask.sum()) - 1
        t0 = time.perf_counter()
        original = self.cache.clone()                                    # llava.py:292
        if self.family == FAMILY_NEXT:
            _ = self.cache.clone()                                       # llavanext.py:502 (second copy)
        ph["copy"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        hid = self._forward_one(x, pos, self.cache, base_mask)           # llava.py:294-303 (mutates live cache)
        base_logits = lm_logits(self.cfg, self.w, hid)[0]
        ph["lm"] += time.perf_counter() - t0
        if not self.dropout:
            tok = int(torch.argmax(base_logits))
            return StepRecord(tok, np.zeros(self.L, bool), None, np.zeros((0, self.L), bool), [], [tok], 0, tok,
                              base_logits.numpy().copy(), base_logits.numpy().copy(), ph)
        t0 = time.perf_counter()
        K = len(self.mprobs)
        mode = self._mode()
        if mode == DR.MODE_IBLIP_KL:
            keep = DR.kl_keep(self.image_logits, base_logits)            # instructblip.py:483-485
        else:
            keep = DR.overlap_keep(base_logits, self.topk_ids)           # llava.py:603, 443-482
        uniforms = None
        if mode != DR.MODE_IBLIP_QUANTILE:
            # one rand_like(epi) per member, in list order (llava.py:650); contiguous stream
            uniforms = torch.from_numpy(np.stack([self.rng.rand_f32(self.L) for _ in range(K)]))
        drop = DR.sample_masks(self.epi, self.mprobs, keep, mode, uniforms)
        ph["mask"] += time.perf_counter() - t0
        member_hid, member_cache = [], []
        for k in range(K):                                               # llava.py:342-359 (sequential)
            t0 = time.perf_counter()
            c = original.clone()                                         # llava.py:343
            ph["copy"] += time.perf_counter() - t0
            km = torch.ones(T + 1, dtype=torch.long)
            km[self.span_start:self.span_start + self.L][drop[k]] = 0
            t0 = time.perf_counter()
            member_hid.append(self._forward_one(x, pos, c, km))
            ph["lm"] += time.perf_counter() - t0
            member_cache.append(c)
        t0 = time.perf_counter()
        member_logits = [lm_logits(self.cfg, self.w, h)[0] for h in member_hid]
        # Q3: InstructBLIP votes on argmax over the flattened final hidden state (instructblip.py:125-137); LLaVA / NeXT on
        # the logits argmax (llava.py:27, 361); `use_avg` takes the mean instead (llava.py:37-52)
        win, logits, ids = self._select(member_logits, member_hid)
        if self.family == FAMILY_IBLIP:
            self.leaked = drop[K - 1].clone()
        ph["vote"] += time.perf_counter() - t0
        voted = [h[-1].float() for h in member_hid] if (self.family == FAMILY_IBLIP and not self.use_avg) else member_logits
        margins = []
        for v in voted:
            top2 = torch.topk(v.float().flatten(), 2).values
            margins.append(float((top2[0] - top2[1]) / v.float().abs().max().clamp_min(1e-30)))
        self.cache = member_cache[win]                                   # llava.py:373
        tok = int(torch.argmax(logits))                                  # HF greedy
        return StepRecord(int(torch.argmax(base_logits)), keep.numpy().copy(),
                          None if uniforms is None else uniforms.numpy().copy(), drop.numpy().copy(),
                          [int(d.sum()) for d in drop], ids, win, tok, logits.numpy().copy(),
                          base_logits.numpy().copy(), ph, margins)

    def generate(self, embeds: torch.Tensor, span_start: int, span_len: int, n_new: int,
                 eos: Optional[int] = None) -> List[int]:
        tok = (self.prefill_first_step if (self.first_step_ensemble and self.dropout) else self.prefill)(embeds, span_start, span_len)
        out = [tok]
        self.records = []
        while len(out) < n_new and (eos is None or tok != eos):
            rec = self.step(tok)
            self.records.append(rec)
            tok = rec.token
            out.append(tok)
        return out
