"""Dropout-specific pure functions of the reference, restated on torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference
file:line it follows; citations are into the reference tree (kigb/DropoutDecoding).
Checked against tests/golden/g1..g4 (vectors produced by the reference's own functions).
"""
from __future__ import annotations

from collections import Counter
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

# mask-sampler modes (same numbering as DD_MASK_* in include/dropdec.h)
MODE_LLAVA_CUMULATIVE = 0   # models/llava.py:342-346 — mask NOT reset between members (:344 commented out)
MODE_NEXT_RESET = 1         # models/llavanext.py:546-551 — reset before every member, keep-restore
MODE_NEXT_NO_OVERLAP = 2    # models/llavanext.py:809-829 — reset, no keep-restore ("epis_no_overlap")
MODE_IBLIP_QUANTILE = 3     # models/instructblip.py:447-460 — deterministic top-quantile, reset, keep-restore
MODE_LLAVA_CUMULATIVE_NO_OVERLAP = 4   # models/llava.py:663-683 at the :344 call site — cumulative, no keep-restore
MODE_IBLIP_KL = 5           # models/instructblip.py:464-485 ("epis_kl", the commented call at :123) — reset, stochastic like
                            # NeXT's rule, and the tokens restored are the 10 % with the LOWEST KL(step || token) instead of
                            # the overlap keep set


def vision_uncertainty(logits: torch.Tensor) -> Dict[str, torch.Tensor]:
    """models/llava.py:710-756 (identical copies llavanext.py:878-924, instructblip.py:511-557).

    logits [B, L, V] fp32 -> dict of six tensors, same keys as the reference.
    """
    p = torch.softmax(logits, dim=-1)                                   # :722
    var_tok = torch.var(p, dim=-1)                                      # :728 (unbiased)
    var = var_tok.mean(dim=-1)                                          # :729
    p_avg = p.mean(dim=1)                                               # :732  mean over the L tokens
    epi_tok = (p * (torch.log(p + 1e-10) - torch.log(p_avg.unsqueeze(1) + 1e-10))).sum(dim=-1)  # :735-736
    alea_tok = -(p * torch.log(p + 1e-10)).sum(dim=-1)                  # :739
    return {
        "variance_per_token": var_tok,
        "epis_uncert_per_token": epi_tok,
        "alea_uncert_per_token": alea_tok,
        "variance": var,
        "epis_uncert": epi_tok.mean(dim=-1),                            # :743
        "alea_uncert": alea_tok.mean(dim=-1),                           # :744
    }


def topk_tokens(image_logits: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """models/llava.py:428-441 (`get_topk_token_id`); k=5 LLaVA-1.5 (:408), 10 NeXT/IB."""
    return torch.topk(image_logits, k, dim=-1)


def overlap_keep(step_logits: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
    """models/llava.py:443-482 (`get_overlap_image_tokens`) as a bool[L] keep flag.

    step_logits [V] (the un-masked pass' last-position logits), topk_ids [L, k].
    keep[l] = argmax(step_logits) in topk_ids[l, :].  (The reference returns
    `nonzero(keep).squeeze() + start`; the flag form carries the same information.)
    """
    tok = torch.argmax(step_logits, dim=-1)                             # :459
    return (topk_ids == tok).any(dim=1)                                 # :470-473


def drop_probability(epi: torch.Tensor, mprob: float) -> torch.Tensor:
    """models/llava.py:641-647: p = 0.1 + (mprob-0.1)*(clamp(e,lo,hi)-lo)/(hi-lo), fp32.

    Python evaluates (mprob - 0.1) in double, torch rounds that scalar to fp32 when it
    meets the fp32 tensor; every tensor intermediate is fp32 (SURVEY.md 8a A14 [probe]).
    """
    lo = torch.quantile(epi, 0)                                         # :641 == min
    hi = torch.quantile(epi, 1)                                         # :642 == max
    return 0.1 + (mprob - 0.1) * (epi.clamp(min=lo, max=hi) - lo) / (hi - lo)   # :646-647


def iblip_threshold(epi: torch.Tensor, mprob: float) -> torch.Tensor:
    """models/instructblip.py:450: `torch.quantile(epis_uncert, 1 - prob)` (linear interpolation)."""
    return torch.quantile(epi, 1 - mprob)


def kl_keep(image_logits: torch.Tensor, step_logits: torch.Tensor, percent: float = 0.1) -> torch.Tensor:
    """models/instructblip.py:559-578 (`lowest_percent_kl_indices`; same function at llava.py:758) as a bool[L] flag:
    kl[l] = F.kl_div(log_softmax(image_logits[l]), softmax(step_logits), reduction='none').sum(-1) — i.e.
    KL(step distribution || token l's distribution) — and the int(percent * L) smallest are restored (:483-485)."""
    import torch.nn.functional as F
    img = image_logits.reshape(-1, image_logits.shape[-1])
    kl = F.kl_div(F.log_softmax(img, dim=-1), F.softmax(step_logits.reshape(1, -1), dim=-1).expand_as(img),
                  reduction="none").sum(dim=-1)                          # :567-569
    n = int(percent * kl.numel())                                       # :572
    keep = torch.zeros(kl.numel(), dtype=torch.bool)
    if n > 0:
        keep[torch.topk(kl, n, largest=False).indices] = True           # :575
    return keep


def sample_masks(epi: torch.Tensor, mprobs: Sequence[float], keep: torch.Tensor, mode: int,
                 uniforms: torch.Tensor | None = None) -> torch.Tensor:
    """Per-member visual-token drop flags, bool [K, L] (True = attention mask set to 0).

    uniforms [K, L] are the values `torch.rand_like(epi)` returned for member k
    (models/llava.py:650); unused by MODE_IBLIP_QUANTILE.
    """
    K, L = len(mprobs), epi.numel()
    out = torch.zeros(K, L, dtype=torch.bool)
    running = torch.zeros(L, dtype=torch.bool)       # the in-place-mutated mask, image span only
    for k, mprob in enumerate(mprobs):
        if mode not in (MODE_LLAVA_CUMULATIVE, MODE_LLAVA_CUMULATIVE_NO_OVERLAP):
            running = torch.zeros(L, dtype=torch.bool)                  # llavanext.py:546, instructblip.py:121
        if mode == MODE_IBLIP_QUANTILE:
            drop = epi >= iblip_threshold(epi, mprob)                   # instructblip.py:450-453
        else:
            drop = uniforms[k] < drop_probability(epi, mprob)           # llava.py:650-653
        running = running | drop                                        # llava.py:654-657 (in place)
        if mode not in (MODE_NEXT_NO_OVERLAP, MODE_LLAVA_CUMULATIVE_NO_OVERLAP):
            running = running & ~keep                                   # llava.py:660
        out[k] = running
    return out


def vote(argmax_ids: Sequence[int]) -> Tuple[int, int]:
    """models/llava.py:22-36 (`select_by_vote`): (winner member index, majority token id).

    Counter.most_common(1) returns the first-inserted id among those with the top count;
    the winner is the first member whose argmax equals it.
    """
    c = Counter()
    for t in argmax_ids:
        c[int(t)] += 1
    top = c.most_common(1)[0][0]
    for i, t in enumerate(argmax_ids):
        if int(t) == top:
            return i, top
    raise AssertionError("unreachable")


def entropy_varentropy(logits: torch.Tensor) -> Tuple[float, float]:
    """models/instructblip.py:222-243 (diagnostic appended per step, Q6)."""
    import math
    lp = torch.log_softmax(logits, dim=-1)
    p = torch.exp(lp)
    ent = -(p * lp).sum() / math.log(2)
    ven = (p * (lp / math.log(2) + ent) ** 2).sum()
    return ent.item(), ven.item()
