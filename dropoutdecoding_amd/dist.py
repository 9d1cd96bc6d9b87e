"""Multi-GPU forms of the path (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  Nothing in the reference to mirror: it runs one process, the K members sequentially
(models/llava.py:342-359).  SURVEY.md 8(e).

* image replicas — the path shards over independent images: `shard_images()`; no data-path collective.
* K-shard — `KShardDecoder`: every rank keeps a full replica of weights and prefix KV, runs the un-masked pass and
  draws ALL K masks from the shared mt19937 stream (identical on every rank, no comm), runs members
  {m_lo..m_hi}, then per token: all-reduce(sum) of the 2K argmax ids (64 B), device-side vote, all-reduce(sum) of the
  winner's {logits ‖ new KV rows} record with non-owners contributing zeros (= broadcast from a data-dependent root,
  1.1 MB for LLaVA-1.5-7B, latency-bound on xGMI), commit.  Caches stay identical on all ranks.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_images(n_images: int, rank: int, world: int) -> List[int]:
    """Image indices this rank decodes (round-robin, like splitting chair_test's sampled ids over jobs)."""
    return list(range(rank, n_images, world))


def member_range(K: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of the K members; ranks beyond K get an empty range.  Ranges never straddle a
    group of 8 members because the blocks are cut inside groups first (K <= 8 or world divides the groups)."""
    per = -(-K // world)
    lo = min(K, rank * per)
    hi = min(K, lo + per)
    if lo < hi and (lo >> 3) != ((hi - 1) >> 3):
        hi = ((lo >> 3) + 1) << 3
    return lo, hi


class KShardDecoder:
    """Drives the phased engine API (step_base / step_members / export-import / step_commit) with two collectives per
    token.  `engine` is a DropoutEngine (GPU) or any object with the same phase methods (tests use a CPU stand-in)."""

    def __init__(self, engine, rank: int, world: int, group=None):
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self.ids, self.rec = engine.new_xchg_buffers()

    def decode_step(self, mprobs: Optional[Sequence[float]] = None, uniforms=None) -> None:
        st = getattr(self.e, "torch_stream", None)
        if st is None:
            return self._decode_step(mprobs, uniforms)
        with torch.cuda.stream(st):           # the collectives must run in order with the engine's kernels
            return self._decode_step(mprobs, uniforms)

    def _decode_step(self, mprobs: Optional[Sequence[float]] = None, uniforms=None) -> None:
        e = self.e
        K = e.step_base(mprobs, uniforms)
        if K == 0:
            e.step_commit()
            return
        lo, hi = member_range(K, self.rank, self.world)
        covered = set()
        for r in range(self.world):
            a, b = member_range(K, r, self.world)
            covered.update(range(a, b))
        if covered != set(range(K)):
            raise ValueError(f"K={K} members cannot be block-partitioned over {self.world} ranks inside groups of 8")
        if hi > lo:
            e.step_members(lo, hi)
        e.export_ids(lo, hi, self.ids)
        dist.all_reduce(self.ids, op=dist.ReduceOp.SUM, group=self.group)
        e.import_ids(self.ids)
        e.export_winner(lo, hi, self.rec)
        dist.all_reduce(self.rec, op=dist.ReduceOp.SUM, group=self.group)
        e.import_winner(self.rec)
        e.step_commit()

    def generate(self, n_new: int, mprobs=None) -> List[int]:
        toks = self.e.tokens()
        while len(toks) < n_new:
            for _ in range(min(16, n_new - len(toks))):
                self.decode_step(mprobs)
            toks = self.e.tokens()
        return toks[:n_new]
