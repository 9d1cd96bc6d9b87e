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
    EVENT_RING = 256      # timed tokens kept as event pairs before they are folded into the running totals

    def __init__(self, engine, rank: int, world: int, group=None, time_exchange: bool = False):
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self.ids, self.rec = engine.new_xchg_buffers()
        # time_exchange: bracket every token's exchange (export ids -> all-reduce -> import -> export winner -> all-reduce ->
        # import) with events on the engine's stream; exchange_ms() reports the mean (bench.py --mode kshard)
        self._events = [] if time_exchange else None
        self._ms_sum, self._ms_n = 0.0, 0          # totals of the pairs already read back (the list is drained every EVENT_RING tokens)

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
        ev = None
        if self._events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        e.export_ids(lo, hi, self.ids)
        dist.all_reduce(self.ids, op=dist.ReduceOp.SUM, group=self.group)
        e.import_ids(self.ids)
        e.export_winner(lo, hi, self.rec)
        dist.all_reduce(self.rec, op=dist.ReduceOp.SUM, group=self.group)
        e.import_winner(self.rec)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
            if len(self._events) >= self.EVENT_RING:
                self._drain()
        e.step_commit()

    def exchange_ms(self, reset: bool = False):
        """Mean milliseconds per token of the two collectives with their export / import kernels, and the bytes moved per rank."""
        if self._events is None:
            return None
        self._drain()
        if not self._ms_n:
            return None
        out = {"ms_per_token": round(self._ms_sum / self._ms_n, 4), "tokens": self._ms_n, "bytes_per_token": int(self.ids.numel() * 4 + self.rec.numel() * 4),
               "collectives_per_token": 2, "backend": dist.get_backend(self.group), "world": self.world}
        if reset:
            self._ms_sum, self._ms_n = 0.0, 0
        return out

    def _drain(self) -> None:
        if self._events:
            self._events[-1][1].synchronize()
            self._ms_sum += sum(a.elapsed_time(b) for a, b in self._events)
            self._ms_n += len(self._events)
            self._events = []

    def generate(self, n_new: int, mprobs=None) -> List[int]:
        toks = self.e.tokens()
        while len(toks) < n_new:
            for _ in range(min(16, n_new - len(toks))):
                self.decode_step(mprobs)
            toks = self.e.tokens()
        return toks[:n_new]


# ---------------------------------------------------------------------------------------------------------------------
# Tensor-parallel decode: the only route to single-stream multi-GPU speed-up — K-sharding leaves
# every rank streaming all weights twice per token.  The shard plan (which slice of which tensor a rank holds, head-aligned)
# and the collectives a sweep needs are below; the engine side is csrc/dd_tp.hip (dd_lm_tp_*): a rank's row-parallel
# matrices stop at their partial product, the partials are all-gathered at the two seams of a layer and every rank adds
# them in rank order before the fused epilogue.  `TensorParallelRank` is the one-process-per-GPU driver: it registers the
# all-gather (torch.distributed: "nccl" = RCCL over xGMI; "gloo" stages through the host for the tests) as the engine's
# exchange.  lm.TensorParallelGroup is the single-process form (all ranks on one device).
# ---------------------------------------------------------------------------------------------------------------------
_TP_ERROR_KEY = "dropoutdecoding_amd/tp_exchange_error"
_TP_POLL_S = 0.05             # a rank looks at its job's flag at most this often (one store round trip per look)
_tp_generation: dict = {}      # ranks of a group -> TensorParallelRank objects built on it so far (the same count on every rank)


def _tp_store():
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def tp_error_key(group, generation: int = 0) -> str:
    """The store key of ONE tensor-parallel job: the group's global ranks + the how-many-th job on that group this is.  Sub-groups
    that share the default store and successive jobs on one group therefore never see each other's flag."""
    try:
        ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
    except Exception:
        ranks = []
    return f"{_TP_ERROR_KEY}/{'-'.join(map(str, ranks))}/{generation}"


def tp_flag_error(group, rank: int, what: str, generation: int = 0) -> None:
    """Mark the tensor-parallel job as failed in the rendezvous store (visible to every rank, whatever the backend)."""
    _tp_store().set(tp_error_key(group, generation), f"rank {rank}: {what}"[:400])


def tp_peer_failed(group, generation: int = 0) -> str:
    """'' or the failed rank's message.  No default store (a group built without one) is the only thing read as "nothing to check
    against"; a store that fails to answer raises — a broken store must not look like a healthy job."""
    try:
        st = _tp_store()
    except Exception:
        return ""
    key = tp_error_key(group, generation)
    if st.check([key]):
        return st.get(key).decode(errors="replace")
    return ""


def tp_clear_error(group, generation: int = 0) -> bool:
    """Remove the job's flag (teardown).  False when the store kind cannot delete keys (FileStore): the key is namespaced per job, so
    a flag left behind is never read by a later job."""
    try:
        return bool(_tp_store().delete_key(tp_error_key(group, generation)))
    except Exception:
        return False


class TensorParallelRank:
    """One rank of a sharded model in its own process.  `cfg` is the FULL model's LMConfig; the rank's engine holds its slices."""

    def __init__(self, cfg, rank: int, world: int, group=None, family: str = "llava-1.5", max_seq: int = 1280, max_visual: int = 576,
                 seed: Optional[int] = None, **engine_kw):
        import ctypes as C
        from . import _lib, lm
        self.cfg, self.rank, self.world, self.group = cfg, rank, world, group
        self.engine = lm.DropoutEngine(lm.tp_local_config(cfg, world), family=family, max_seq=max_seq, max_visual=max_visual, seed=seed,
                                       tp=(rank, world), **engine_kw)
        e = self.engine
        self.rows_cap = max_seq + 64
        self.gather = torch.zeros(world * self.rows_cap * cfg.hidden_size, dtype=torch.float32, device=e.device)
        self.exchanges = 0
        self.error = None
        # which job on this group this is: every rank builds its TensorParallelRank objects in the same order, so the count agrees
        gkey = tp_error_key(group).rsplit("/", 1)[0]
        self.generation = _tp_generation[gkey] = _tp_generation.get(gkey, -1) + 1
        self._last_poll = 0.0
        d = cfg.hidden_size

        def exchange(ctx, rows, stream):
            # (no store round trip in here: there are two exchanges per layer and sweep — ~128 per token — and on "nccl" they are
            # asynchronous and latency-bound.  The job's flag is looked at once per prefill / decode step, `_poll_peers`, and again
            # when an exchange raises)
            try:
                n = rows * d
                out = self.gather[: world * n]
                mine = out[rank * n:(rank + 1) * n]
                with torch.cuda.stream(e.torch_stream):         # the engine enqueues on this stream (== `stream`)
                    if dist.get_backend(self.group) == "nccl":
                        dist.all_gather_into_tensor(out, mine, group=self.group)        # in place: RCCL over xGMI
                    else:                                       # gloo (tests): through the host
                        e.torch_stream.synchronize()
                        parts = [torch.empty(n, dtype=torch.float32) for _ in range(world)]
                        dist.all_gather(parts, mine.cpu(), group=self.group)
                        out.copy_(torch.cat(parts).to(out.device))
                self.exchanges += 1
                return 0
            except Exception as ex:                             # an exception must not unwind through the C frames
                try:                                            # was it a peer leaving first?  say so
                    failed = tp_peer_failed(self.group, self.generation)
                except Exception:
                    failed = ""
                self.error = RuntimeError(f"tensor-parallel exchange failed after a peer's failure ({failed}): {ex!r}") if failed else ex
                self._abort_peers()                             # the other ranks are (or will be) waiting in this all-gather
                return 1
        self._cb = _lib.TP_EXCHANGE_FN(exchange)                # kept alive with the object
        self._hs = (C.c_void_p * 1)(e._h)
        # (the library checks the buffer against the engine's KV capacity: a prefill of max_seq rows must fit every rank's slot)
        e._ck(e.lib.dd_lm_tp_set_exchange(e._h, self.gather.data_ptr(), self.gather.numel(), self._cb, None), "dd_lm_tp_set_exchange")

    def load_state_dict(self, sd, prefix: str = "") -> None:
        from . import lm
        # (the class's method, not the instance attribute: vlm.build_engine(tp=...) points engine.load_state_dict at THIS method, so that the
        # drop-in classes can hand over the full state dict)
        lm.DropoutEngine.load_state_dict(self.engine, lm.tp_shard_state_dict(sd, self.cfg, self.rank, self.world, prefix))

    def _abort_peers(self) -> None:
        """A failed exchange on this rank leaves the others blocked in their all-gather until the process-group timeout.  Two things are done
        about it, neither silently: (1) an error flag in the rendezvous store (`tp_flag_error`), which every rank checks BEFORE it enters an
        exchange (`tp_peer_failed`) — works on every backend, and is all that "gloo" offers: a rank already inside the collective stays there
        until the group's timeout; (2) on "nccl" (RCCL) the communicator is aborted, so that ranks already inside fail now.  Whatever goes wrong
        while doing so is kept in `self.abort_error` and warned about, not swallowed."""
        import warnings
        self.abort_error = None
        try:
            tp_flag_error(self.group, self.rank, repr(self.error), getattr(self, "generation", 0))
        except Exception as ex:
            self.abort_error = ex
            warnings.warn(f"TensorParallelRank: could not flag the failed exchange in the store: {ex!r}")
        try:
            pg = self.group if self.group is not None else dist.group.WORLD
            if dist.get_backend(pg) != "nccl":
                return                                          # gloo: no abort; the flag is what the peers get
            if hasattr(pg, "abort"):
                pg.abort()
            elif hasattr(dist, "_abort_process_group"):
                dist._abort_process_group(pg)
            else:
                dist.destroy_process_group(self.group)
        except Exception as ex:
            self.abort_error = ex
            warnings.warn(f"TensorParallelRank: aborting the process group after a failed exchange raised {ex!r}; "
                          "the peers will time out in their all-gather instead of failing now")

    def _poll_peers(self, force: bool = False) -> None:
        """Once per prefill / decode step, and at most every _TP_POLL_S: has a peer flagged this job as failed?  Then this rank must
        not walk into a collective its peer has left."""
        import time
        now = time.monotonic()
        if not force and now - self._last_poll < _TP_POLL_S:
            return
        self._last_poll = now
        failed = tp_peer_failed(self.group, self.generation)
        if failed:
            raise RuntimeError(f"tensor-parallel job: a peer's exchange failed ({failed})")

    def close(self) -> None:
        """Teardown: the job's flag leaves the store (rank 0 of the job deletes it), the engine is released."""
        if self.rank == 0:
            tp_clear_error(self.group, self.generation)
        self.engine.close()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0 and self.error is not None:
            err, self.error = self.error, None
            raise err
        self.engine._ck(rc, what)

    def prefill(self, embeds: torch.Tensor, span_start: int, span_len: int) -> None:
        e = self.engine
        self._poll_peers(force=True)
        x = embeds.reshape(-1, embeds.shape[-1]).float().contiguous()
        e.torch_stream.wait_stream(torch.cuda.current_stream(e.device))
        x.record_stream(e.torch_stream)
        self._check(e.lib.dd_lm_tp_prefill(self._hs, 1, x.data_ptr(), x.shape[0], span_start, span_len, e._s()), "dd_lm_tp_prefill")
        e.L, e.T0, e._last_K, e._n_enqueued = span_len, x.shape[0], 0, 1

    def decode_step(self, mprobs: Optional[Sequence[float]] = None, dropout: bool = True) -> None:
        import ctypes as C
        e = self.engine
        self._poll_peers()
        probs, arr = e._probs(mprobs)
        K = len(probs) if dropout else 0
        rs = (C.c_void_p * 1)(e.rng.handle)
        self._check(e.lib.dd_lm_tp_decode_step(self._hs, 1, arr, K, rs, e._s()), "dd_lm_tp_decode_step")
        e._last_K = K
        e._n_enqueued += 1

    def generate(self, n_new: int, mprobs=None) -> List[int]:
        toks = self.engine.tokens()
        while len(toks) < n_new:                                # every rank decodes the same tokens: the loops stay in step
            self.decode_step(mprobs)
            toks = self.engine.tokens()
        return toks[:n_new]


# ---------------------------------------------------------------------------------------------------------------------
class TensorParallelPlan:
    """Megatron-style split of one decoder layer over `world` ranks:

      q/k/v_proj   column-parallel, by kv head group (a rank owns kv heads [kv_lo, kv_hi) and their q heads): no comm
      attention    local to the rank's heads (its slice of the KV cache: 1/world of the cache bytes)
      o_proj       row-parallel over the rank's head columns -> partial [rows, d]  -> ALL-REDUCE (sum) #1
      gate/up      column-parallel over d_ff / world                                 no comm
      down_proj    row-parallel over the rank's d_ff slice -> partial [rows, d]     -> ALL-REDUCE (sum) #2
      lm_head      column-parallel over the vocabulary; the argmax needs (value, index) pairs: one small all-gather per sweep

    Per sweep: 2 * n_layers all-reduces of rows * d fp32 (LLaVA-1.5-7B, 9 rows: 147 KB each) — latency-bound on xGMI."""

    def __init__(self, num_heads: int, num_kv_heads: int, head_dim: int, hidden: int, intermediate: int, world: int):
        if num_kv_heads % world or intermediate % (world * 16):
            raise ValueError(f"{num_kv_heads} kv heads / d_ff {intermediate} do not split over {world} ranks on head / tile boundaries")
        self.world, self.H, self.Hkv, self.hd, self.d, self.dff = world, num_heads, num_kv_heads, head_dim, hidden, intermediate
        self.G = num_heads // num_kv_heads

    def kv_heads(self, rank: int) -> Tuple[int, int]:
        per = self.Hkv // self.world
        return rank * per, (rank + 1) * per

    def ff_range(self, rank: int) -> Tuple[int, int]:
        per = self.dff // self.world
        return rank * per, (rank + 1) * per

    def shard_layer(self, sd: dict, prefix: str, rank: int) -> dict:
        """This rank's slices of one layer's HF tensors (norm vectors are replicated)."""
        k0, k1 = self.kv_heads(rank)
        q0, q1 = k0 * self.G * self.hd, k1 * self.G * self.hd
        f0, f1 = self.ff_range(rank)
        g = lambda n: sd[prefix + n]
        return {
            "input_layernorm.weight": g("input_layernorm.weight"), "post_attention_layernorm.weight": g("post_attention_layernorm.weight"),
            "self_attn.q_proj.weight": g("self_attn.q_proj.weight")[q0:q1], "self_attn.k_proj.weight": g("self_attn.k_proj.weight")[k0 * self.hd:k1 * self.hd],
            "self_attn.v_proj.weight": g("self_attn.v_proj.weight")[k0 * self.hd:k1 * self.hd], "self_attn.o_proj.weight": g("self_attn.o_proj.weight")[:, q0:q1],
            "mlp.gate_proj.weight": g("mlp.gate_proj.weight")[f0:f1], "mlp.up_proj.weight": g("mlp.up_proj.weight")[f0:f1],
            "mlp.down_proj.weight": g("mlp.down_proj.weight")[:, f0:f1],
        }

    def collectives_per_sweep(self, n_layers: int, rows: int) -> dict:
        return {"all_reduce": 2 * n_layers, "bytes_each": rows * self.d * 4, "all_gather": 1, "all_gather_bytes": rows * 8 * self.world}

    def weight_bytes_per_rank(self, n_layers: int, vocab: int, bytes_per_weight: int = 2) -> int:
        per_layer = (self.H * self.hd * self.d * 2 + 2 * self.Hkv * self.hd * self.d + 3 * self.d * self.dff) // self.world
        return (n_layers * per_layer + vocab * self.d // self.world) * bytes_per_weight
