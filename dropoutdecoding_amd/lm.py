"""DropoutEngine — Python handle on the dd_lm_* C-ABI (the K-way masked-context decode step).

It replaces, for one sequence, what the reference's forward() does per token: the un-masked pass, the keep
set, the K dropout masks, the K masked forwards on copied KV caches, the vote and the return of the
winner's logits + cache (reference models/llava.py:254-376; llavanext.py:490-600; instructblip.py:59-165).
torch supplies device memory and the stream; the arithmetic is in libdropdec.so.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass
from typing import Tuple, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .config import settings
from .dropout import (MASK_IBLIP_KL, MASK_IBLIP_QUANTILE, MASK_LLAVA_CUMULATIVE, MASK_LLAVA_CUMULATIVE_NO_OVERLAP, MASK_NEXT_NO_OVERLAP,
                      MASK_NEXT_RESET,
                      TorchCpuCompatRNG, TorchGpuCompatRNG)

FAMILY_LLAVA = "llava-1.5"
FAMILY_NEXT = "llava-next"
FAMILY_IBLIP = "instructblip"

VOTE_LOGITS, VOTE_HIDDEN, VOTE_AVERAGE = 0, 1, 2

# per-family behaviour pinned by SURVEY.md 8(a) Q1-Q5 and tests/golden/g5_*
_FAMILY = {
    FAMILY_LLAVA: dict(k_top=5, mask_mode=MASK_LLAVA_CUMULATIVE, vote_on=VOTE_LOGITS, leak_mask=0, seed=24),
    FAMILY_NEXT: dict(k_top=10, mask_mode=MASK_NEXT_RESET, vote_on=VOTE_LOGITS, leak_mask=0, seed=506),
    FAMILY_IBLIP: dict(k_top=10, mask_mode=MASK_IBLIP_QUANTILE, vote_on=VOTE_HIDDEN, leak_mask=1, seed=5217),
}

(GET_TOKENS, GET_LOGITS, GET_EPI, GET_ALEA, GET_VAR, GET_UNCERT_SCALARS, GET_TOPK_IDS, GET_TOPK_VALS, GET_DROP,
 GET_N_DROP, GET_MEMBER_ARGMAX, GET_WINNER, GET_BASE_LOGITS, GET_IMAGE_LOGITS, GET_KEEP, GET_KV_SUMS, GET_SEQ_LEN,
 GET_HIDDEN, GET_SPEC_OK) = range(19)

(T_EMBED, T_ATTN_NORM, T_WQ, T_WK, T_WV, T_WO, T_MLP_NORM, T_WGATE, T_WUP, T_WDOWN, T_FINAL_NORM, T_LM_HEAD) = range(12)

_LAYER_TENSORS = {
    "input_layernorm.weight": T_ATTN_NORM, "self_attn.q_proj.weight": T_WQ, "self_attn.k_proj.weight": T_WK,
    "self_attn.v_proj.weight": T_WV, "self_attn.o_proj.weight": T_WO, "post_attention_layernorm.weight": T_MLP_NORM,
    "mlp.gate_proj.weight": T_WGATE, "mlp.up_proj.weight": T_WUP, "mlp.down_proj.weight": T_WDOWN,
}


@dataclass
class LMConfig:
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    head_dim: int = 128
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0

    @classmethod
    def from_hf(cls, tc) -> "LMConfig":
        rp = getattr(tc, "rope_parameters", None) or {}
        theta = rp.get("rope_theta", getattr(tc, "rope_theta", 10000.0))
        return cls(tc.vocab_size, tc.hidden_size, tc.intermediate_size, tc.num_hidden_layers, tc.num_attention_heads,
                   getattr(tc, "num_key_value_heads", tc.num_attention_heads),
                   getattr(tc, "head_dim", None) or tc.hidden_size // tc.num_attention_heads,
                   tc.rms_norm_eps, float(theta))


LLAVA15_7B = LMConfig(32064, 4096, 11008, 32, 32, 32, 128, 1e-5, 10000.0)
VICUNA_7B = LMConfig(32001, 4096, 11008, 32, 32, 32, 128, 1e-6, 10000.0)
MISTRAL_7B = LMConfig(32064, 4096, 14336, 32, 32, 8, 128, 1e-5, 1000000.0)


def _eos_ids(eos) -> tuple:
    if eos is None:
        return ()
    if isinstance(eos, (list, tuple, set, frozenset)):
        return tuple(sorted({int(t) for t in eos}))
    return (int(eos),)


def quantize_fp8(w: torch.Tensor):
    """Per-output-row absmax quantisation to OCP fp8 e4m3fn: w ~= scale[:, None] * q.  Returns (q as uint8 [N, K],
    scale fp32 [N]).  Build-defined (the reference has no fp8 path): BASELINE config 5."""
    wf = w.float()
    scale = wf.abs().amax(dim=1).clamp(min=1e-12) / 448.0
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale.float()


def dequantize_fp8(q_u8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q_u8.view(torch.float8_e4m3fn).float() * scale[:, None]


class DropoutEngine:
    def __init__(self, cfg: LMConfig, family: str = FAMILY_LLAVA, max_seq: int = 1280, max_visual: int = 576,
                 seed: Optional[int] = None, use_random: bool = False, device: Optional[torch.device] = None,
                 iblip_positions: str = "cache", weight_format: str = "bf16", mask_method: str = "epis",
                 use_avg: bool = False, share_weights_with: Optional["DropoutEngine"] = None, kv_format: str = "fp32",
                 rng_stream: str = "cpu", lib=None, tp: Optional[Tuple[int, int]] = None):
        """tp = (rank, world): this engine is one tensor-parallel shard — `cfg` then carries the rank's LOCAL head counts and
        intermediate size (tp_local_config) and the engine is driven through TensorParallelGroup / dist.TensorParallelRank."""
        if family not in _FAMILY:
            raise ValueError(f"unknown family {family!r}")
        if not torch.cuda.is_available():
            raise _lib.DDError("DropoutEngine needs a GPU (MI355X); there is no CPU fallback for the product path")
        # lib: another instance of the library than the process's (bench.py's roofline leg: _lib.load_tools())
        self.lib = lib if lib is not None else (share_weights_with.lib if share_weights_with is not None else _lib.load())
        self.cfg, self.family = cfg, family
        # remembered so that a lane over these weights can be created with the same behaviour (DropoutVLM.spawn_lane)
        self.max_seq, self.max_visual, self.use_random, self.iblip_positions = max_seq, max_visual, use_random, iblip_positions
        self.mask_method, self.use_avg = mask_method, use_avg
        if kv_format not in ("fp32", "fp16"):
            raise ValueError(f"kv_format {kv_format!r}: 'fp32' or 'fp16' (the reference's cache width)")
        self.kv_format = kv_format
        fam = dict(_FAMILY[family])
        if family == FAMILY_NEXT and use_random:
            fam["mask_mode"] = MASK_NEXT_NO_OVERLAP           # settings['use_random'][0] (llavanext.py:547-550)
        if mask_method == "epis_no_overlap":                   # dormant variant (llava.py:663-683, instructblip.py:486-505)
            fam["mask_mode"] = MASK_LLAVA_CUMULATIVE_NO_OVERLAP if family == FAMILY_LLAVA else MASK_NEXT_NO_OVERLAP
        elif mask_method == "epis_kl":                         # dormant variant, InstructBLIP only (instructblip.py:123, 464-485)
            if family != FAMILY_IBLIP:
                raise ValueError("mask_method 'epis_kl' exists for InstructBLIP only (models/instructblip.py:464-485)")
            fam["mask_mode"] = MASK_IBLIP_KL
        elif mask_method != "epis":
            raise ValueError(f"mask_method {mask_method!r}: 'epis' (the shipped call sites), 'epis_no_overlap' or 'epis_kl'")
        if use_avg:
            fam["vote_on"] = VOTE_AVERAGE                       # select_by_average, llava.py:37-52 (settings['use_avg'])
        if family == FAMILY_IBLIP and iblip_positions == "mask":
            fam["leak_mask"] = 2                                # transformers 4.44 position rule
        self.k_top = fam["k_top"]
        dev = torch.device(device or "cuda")
        self.device = torch.device("cuda", torch.cuda.current_device() if dev.index is None else dev.index)
        torch.cuda.set_device(self.device)
        c = _lib.LMConfigC(cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size, cfg.num_layers, cfg.num_heads,
                           cfg.num_kv_heads, cfg.head_dim, cfg.rms_eps, cfg.rope_theta, max_seq, max_visual,
                           fam["k_top"], fam["mask_mode"], fam["vote_on"], fam["leak_mask"],
                           {"bf16": 0, "fp8": 1, "fp16": 2}[weight_format], {"fp32": 0, "fp16": 1}[kv_format])
        self.tp = tp
        if tp is not None:
            c.reserved[0], c.reserved[1] = int(tp[1]), int(tp[0])
        self.weight_format = weight_format
        self._h = C.c_void_p()
        self.weight_owner = share_weights_with         # kept alive: a lane borrows the owner's weight memory
        if share_weights_with is not None:
            # a lane: another sequence (own KV cache, state, rng stream) over the same weights — see EngineGroup
            if share_weights_with.weight_owner is not None:
                share_weights_with = self.weight_owner = share_weights_with.weight_owner
            self._ck(self.lib.dd_lm_create_shared(C.byref(c), share_weights_with._h, C.byref(self._h)), "dd_lm_create_shared")
        else:
            self._ck(self.lib.dd_lm_create(C.byref(c), C.byref(self._h)), "dd_lm_create")
        # the reference seeds torch's global generator at import (llava.py:16-20); under chair_test all three
        # modules are imported so 5217 is in force. Default here: the family's own module seed.
        self.seed = fam["seed"] if seed is None else seed
        # which torch generator the draws of llava.py:650 restate: "cpu" = mt19937 (the reference run on CPU, the golden
        # fixtures), "gpu" = Philox4x32-10 (the reference run on a GPU; equals torch.rand(..., device="cuda") on ROCm)
        if rng_stream not in ("cpu", "gpu"):
            raise ValueError(f"rng_stream {rng_stream!r}: 'cpu' (mt19937) or 'gpu' (Philox)")
        self.rng_stream = rng_stream
        self.rng = (TorchGpuCompatRNG if rng_stream == "gpu" else TorchCpuCompatRNG)(self.seed, lib=self.lib)
        # the engine enqueues on its own (non-default) stream: decode steps can then be captured into hipGraphs, and
        # torch work of the caller (next image's preprocessing) does not interleave with the dependent chain
        self.torch_stream = (share_weights_with.torch_stream if share_weights_with is not None
                             else torch.cuda.Stream(device=self.device))     # lanes of one owner share its stream
        self.L = 0
        self.masked_numbers: List[int] = []
        self._peek_buf = np.zeros(8192, dtype=np.int32)
        self._n_enqueued = 0
        self._eos_dev: Optional[tuple] = None        # eos ids currently held in the sequence's device state
        self.sync_steps = True                       # generate(): host-decided fallback of the speculative step

    def _s(self) -> int:
        return self.torch_stream.cuda_stream

    def _ck(self, rc: int, what: str) -> None:
        _lib.check(rc, what, self.lib)              # the error text of THIS engine's library instance

    # ---- weights ---------------------------------------------------------------------------
    def _load(self, tid: int, layer: int, t: torch.Tensor) -> None:
        t = t.detach()
        if t.dim() == 1:
            t = t[None]
        t = t.to(torch.float16 if self.weight_format == "fp16" else torch.bfloat16).contiguous()
        self._ck(self.lib.dd_lm_load_tensor(self._h, tid, layer, t.view(torch.int16).data_ptr(), t.shape[0],
                                              t.shape[1], 1 if t.is_cuda else 0), f"dd_lm_load_tensor({tid},{layer})")

    def _load_fp8(self, tid: int, layer: int, t: torch.Tensor) -> None:
        q, s = quantize_fp8(t.detach())
        q, s = q.contiguous(), s.contiguous()
        self._ck(self.lib.dd_lm_load_tensor_fp8(self._h, tid, layer, q.data_ptr(), s.data_ptr(), q.shape[0], q.shape[1],
                                                  1 if q.is_cuda else 0), f"dd_lm_load_tensor_fp8({tid},{layer})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        """HF LlamaForCausalLM / MistralForCausalLM parameter names (optionally under `prefix`).  An fp8 engine
        quantises every matrix with `quantize_fp8` (per-row absmax / 448, OCP e4m3fn) on the way in; a bf16 engine rounds to
        bf16 (exact for bf16 checkpoints), an fp16 engine to fp16 (exact for the fp16 checkpoints the reference loads)."""
        g = lambda k: sd[prefix + k]
        mat = self._load_fp8 if self.weight_format == "fp8" else self._load
        self._load(T_EMBED, 0, g("model.embed_tokens.weight"))
        self._load(T_FINAL_NORM, 0, g("model.norm.weight"))
        mat(T_LM_HEAD, 0, g("lm_head.weight"))
        for i in range(self.cfg.num_layers):
            for name, tid in _LAYER_TENSORS.items():
                (self._load if tid in (T_ATTN_NORM, T_MLP_NORM) else mat)(tid, i, g(f"model.layers.{i}.{name}"))

    def load_synthetic(self, seed: int = 0, std: float = 0.02) -> None:
        self._ck(self.lib.dd_lm_load_synthetic(self._h, seed, std), "dd_lm_load_synthetic")

    @property
    def device_bytes(self) -> int:
        return int(self.lib.dd_lm_device_bytes(self._h))

    # ---- the path ---------------------------------------------------------------------------
    def prefill(self, embeds: torch.Tensor, span_start: int, span_len: int, first_step_ensemble: bool = False,
                mprobs: Optional[Sequence[float]] = None, uniforms: Optional[torch.Tensor] = None,
                stream: Optional[torch.cuda.Stream] = None) -> None:
        """`stream`: run this prefill on another stream than the engine's (GroupPipeline overlaps the next batch's prefill
        with the current batch's decode); the caller orders the engine's stream after it (event) before decoding."""
        if not embeds.is_cuda:
            raise ValueError("embeds must be on the GPU")
        e = embeds.reshape(-1, embeds.shape[-1]).float().contiguous()
        if e.shape[1] != self.cfg.hidden_size:
            raise ValueError(f"embeds have width {e.shape[1]}, model hidden size is {self.cfg.hidden_size}")
        pstream = stream if stream is not None else self.torch_stream
        pstream.wait_stream(torch.cuda.current_stream(self.device))   # embeds come from the caller's stream
        e.record_stream(pstream)
        ps = pstream.cuda_stream
        K = 0
        if first_step_ensemble:
            # the reference's `# if True:` toggle (llava.py:336-337): the ensemble also picks the first token
            probs, arr = self._probs(mprobs)
            K = len(probs)
            un = None
            if uniforms is not None:
                un = uniforms.float().contiguous()
                un.record_stream(pstream)
            self._ck(self.lib.dd_lm_prefill_ensemble(self._h, e.data_ptr(), e.shape[0], span_start, span_len, arr, K,
                                                       self.rng.handle, un.data_ptr() if un is not None else None,
                                                       ps), "dd_lm_prefill_ensemble")
        else:
            self._ck(self.lib.dd_lm_prefill(self._h, e.data_ptr(), e.shape[0], span_start, span_len, ps),
                       "dd_lm_prefill")
        self.L, self.T0 = span_len, e.shape[0]
        self._last_K = K
        self._n_enqueued = 1                       # the prefill's greedy token

    def truncate(self, T_keep: int, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Cut the sequence back to its first T_keep positions (>= end of the visual span); see prefill_extend."""
        pstream = stream if stream is not None else self.torch_stream
        if stream is not None:
            pstream.wait_stream(self.torch_stream)
        self._ck(self.lib.dd_lm_truncate(self._h, int(T_keep), pstream.cuda_stream), "dd_lm_truncate")
        self.T0 = int(T_keep)
        self._last_K = 0
        self._n_enqueued = 0

    def prefill_extend(self, embeds: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Append more PROMPT positions to a prefilled / truncated sequence (chunked prefill against the cache) and emit
        the greedy first token, as a full prefill of the longer prompt would.  With truncate(): several questions about
        one image (POPE asks 6) without re-running the visual positions."""
        e = embeds.reshape(-1, embeds.shape[-1]).float().contiguous()
        if not e.is_cuda or e.shape[1] != self.cfg.hidden_size:
            raise ValueError("embeds must be [n, hidden] on the GPU")
        pstream = stream if stream is not None else self.torch_stream
        pstream.wait_stream(torch.cuda.current_stream(self.device))
        e.record_stream(pstream)
        self._ck(self.lib.dd_lm_prefill_extend(self._h, e.data_ptr(), e.shape[0], pstream.cuda_stream), "dd_lm_prefill_extend")
        self.T0 += e.shape[0]
        self._last_K = 0
        self._n_enqueued = 1

    def _probs(self, mprobs):
        probs = list(settings["voting_numbers"] if mprobs is None else mprobs)   # read at every step (llava.py:340)
        return probs, (C.c_double * max(len(probs), 1))(*[float(p) for p in probs])

    def decode_step(self, mprobs: Optional[Sequence[float]] = None, uniforms: Optional[torch.Tensor] = None,
                    dropout: bool = True) -> None:
        """Enqueue one ensemble step (no host sync). dropout=False is the stock greedy step (`--original`)."""
        probs, arr = self._probs(mprobs)
        K = len(probs) if dropout else 0
        un = None
        if uniforms is not None:
            un = uniforms.float().contiguous()
            self._keepalive = un
            self.torch_stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ck(self.lib.dd_lm_decode_step(self._h, arr, K, self.rng.handle, un.data_ptr() if un is not None else None,
                                              self._s()), "dd_lm_decode_step")
        self._last_K = K
        self._n_enqueued += 1

    def decode_step_sync(self, mprobs: Optional[Sequence[float]] = None, dropout: bool = True) -> int:
        """One ensemble step with the speculative step's fallback decided by the host (dd_lm_decode_step_sync): the calling
        thread waits for the check (microseconds after the sweep) and the members are re-run only when needed.  Same results
        as decode_step().  Returns 1 (the speculative members stood), 0 (re-run) or -1 (went through decode_step's path)."""
        probs, arr = self._probs(mprobs)
        K = len(probs) if dropout else 0
        held = C.c_int(-1)
        self._ck(self.lib.dd_lm_decode_step_sync(self._h, arr, K, self.rng.handle, self._s(), C.byref(held)),
                   "dd_lm_decode_step_sync")
        self._last_K = K
        self._n_enqueued += 1
        return held.value

    _SPEC_MODES = {"default": -1, "never": 0, "always": 1, "adaptive": 2}

    def set_speculation(self, mode: str) -> None:
        """When single-sequence steps take the speculative one-sweep form (dd_lm_set_speculation): 'never' (always the
        un-masked sweep, then the members), 'always', 'adaptive' (speculate while enough of the recent checks held, the
        library's default) or 'default' (the process-wide default).  Results never depend on it."""
        self._ck(self.lib.dd_lm_set_speculation(self._h, self._SPEC_MODES[mode]), "dd_lm_set_speculation")

    def spec_stats(self, reset: bool = False) -> Dict[str, float]:
        """Counts since creation / the last reset: speculative steps that held, that were re-run, plain two-sweep steps the
        adaptive policy issued, times it switched speculation off; hit_rate = held / speculative steps."""
        out = (C.c_int64 * 4)()
        self._ck(self.lib.dd_lm_spec_stats(self._h, out, 1 if reset else 0), "dd_lm_spec_stats")
        held, rerun, plain, off = (int(x) for x in out)
        return {"held": held, "rerun": rerun, "plain": plain, "switched_off": off,
                "hit_rate": (held / (held + rerun)) if held + rerun else None,
                "sweeps_per_step": ((held + 2 * rerun + 2 * plain) / (held + rerun + plain)) if held + rerun + plain else None}

    # phased form for K-sharding (see dist.py)
    def step_base(self, mprobs=None, uniforms=None) -> int:
        probs, arr = self._probs(mprobs)
        un = uniforms.float().contiguous() if uniforms is not None else None
        self._keepalive = un
        if un is not None:
            self.torch_stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ck(self.lib.dd_lm_step_base(self._h, arr, len(probs), self.rng.handle,
                                            un.data_ptr() if un is not None else None, self._s()), "dd_lm_step_base")
        self._last_K = len(probs)
        return len(probs)

    def step_members(self, m_lo: int, m_hi: int) -> None:
        self._ck(self.lib.dd_lm_step_members(self._h, m_lo, m_hi, self._s()), "dd_lm_step_members")

    def step_commit(self) -> None:
        self._ck(self.lib.dd_lm_step_commit(self._h, self._last_K, self._s()), "dd_lm_step_commit")
        self._n_enqueued += 1

    # exchange records for K-sharding (dist.py); tensors are torch CUDA tensors owned by the caller
    def xchg_stride(self) -> int:
        return int(self.lib.dd_lm_xchg_stride(self._h))

    def export_ids(self, m_lo: int, m_hi: int, ids: torch.Tensor) -> None:
        self._ck(self.lib.dd_lm_xchg_export_ids(self._h, m_lo, m_hi, ids.data_ptr(), self._s()), "dd_lm_xchg_export_ids")

    def import_ids(self, ids: torch.Tensor) -> None:
        self._ck(self.lib.dd_lm_xchg_import_ids(self._h, ids.data_ptr(), self._s()), "dd_lm_xchg_import_ids")

    def export_winner(self, m_lo: int, m_hi: int, rec: torch.Tensor) -> None:
        self._ck(self.lib.dd_lm_xchg_export_winner(self._h, m_lo, m_hi, rec.data_ptr(), self._s()), "dd_lm_xchg_export_winner")

    def import_winner(self, rec: torch.Tensor) -> None:
        self._ck(self.lib.dd_lm_xchg_import_winner(self._h, rec.data_ptr(), self._s()), "dd_lm_xchg_import_winner")

    def new_xchg_buffers(self):
        return (torch.zeros(32, dtype=torch.int32, device=self.device),
                torch.zeros(self.xchg_stride(), dtype=torch.float32, device=self.device))

    def set_eos(self, eos) -> None:
        """EOS ids of HF's greedy loop, kept in the sequence's DEVICE state: the step that emits one ends
        the sequence there, and steps already enqueued beyond it are no-ops — they draw nothing from the rng stream, so
        the next image continues the stream exactly where the reference's would (models/llava.py:16-20, :650)."""
        ids = _eos_ids(eos)
        if self._eos_dev == ids:
            return
        if len(ids) > 8:
            raise ValueError("at most 8 eos ids")
        arr = (C.c_int32 * max(len(ids), 1))(*ids)
        self._ck(self.lib.dd_lm_set_eos(self._h, arr, len(ids), self._s()), "dd_lm_set_eos")
        self._eos_dev = ids

    def set_next_token(self, token: int) -> None:
        self._ck(self.lib.dd_lm_set_next_token(self._h, int(token), self._s()), "dd_lm_set_next_token")

    def peek_tokens(self) -> List[int]:
        """Tokens emitted so far WITHOUT synchronising (pinned host mirror written by the step kernels)."""
        buf = self._peek_buf
        n = self.lib.dd_lm_peek_tokens(self._h, buf.ctypes.data, buf.size)
        return buf[:n].tolist()

    def generate(self, n_new: int, eos=None, mprobs=None, dropout: bool = True, lookahead: int = 6, step_fn=None) -> List[int]:
        """Greedy loop of HF `_sample`: the prefill's token first, then ensemble steps until EOS or n_new.
        Steps are enqueued without synchronising; the host watches the pinned token mirror and stops enqueueing as soon
        as an EOS appears.  The stop itself is device-side (set_eos): up to `lookahead` steps enqueued beyond the EOS
        step run as no-ops that neither emit tokens nor consume the rng stream, so results do not depend on host
        timing."""
        eos_set = set(_eos_ids(eos))
        self._sync_eos(eos_set)
        if step_fn is None and dropout and self.sync_steps and 1 <= len(self._probs(mprobs)[0]) <= 8:
            # one sequence on its own: the host decides the speculative step's fallback (it waits for the step's check, so no
            # queue of steps builds up ahead of the GPU; a step enqueued after the EOS step — the mirror lags by one — is a
            # device-side no-op).  While the adaptive policy issues plain two-sweep steps those are queued like decode_step's
            # and the look-ahead limit applies.
            step_fn = lambda: self.decode_step_sync(mprobs)
        step = step_fn or (lambda: self.decode_step(mprobs, dropout=dropout))
        enq = self._n_enqueued
        while enq < n_new:
            seen = self.peek_tokens()
            if eos_set and any(t in eos_set for t in seen):
                break
            if enq - len(seen) >= lookahead:          # far enough ahead of the GPU: let it catch up
                time.sleep(0.0002)
                continue
            step()
            enq = self._n_enqueued
        toks = self.tokens()              # the device stopped at the EOS step; nothing to cut off
        self._n_enqueued = len(toks)
        return toks[:n_new]

    def _sync_eos(self, eos_set) -> None:
        """The device-side eos list must be the loop's: set it before the first step is enqueued.  If the prefill's greedy
        token (emitted before the list was known) already is an EOS, the loop below enqueues nothing, so no flag is needed."""
        self.set_eos(sorted(eos_set))

    # ---- read-backs (synchronise) -------------------------------------------------------------
    def _get(self, what: int, n: int, dtype) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        self._ck(self.lib.dd_lm_get(self._h, what, out.ctypes.data, out.nbytes, self._s()), f"dd_lm_get({what})")
        return out

    def n_tokens(self) -> int:
        self.torch_stream.synchronize()
        return self.T() - self.T0 + 1

    def T(self) -> int:
        return int(self._get(GET_SEQ_LEN, 1, np.int32)[0])

    def tokens(self) -> List[int]:
        n = self.T() - self.T0 + 1
        return self._get(GET_TOKENS, n, np.int32).tolist()

    def logits(self) -> np.ndarray:
        return self._get(GET_LOGITS, self.cfg.vocab_size, np.float32)

    def base_logits(self) -> np.ndarray:
        return self._get(GET_BASE_LOGITS, self.cfg.vocab_size, np.float32)

    def hidden(self) -> np.ndarray:
        return self._get(GET_HIDDEN, self.cfg.hidden_size, np.float32)

    def image_logits(self) -> np.ndarray:
        return self._get(GET_IMAGE_LOGITS, self.L * self.cfg.vocab_size, np.float32).reshape(self.L, -1)

    def vision_uncert_dict(self) -> Dict[str, np.ndarray]:
        sc = self._get(GET_UNCERT_SCALARS, 3, np.float32)
        return {"variance_per_token": self._get(GET_VAR, self.L, np.float32)[None],
                "epis_uncert_per_token": self._get(GET_EPI, self.L, np.float32)[None],
                "alea_uncert_per_token": self._get(GET_ALEA, self.L, np.float32)[None],
                "variance": sc[0:1], "epis_uncert": sc[1:2], "alea_uncert": sc[2:3]}

    def topk(self):
        return (self._get(GET_TOPK_VALS, self.L * self.k_top, np.float32).reshape(self.L, self.k_top),
                self._get(GET_TOPK_IDS, self.L * self.k_top, np.int32).reshape(self.L, self.k_top))

    def last_step(self) -> Dict[str, np.ndarray]:
        K = self._last_K
        w = self._get(GET_WINNER, 2, np.int32)
        return {"drop": self._get(GET_DROP, K * self.L, np.uint8).reshape(K, self.L).astype(bool),
                "masked_numbers": self._get(GET_N_DROP, K, np.int32), "keep": self._get(GET_KEEP, self.L, np.uint8).astype(bool),
                "member_argmax": self._get(GET_MEMBER_ARGMAX, K, np.int32), "winner": int(w[0]), "voted": int(w[1])}

    def spec_ok(self) -> int:
        """1: the last single-sequence step finished in one sweep (speculative masks stood); 0: the members were re-run."""
        return int(self._get(GET_SPEC_OK, 1, np.int32)[0])

    def kv_sums(self) -> np.ndarray:
        return self._get(GET_KV_SUMS, self.cfg.num_layers * 2, np.float64).reshape(-1, 2)

    def algorithmic_bytes(self, K: int) -> float:
        return float(self.lib.dd_lm_step_algorithmic_bytes(self._h, K))

    # measurement hooks: only on engines created through libdropdec_tools.so (lib=_lib.load_tools())
    def time_sweep(self, nb: int, iters: int) -> float:
        ms = C.c_float()
        self._ck(self.lib.dd_lm_time_sweep(self._h, nb, iters, C.byref(ms), self._s()), "dd_lm_time_sweep")
        return float(ms.value)

    def time_gemv(self, which: int, nb: int, iters: int):
        """(mean ms per launch, algorithmic bytes per launch) of one decode GEMV kind; 0 qkv, 1 o, 2 gate/up, 3 down."""
        ms, by = C.c_float(), C.c_double()
        self._ck(self.lib.dd_lm_time_gemv(self._h, which, nb, iters, C.byref(ms), C.byref(by), self._s()), "dd_lm_time_gemv")
        return float(ms.value), float(by.value)

    def last_gemv_kernel(self) -> str:
        """Name (as rocprofv3's kernel trace prints it) of the streaming kernel the last time_gemv() launched."""
        return self.lib.dd_tools_last_gemv_kernel().decode()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.dd_lm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- tensor-parallel decode (include/dropdec.h dd_lm_tp_*) ---------------------------------------------
def tp_local_config(cfg: LMConfig, world: int) -> LMConfig:
    """A rank's LOCAL dimensions: heads and kv heads / world, d_ff / world padded up to a multiple of 256 (the padding rows of
    gate / up and columns of down are zeros: silu(0) * 0 = 0 contributes nothing)."""
    if cfg.num_kv_heads % world or cfg.num_heads % world or (cfg.num_heads // world) % 2 or cfg.intermediate_size % (16 * world):
        raise ValueError(f"{cfg.num_heads} heads / {cfg.num_kv_heads} kv heads / d_ff {cfg.intermediate_size} do not split over {world} ranks "
                         "(whole kv-head groups, an even number of q heads and whole 16-row tiles per rank)")
    ff = cfg.intermediate_size // world
    return LMConfig(cfg.vocab_size, cfg.hidden_size, (ff + 255) // 256 * 256, cfg.num_layers, cfg.num_heads // world,
                    cfg.num_kv_heads // world, cfg.head_dim, cfg.rms_eps, cfg.rope_theta)


def tp_shard_state_dict(sd: Dict[str, torch.Tensor], cfg: LMConfig, rank: int, world: int, prefix: str = "") -> Dict[str, torch.Tensor]:
    """This rank's slices of a LlamaForCausalLM state dict (dist.TensorParallelPlan.shard_layer per layer, d_ff padded);
    embeddings, norm vectors and lm_head are replicated."""
    from .dist import TensorParallelPlan
    plan = TensorParallelPlan(cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size, world)
    ff_pad = tp_local_config(cfg, world).intermediate_size
    out = {k: sd[prefix + k] for k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}
    for i in range(cfg.num_layers):
        lp = f"model.layers.{i}."
        for name, t in plan.shard_layer(sd, prefix + lp, rank).items():
            pad = ff_pad - cfg.intermediate_size // world
            if pad and name in ("mlp.gate_proj.weight", "mlp.up_proj.weight"):
                t = torch.cat([t, t.new_zeros(pad, t.shape[1])], dim=0)
            elif pad and name == "mlp.down_proj.weight":
                t = torch.cat([t, t.new_zeros(t.shape[0], pad)], dim=1)
            out[lp + name] = t
    return out


class TensorParallelGroup:
    """All `world` ranks of a sharded model in ONE process on one device ("linked": dd_lm_tp_link) — the form a single GPU can
    run and test: every rank's kernels are issued in lock step on one stream and the seams read the shared gather buffer.
    Same surface as DropoutEngine where it matters (prefill / decode_step / generate / tokens / logits / last_step)."""

    def __init__(self, cfg: LMConfig, world: int, family: str = FAMILY_LLAVA, max_seq: int = 1280, max_visual: int = 576,
                 seed: Optional[int] = None, **kw):
        self.cfg, self.world = cfg, world
        local = tp_local_config(cfg, world)
        self.ranks = [DropoutEngine(local, family=family, max_seq=max_seq, max_visual=max_visual, seed=seed, tp=(r, world), **kw)
                      for r in range(world)]
        e0 = self.ranks[0]
        for e in self.ranks[1:]:
            e.torch_stream = e0.torch_stream                     # one stream: the ranks' phases are issued in lock step
        self.lib = e0.lib
        self._hs = (C.c_void_p * world)(*[e._h for e in self.ranks])
        e0._ck(self.lib.dd_lm_tp_link(self._hs, world, max_seq), "dd_lm_tp_link")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        for r, e in enumerate(self.ranks):
            e.load_state_dict(tp_shard_state_dict(sd, self.cfg, r, self.world, prefix))

    def manual_seed(self, seed: int) -> None:
        for e in self.ranks:
            e.rng.manual_seed(seed)

    def prefill(self, embeds: torch.Tensor, span_start: int, span_len: int) -> None:
        e0 = self.ranks[0]
        x = embeds.reshape(-1, embeds.shape[-1]).float().contiguous()
        if not x.is_cuda or x.shape[1] != self.cfg.hidden_size:
            raise ValueError("embeds must be [T0, hidden] on the GPU")
        e0.torch_stream.wait_stream(torch.cuda.current_stream(e0.device))
        x.record_stream(e0.torch_stream)
        e0._ck(self.lib.dd_lm_tp_prefill(self._hs, self.world, x.data_ptr(), x.shape[0], span_start, span_len, e0._s()), "dd_lm_tp_prefill")
        for e in self.ranks:
            e.L, e.T0, e._last_K, e._n_enqueued = span_len, x.shape[0], 0, 1

    def decode_step(self, mprobs: Optional[Sequence[float]] = None, dropout: bool = True) -> None:
        e0 = self.ranks[0]
        probs, arr = e0._probs(mprobs)
        K = len(probs) if dropout else 0
        rs = (C.c_void_p * self.world)(*[e.rng.handle for e in self.ranks])
        e0._ck(self.lib.dd_lm_tp_decode_step(self._hs, self.world, arr, K, rs, e0._s()), "dd_lm_tp_decode_step")
        for e in self.ranks:
            e._last_K = K
            e._n_enqueued += 1

    def generate(self, n_new: int, mprobs=None, eos=None, dropout: bool = True) -> List[int]:
        eos_set = set(_eos_ids(eos))
        for e in self.ranks:
            e.set_eos(sorted(eos_set))
        toks = self.ranks[0].tokens()
        while len(toks) < n_new and not (eos_set and toks[-1] in eos_set):
            self.decode_step(mprobs, dropout=dropout)
            toks = self.ranks[0].tokens()
        return toks[:n_new]

    # every rank holds the same tokens, logits, masks: read rank 0's (tests compare the others)
    def tokens(self) -> List[int]:
        return self.ranks[0].tokens()

    def logits(self) -> np.ndarray:
        return self.ranks[0].logits()

    def last_step(self):
        return self.ranks[0].last_step()

    def close(self) -> None:
        for e in reversed(self.ranks):
            e.close()


def prefill_group(engines: Sequence["DropoutEngine"], embeds: Sequence[torch.Tensor], spans: Sequence[Tuple[int, int]],
                  stream: Optional[torch.cuda.Stream] = None) -> None:
    """DropoutEngine.prefill for several sequences over one set of weights in ONE pass over the weights
    (dd_lm_prefill_group): engines[i] receives embeds[i] with the visual span spans[i] = (start, length).  Each engine ends up
    bit for bit as its own prefill() would leave it."""
    if not (len(engines) == len(embeds) == len(spans)) or not 1 <= len(engines) <= 32:
        raise ValueError("prefill_group: one embeds tensor and one span per engine, 1..32 of them")
    e0 = engines[0]
    pstream = stream if stream is not None else e0.torch_stream
    pstream.wait_stream(torch.cuda.current_stream(e0.device))
    es = []
    for eng, x in zip(engines, embeds):
        if not x.is_cuda:
            raise ValueError("embeds must be on the GPU")
        x = x.reshape(-1, x.shape[-1]).float().contiguous()
        if x.shape[1] != eng.cfg.hidden_size:
            raise ValueError(f"embeds have width {x.shape[1]}, model hidden size is {eng.cfg.hidden_size}")
        x.record_stream(pstream)
        es.append(x)
    n = len(engines)
    hs = (C.c_void_p * n)(*[e._h for e in engines])
    ps = (C.c_void_p * n)(*[x.data_ptr() for x in es])
    t0 = (C.c_int32 * n)(*[x.shape[0] for x in es])
    s0 = (C.c_int32 * n)(*[int(s[0]) for s in spans])
    sl = (C.c_int32 * n)(*[int(s[1]) for s in spans])
    _lib.check(e0.lib.dd_lm_prefill_group(hs, n, ps, t0, s0, sl, pstream.cuda_stream), "dd_lm_prefill_group")
    for eng, x, sp in zip(engines, es, spans):
        eng.L, eng.T0 = int(sp[1]), x.shape[0]
        eng._last_K = 0
        eng._n_enqueued = 1


class EngineGroup:
    """Up to 64 sequences ("lanes") decoded together over ONE set of weights (dd_lm_group_step).

    The reference decodes one image at a time and shards 500 images over processes, each with its own torch generator
    (chair_test.py:270-346).  A lane is such a process: its own KV cache, state and rng stream.  Every lane's
    tokens, masks and logits are bit-identical to decoding it alone; what changes is the cost — the un-masked base
    passes of all lanes run as one sweep over the weights, so a token costs 1/n + 1 sweeps instead of 2.
    """

    def __init__(self, engines: Sequence[DropoutEngine]):
        engines = list(engines)
        if not 1 <= len(engines) <= 64:
            raise ValueError("a group holds 1..64 sequences")
        owner = engines[0].weight_owner or engines[0]
        for e in engines:
            if (e.weight_owner or e) is not owner:
                raise ValueError("all sequences of a group must share one set of weights (share_weights_with=...)")
        self.engines = engines
        self.lib = engines[0].lib
        self._ck = engines[0]._ck

    def decode_step(self, mprobs: Optional[Sequence[float]] = None, dropout: bool = True,
                    active: Optional[Sequence[int]] = None) -> None:
        """One token for every (active) lane; enqueued without host sync."""
        lanes = [self.engines[i] for i in (range(len(self.engines)) if active is None else active)]
        if not lanes:
            return
        probs, arr = lanes[0]._probs(mprobs)
        K = len(probs) if dropout else 0
        hs = (C.c_void_p * len(lanes))(*[e._h for e in lanes])
        rs = (C.c_void_p * len(lanes))(*[e.rng.handle for e in lanes])
        self._ck(self.lib.dd_lm_group_step(hs, len(lanes), arr, K, rs, lanes[0]._s()), "dd_lm_group_step")
        for e in lanes:
            e._last_K = K
            e._n_enqueued += 1

    def generate(self, n_new: int, eos=None, mprobs=None, dropout: bool = True, lookahead: int = 6,
                 idle=None) -> List[List[int]]:
        """Greedy loops of all lanes in lockstep (each lane as DropoutEngine.generate): a lane stops at its EOS or at
        n_new; the others go on with fewer rows in the fused base pass.  `idle()` (optional) is called while the GPU has
        `lookahead` steps queued; it does one unit of other host work (e.g. enqueue the next image's prefill on another
        stream) and returns False when it has nothing left."""
        eos_set = set(_eos_ids(eos))
        E = self.engines
        for e in E:
            e._sync_eos(eos_set)
        while True:
            seen = [e.peek_tokens() for e in E]
            active = [i for i, e in enumerate(E)
                      if e._n_enqueued < n_new and not (eos_set and any(t in eos_set for t in seen[i]))]
            if not active:
                break
            if min(E[i]._n_enqueued - len(seen[i]) for i in active) >= lookahead:
                if idle is None or not idle():
                    time.sleep(0.0002)                  # far enough ahead of the GPU
                continue
            # whole groups of eight keep the rider form of the step (csrc/dd_engine.hip group_step_rider): sequences that ended at an EOS
            # stay in the line-up while they fill the last group — their steps are no-ops on the device (DDState::done: nothing is
            # written, nothing drawn from their rng streams)
            # (K <= 4: groups of fourteen — two sequences per operand plane, seven planes + two riding planes)
            group = 14 if (dropout and len(E[0]._probs(mprobs)[0]) <= 4 and len(E) % 14 == 0 and len(active) >= 28) else 8
            if eos_set and len(active) >= 16 and len(active) % group:
                # (a filler's host-side length still advances with every step it rides along: only lanes whose cache has room for
                # all the steps the longest active lane may still take)
                left = n_new - min(E[i]._n_enqueued for i in active)
                cap = lambda e: (e.max_seq + 63) // 64 * 64
                ended = [i for i in range(len(E)) if i not in active and any(t in eos_set for t in seen[i])
                         and E[i].T0 + E[i]._n_enqueued + left + 1 < cap(E[i])]
                need = -len(active) % group
                if len(ended) >= need:
                    active = sorted(active + ended[:need])
            self.decode_step(mprobs, dropout=dropout, active=active)
        out = []
        for e in E:
            toks = e.tokens()             # each lane stopped at its own EOS step on the device
            e._n_enqueued = len(toks)
            out.append(toks[:n_new])
        return out
