"""Minimal LLaMA/Mistral decoder forward on torch-CPU (oracle; test infrastructure only).

Restates the third-party LM the reference calls at models/llava.py:294-303,350-359,
models/llavanext.py:505-514,553-562 and models/instructblip.py:68-82,125-140
(`transformers` LlamaForCausalLM / MistralForCausalLM; pinned 4.44.0 in the reference's
environment.yml:106, source absent from the reference tree).  The op order follows the
installed transformers 5.15 `modeling_llama.py` (RMSNorm in fp32; half-split
`rotate_half` RoPE with fp32 cos/sin; bias-free q/k/v/o and SwiGLU MLP; GQA by
`repeat_kv`; softmax(q.k^T * d_head^-0.5 + additive mask) in fp32; cache append before
attention; final norm; lm_head; logits as fp32 — the 4.44 `logits.float()` rule, cf. the
verbatim copy at reference models/llama.py:46-47).
Checked against HF's own LlamaForCausalLM in tests/test_oracle_lm.py and, through the
reference's forward, by tests/golden/g5_*.npz.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


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

    @property
    def q_dim(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_kv_heads * self.head_dim


LLAVA15_7B = LMConfig(32064, 4096, 11008, 32, 32, 32, 128, 1e-5, 10000.0)
VICUNA_7B = LMConfig(32001, 4096, 11008, 32, 32, 32, 128, 1e-6, 10000.0)
MISTRAL_7B = LMConfig(32064, 4096, 14336, 32, 32, 8, 128, 1e-5, 1000000.0)


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16, returned as fp32 (weights are bf16-valued on both sides)."""
    return t.to(torch.bfloat16).to(torch.float32)


def random_weights(cfg: LMConfig, seed: int, std: float = 0.05, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded bf16-representable weights with HF parameter names (numpy legacy RandomState:
    the stream is frozen across numpy versions, so a fixture only needs the seed)."""
    rs = np.random.RandomState(seed)

    def mat(n, k, s=std):
        return bf16_round(torch.from_numpy(rs.standard_normal((n, k)).astype(np.float32) * s)).to(dtype)

    def vec(n):
        return bf16_round(torch.from_numpy((1.0 + 0.1 * rs.standard_normal(n)).astype(np.float32))).to(dtype)

    w = {"model.embed_tokens.weight": mat(cfg.vocab_size, cfg.hidden_size, 1.0)}
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = vec(cfg.hidden_size)
        w[p + "self_attn.q_proj.weight"] = mat(cfg.q_dim, cfg.hidden_size)
        w[p + "self_attn.k_proj.weight"] = mat(cfg.kv_dim, cfg.hidden_size)
        w[p + "self_attn.v_proj.weight"] = mat(cfg.kv_dim, cfg.hidden_size)
        w[p + "self_attn.o_proj.weight"] = mat(cfg.hidden_size, cfg.q_dim)
        w[p + "post_attention_layernorm.weight"] = vec(cfg.hidden_size)
        w[p + "mlp.gate_proj.weight"] = mat(cfg.intermediate_size, cfg.hidden_size)
        w[p + "mlp.up_proj.weight"] = mat(cfg.intermediate_size, cfg.hidden_size)
        w[p + "mlp.down_proj.weight"] = mat(cfg.hidden_size, cfg.intermediate_size)
    w["model.norm.weight"] = vec(cfg.hidden_size)
    w["lm_head.weight"] = mat(cfg.vocab_size, cfg.hidden_size)
    return w


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_cos_sin(cfg: LMConfig, positions: torch.Tensor, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    freqs = positions.to(torch.float32)[:, None] * inv_freq[None, :]        # [T, d/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _lin(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """nn.Linear without bias, as HF calls it (F.linear on the contiguous [out, in] weight: the batch-1 GEMV then streams
    the weight rows at memory speed — `x @ w.T` in bf16 went through a several-times slower path on the bench host)."""
    return torch.nn.functional.linear(x, w)


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


@dataclass
class KVCache:
    """Per layer K,V as [n_kv, T, d_head] tensors (one sequence)."""
    k: List[torch.Tensor] = field(default_factory=list)
    v: List[torch.Tensor] = field(default_factory=list)

    def clone(self) -> "KVCache":
        """The reference's `copy.deepcopy(past_key_values)` (models/llava.py:292,343)."""
        return KVCache([t.clone() for t in self.k], [t.clone() for t in self.v])

    @property
    def length(self) -> int:
        return 0 if not self.k else self.k[0].shape[1]


def lm_hidden(cfg: LMConfig, w: Dict[str, torch.Tensor], x: torch.Tensor, positions: torch.Tensor,
              cache: KVCache, key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Final-normed hidden states [T_new, d] for new embeddings x [T_new, d]; appends to `cache`.

    key_mask: optional bool/int [T_total] 2-D-style mask over ALL key positions (past + new);
    zeros become an additive finfo.min column exactly as HF builds it from a 2-D
    attention_mask; causal masking among the new tokens is always applied.
    """
    dt = x.dtype
    T_new = x.shape[0]
    cos, sin = rope_cos_sin(cfg, positions, dt)
    past = cache.length
    T_tot = past + T_new
    neg = torch.finfo(dt).min
    add = torch.zeros(T_new, T_tot, dtype=dt)
    if T_new > 1:
        causal = torch.ones(T_new, T_tot, dtype=torch.bool).tril(diagonal=past)
        add = add.masked_fill(~causal, neg)
    if key_mask is not None:
        add = add.masked_fill((key_mask.reshape(1, T_tot) == 0).expand(T_new, T_tot), neg)
    first = len(cache.k) == 0
    h = x
    groups = cfg.num_heads // cfg.num_kv_heads
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        r = h
        hn = rms_norm(h, w[p + "input_layernorm.weight"], cfg.rms_eps)
        q = _lin(hn, w[p + "self_attn.q_proj.weight"]).view(T_new, cfg.num_heads, cfg.head_dim).transpose(0, 1)
        k = _lin(hn, w[p + "self_attn.k_proj.weight"]).view(T_new, cfg.num_kv_heads, cfg.head_dim).transpose(0, 1)
        v = _lin(hn, w[p + "self_attn.v_proj.weight"]).view(T_new, cfg.num_kv_heads, cfg.head_dim).transpose(0, 1)
        q = q * cos[None] + _rotate_half(q) * sin[None]
        k = k * cos[None] + _rotate_half(k) * sin[None]
        if first:
            cache.k.append(k)
            cache.v.append(v)
        else:
            cache.k[i] = torch.cat((cache.k[i], k), dim=1)
            cache.v[i] = torch.cat((cache.v[i], v), dim=1)
        kk = cache.k[i].repeat_interleave(groups, dim=0)
        vv = cache.v[i].repeat_interleave(groups, dim=0)
        att = (q @ kk.transpose(1, 2)) * (cfg.head_dim ** -0.5) + add[None]
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(dt)
        o = (att @ vv).transpose(0, 1).reshape(T_new, cfg.q_dim)
        h = r + _lin(o, w[p + "self_attn.o_proj.weight"])
        r = h
        hn = rms_norm(h, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        g = _lin(hn, w[p + "mlp.gate_proj.weight"])
        u = _lin(hn, w[p + "mlp.up_proj.weight"])
        h = r + _lin(torch.nn.functional.silu(g) * u, w[p + "mlp.down_proj.weight"])
    return rms_norm(h, w["model.norm.weight"], cfg.rms_eps)


def lm_logits(cfg: LMConfig, w: Dict[str, torch.Tensor], hidden: torch.Tensor) -> torch.Tensor:
    return _lin(hidden, w["lm_head.weight"]).float()
