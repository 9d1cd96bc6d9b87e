"""LLaVA-1.5-7B at FULL depth and width against the oracle: 32 layers, d = 4096, d_ff = 11008, V = 32064, 576 visual + 32
prompt tokens (BASELINE configs 2 / 3), K = 8 — prefill, the uncertainty scorer and three ensemble steps of `generate()`'s
engine against the fp32 `RefDecoder` (models/llava.py:229-376) on the host cores, with the margin at every discrete decision.

Opt-in (DD_FULL_DEPTH=1): it needs ~60 GB of host memory (27 GB of fp32 weights + the engine's staging) and a few minutes of
host time, so it is run once per round on the GPU box and its output kept under profiles/ (r06_full_size_oracle.log); the
two-layer tests of tests/test_gpu_7b_shapes_vs_oracle.py are the in-suite anchor of the same kernels.
"""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_LLAVA, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, bf16_round

from test_gpu_7b_shapes_vs_oracle import K8, TOL, LaneCheck


def _weights(rc, seed, std):
    """bf16-representable fp32 weights under HF names (torch's generator: numpy's legacy stream takes minutes for 6.7 G draws)."""
    g = torch.Generator().manual_seed(seed)
    mat = lambda n, k, s=std: bf16_round(torch.randn(n, k, generator=g) * s)
    vec = lambda n: bf16_round(1.0 + 0.1 * torch.randn(n, generator=g))
    w = {"model.embed_tokens.weight": mat(rc.vocab_size, rc.hidden_size, 1.0)}
    for i in range(rc.num_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = vec(rc.hidden_size)
        w[p + "self_attn.q_proj.weight"] = mat(rc.q_dim, rc.hidden_size)
        w[p + "self_attn.k_proj.weight"] = mat(rc.kv_dim, rc.hidden_size)
        w[p + "self_attn.v_proj.weight"] = mat(rc.kv_dim, rc.hidden_size)
        w[p + "self_attn.o_proj.weight"] = mat(rc.hidden_size, rc.q_dim)
        w[p + "post_attention_layernorm.weight"] = vec(rc.hidden_size)
        w[p + "mlp.gate_proj.weight"] = mat(rc.intermediate_size, rc.hidden_size)
        w[p + "mlp.up_proj.weight"] = mat(rc.intermediate_size, rc.hidden_size)
        w[p + "mlp.down_proj.weight"] = mat(rc.hidden_size, rc.intermediate_size)
    w["model.norm.weight"] = vec(rc.hidden_size)
    w["lm_head.weight"] = mat(rc.vocab_size, rc.hidden_size)
    return w


@pytest.mark.skipif(os.environ.get("DD_FULL_DEPTH") != "1", reason="opt-in: DD_FULL_DEPTH=1 (minutes of host time, ~60 GB of host memory)")
def test_llava15_7b_full_depth_prefill_and_three_steps_vs_oracle():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm as E
    dims = (32064, 4096, 11008, 32, 32, 32, 128, 1e-5, 10000.0)
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    t0 = time.time()
    w = _weights(rc, 7, 0.012)
    print(f"\nweights: {sum(v.numel() for v in w.values()) / 1e9:.2f} G parameters in {time.time() - t0:.0f} s")
    L, T0, s0, steps = 576, 608, 5, 3
    emb = torch.randn(T0, 4096, generator=torch.Generator().manual_seed(9)) * 0.5
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=784, max_visual=L, seed=5217, kv_format="fp16")
    eng.load_state_dict(w)
    t0 = time.time()
    ref = RefDecoder(FAMILY_LLAVA, rc, w, K8, seed=5217)
    ref.tokens_out = ref.generate(emb, s0, L, steps + 1)
    print(f"oracle (fp32, {torch.get_num_threads()} threads): prefill + {steps} ensemble steps in {time.time() - t0:.0f} s")
    for mode in ("never", "always"):
        eng.set_speculation(mode)
        eng.rng.manual_seed(5217)
        eng.prefill(emb.cuda(), s0, L)
        eng.set_eos([])
        c = LaneCheck(f"LLaVA-1.5-7B full depth (speculation {mode})", eng, ref, K8)
        c.prefill()
        def rel(a, b):
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            return float(np.abs(a - b).max() / np.abs(b).max())
        print(f"[{mode}] prefill: last-row logits {rel(eng.logits(), ref.prefill_logits[-1].numpy()):.2e}, image-span logits "
              f"{rel(eng.image_logits(), ref.image_logits.numpy()):.2e}, epistemic uncertainty "
              f"{rel(eng.vision_uncert_dict()['epis_uncert_per_token'][0], ref.epi.numpy()):.2e} (relative to the largest); "
              f"first token {eng.tokens()[0]} (oracle {ref.tokens_out[0]}); {c.topk_note or 'top-k id sets equal'}")
        for s in range(steps):
            eng.decode_step(K8)
            c.step(s)
            r, st = ref.records[s], eng.last_step()
            t2 = np.sort(r.logits)[-2:]
            print(f"[{mode}] step {s}: masks equal {bool(np.array_equal(st['drop'], r.drop))} (masked {r.masked_numbers}), member argmax "
                  f"{st['member_argmax'].tolist()} (oracle {r.member_argmax}), winner {st['winner']} ({r.winner}), winner's logits "
                  f"{rel(eng.logits(), r.logits):.2e}, un-masked {rel(eng.base_logits(), r.base_logits):.2e}; oracle margins: token "
                  f"{(t2[1] - t2[0]) / np.abs(r.logits).max():.2e}, smallest member {min(r.member_margin):.2e}; live {c.live}")
        c.tokens()
        print(f"[{mode}] tokens {eng.tokens()} (oracle {ref.tokens_out}); worst logits error {c.worst:.2e} (tolerance {TOL:g})"
              + ("" if c.live else f"; FIRST DIVERGENCE (excused near-tie): {c.excuse}"))
    eng.close()
