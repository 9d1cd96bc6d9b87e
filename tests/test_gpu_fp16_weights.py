"""fp16 weight storage (weight_format="fp16"): checkpoints whose tensors are float16 — every model the reference loads
(chair_test/chair_test.py:189-213, torch_dtype=float16) — keep their exact values; activations are split hi + lo in fp16 and
every weight product runs on the f16 MFMA.  Against the fp32 oracle on the SAME fp16-valued weights: tokens, masks, votes exact,
logits to the usual 1e-3 (observed far below).  Next to it: what a cast of the same checkpoint to bf16 costs (the default
weight_format) — the reason the mode exists."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg

RC = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def fp16_weights(cfg, seed, std=0.05):
    """random weights that are fp16-representable but NOT bf16-representable (11 significant bits)"""
    g = torch.Generator().manual_seed(seed)
    r16 = lambda t: t.to(torch.float16).float()
    w = {"model.embed_tokens.weight": r16(torch.randn(cfg.vocab_size, cfg.hidden_size, generator=g))}
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = r16(1.0 + 0.1 * torch.randn(cfg.hidden_size, generator=g))
        w[p + "post_attention_layernorm.weight"] = r16(1.0 + 0.1 * torch.randn(cfg.hidden_size, generator=g))
        w[p + "self_attn.q_proj.weight"] = r16(torch.randn(cfg.q_dim, cfg.hidden_size, generator=g) * std)
        w[p + "self_attn.k_proj.weight"] = r16(torch.randn(cfg.kv_dim, cfg.hidden_size, generator=g) * std)
        w[p + "self_attn.v_proj.weight"] = r16(torch.randn(cfg.kv_dim, cfg.hidden_size, generator=g) * std)
        w[p + "self_attn.o_proj.weight"] = r16(torch.randn(cfg.hidden_size, cfg.q_dim, generator=g) * std)
        w[p + "mlp.gate_proj.weight"] = r16(torch.randn(cfg.intermediate_size, cfg.hidden_size, generator=g) * std)
        w[p + "mlp.up_proj.weight"] = r16(torch.randn(cfg.intermediate_size, cfg.hidden_size, generator=g) * std)
        w[p + "mlp.down_proj.weight"] = r16(torch.randn(cfg.hidden_size, cfg.intermediate_size, generator=g) * std)
    w["model.norm.weight"] = r16(1.0 + 0.1 * torch.randn(cfg.hidden_size, generator=g))
    w["lm_head.weight"] = r16(torch.randn(cfg.vocab_size, cfg.hidden_size, generator=g) * std)
    return w


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(np.asarray(b, np.float64)).max())


@pytest.mark.parametrize("family,K", [(FAMILY_LLAVA, 8), (FAMILY_NEXT, 4), (FAMILY_IBLIP, 3)])
def test_fp16_weights_exact_against_the_oracle(E, family, K):
    w = fp16_weights(RC, 5)
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L = 32 if family == FAMILY_IBLIP else 40
    s0 = 0 if family == FAMILY_IBLIP else 3
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.8
    ref = RefDecoder(family, RC, w, probs, seed=7)
    want = ref.generate(emb, s0, L, 25)
    out = {}
    for fmt in ("fp16", "bf16"):
        eng = E.DropoutEngine(cfg, family=family, max_seq=192, max_visual=L, seed=7, weight_format=fmt)
        eng.load_state_dict(w)
        eng.prefill(emb.cuda(), s0, L)
        worst, agree, masks_ok = rel(eng.image_logits(), ref.prefill_logits[s0:s0 + L].numpy()), 0, True
        for s in range(24):
            eng.decode_step(probs)
            st, r = eng.last_step(), ref.records[s]
            if fmt == "fp16":
                np.testing.assert_array_equal(st["drop"], r.drop, err_msg=f"step {s}")
                assert st["member_argmax"].tolist() == r.member_argmax and st["winner"] == r.winner, f"step {s}"
            else:
                masks_ok &= bool((st["drop"] == r.drop).all())
            if eng.tokens()[:s + 2] == want[:s + 2]:
                agree = s + 1
                worst = max(worst, rel(eng.logits(), r.logits))
        out[fmt] = (worst, agree, masks_ok, eng.tokens())
        eng.close()
    print(f"\n[fp16 checkpoint, {family} K={K}] weight_format=fp16: logits error {out['fp16'][0]:.2e}, tokens exact; "
          f"weight_format=bf16 (cast): logits error {out['bf16'][0]:.2e} over the {out['bf16'][1]} steps its tokens agree, masks equal: {out['bf16'][2]}")
    assert out["fp16"][3] == want and out["fp16"][0] <= 1e-3


@pytest.mark.parametrize("wfmt,kvfmt", [("fp16", "fp16"), ("fp16", "fp32"), ("bf16", "fp16")])
def test_fp16_weights_lanes_speculation_and_7b_shape_kernels(E, wfmt, kvfmt):
    """Every kernel family with the f16 MFMA: nine lanes (one 64-row member pass through the eight-plane slice kernels at 7B
    shapes + one 8-row pass; 32-row fused base pass) bit-identical to solo runs, speculative and two-sweep steps equal."""
    cfg = E.LMConfig(2048, 4096, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L = 24
    engines = []
    for i in range(9):
        engines.append(E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=L + 96, max_visual=L, seed=50 + i, weight_format=wfmt,
                                       kv_format=kvfmt, share_weights_with=engines[0] if engines else None))
    engines[0].load_synthetic(3, 0.02)
    gen = torch.Generator().manual_seed(9)
    embs = [(torch.randn(L + 6 + i, 4096, generator=gen) * 0.5).cuda() for i in range(9)]
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    for e, x in zip(engines, embs):
        e.prefill(x, 2, L)
    grp = E.EngineGroup(engines)
    recs = []
    for s in range(4):
        grp.decode_step(probs)
        recs.append([e.logits().copy() for e in engines])
    toks = [e.tokens() for e in engines]
    assert all(np.isfinite(r).all() for step in recs for r in step)
    lib = engines[0].lib
    for spec in (1, 0):
        lib.dd_set_tuning(14, spec)
        try:
            for li in (2, 8):                                  # a lane of the 64-row pass, and the one of the 8-row pass
                e = engines[li]
                e.rng.manual_seed(50 + li)
                e.prefill(embs[li], 2, L)
                for s in range(4):
                    e.decode_step(probs)
                    np.testing.assert_array_equal(e.logits(), recs[s][li], err_msg=f"spec={spec} lane {li} step {s}")
                assert e.tokens() == toks[li]
        finally:
            lib.dd_set_tuning(14, 1)
    for e in reversed(engines):
        e.close()
