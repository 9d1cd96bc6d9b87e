"""BASELINE configs 4 and 5 at their real sizes (synthetic weights of the real shapes; the oracle cannot run there in
seconds, so size-independent properties and own-kernel equalities are checked, and the measured ms per step is printed):

  config 4  InstructBLIP-Vicuna-7B, Q-Former's 32 visual tokens, K = 8, the leaked mask of models/instructblip.py:111-122
  config 5  LLaVA-NeXT-Mistral-7B, anyres 5 x 576 + 48 newline tokens (L = 2928, T0 = 3060), GQA 4, K = 8, fp8 weights,
            decoded across two attention launch buckets (T = 3072 and 3328)
Plus: config 5's weight format is reachable from the drop-in class (settings['weight_format'] = 'fp8')."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, bf16_round

PROBS8 = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _timed_steps(eng, probs, n):
    eng.torch_stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.decode_step(probs)
    eng.torch_stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def test_config4_instructblip_vicuna7b_shapes_K8(E):
    L, T0 = 32, 32 + 24
    eng = E.DropoutEngine(E.VICUNA_7B, family=FAMILY_IBLIP, max_seq=320, max_visual=L, kv_format="fp16")
    eng.load_synthetic(4, 0.02)
    emb = (torch.randn(T0, 4096, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    lib = eng.lib
    runs = {}
    for name, (spec, graph) in {"two-sweep eager": (0, 0), "speculative eager": (1, 0), "speculative graph": (1, 1)}.items():
        lib.dd_set_tuning(14, spec)
        lib.dd_set_tuning(8, graph)
        try:
            eng.prefill(emb, 0, L)
            recs = []
            for s in range(12):
                eng.decode_step(PROBS8)
                st = eng.last_step()
                recs.append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], eng.logits().copy(), eng.base_logits().copy()))
            runs[name] = (recs, eng.tokens(), eng.kv_sums().copy())
        finally:
            lib.dd_set_tuning(14, 1)
            lib.dd_set_tuning(8, 1)
    ref = runs["two-sweep eager"]
    for name, got in runs.items():
        assert got[1] == ref[1], name
        for s, (a, b) in enumerate(zip(got[0], ref[0])):
            np.testing.assert_array_equal(a[0], b[0], err_msg=f"{name} step {s}")
            assert a[1] == b[1] and a[2] == b[2]
            np.testing.assert_array_equal(a[3], b[3])
            np.testing.assert_array_equal(a[4], b[4])
        np.testing.assert_array_equal(got[2], ref[2])
    # the deterministic quantile masks (instructblip.py:447-460): member k drops the top mprob_k share of the 32 tokens, nested,
    # minus the kept tokens; the NEXT step's un-masked pass sees the last member's zeros (Q2) — a step with the leak differs from
    # the same step without it
    drop = ref[0][-1][0]
    counts = drop.sum(1)
    assert all(counts[k] <= counts[k + 1] for k in range(7)) and 1 <= counts[0] <= 5 and 20 <= counts[7] <= 27
    # Q2 (instructblip.py:111-122): the next step's un-masked pass sees the last member's zeros.  With them, step 2's base logits
    # differ from a stock-greedy (K = 0, nothing leaks) second step on the same prefix; both are finite
    eng.prefill(emb, 0, L)
    eng.decode_step(PROBS8)
    tok1 = eng.tokens()[-1]
    eng.decode_step(PROBS8)
    leaked = eng.base_logits().copy()
    eng.prefill(emb, 0, L)
    eng.decode_step(PROBS8, dropout=False)
    if eng.tokens()[-1] == tok1:                               # same input token: the only difference is the leaked mask
        eng.decode_step(PROBS8, dropout=False)
        assert np.isfinite(leaked).all() and np.abs(leaked - eng.base_logits()).max() > 0
    # four lanes (32 member rows per sweep) bit-identical to this solo run
    lanes = [eng] + [E.DropoutEngine(E.VICUNA_7B, family=FAMILY_IBLIP, max_seq=320, max_visual=L, kv_format="fp16", share_weights_with=eng)
                     for _ in range(3)]
    embs = [emb] + [(torch.randn(T0 + 3 * i, 4096, generator=torch.Generator().manual_seed(10 + i)) * 0.5).cuda() for i in range(1, 4)]
    for e, x in zip(lanes, embs):
        e.prefill(x, 0, L)
    grp = E.EngineGroup(lanes)
    for s in range(12):
        grp.decode_step(PROBS8)
        np.testing.assert_array_equal(eng.logits(), ref[0][s][3], err_msg=f"lane 0 in a group of 4, step {s}")
    assert eng.tokens() == ref[1]
    grp.decode_step(PROBS8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        grp.decode_step(PROBS8)
    torch.cuda.synchronize()
    t_grp = (time.perf_counter() - t0) / 10 * 1e3
    eng.prefill(emb, 0, L)
    eng.decode_step(PROBS8)
    eng.decode_step(PROBS8)
    t_solo = _timed_steps(eng, PROBS8, 20)
    print(f"\n[config 4] InstructBLIP-Vicuna-7B shapes, L=32, K=8, fp16 KV: {t_solo:.2f} ms per ensemble step (one sequence, speculative "
          f"single-sweep steps where the keep set allows), {t_grp:.2f} ms per group step of 4 lanes = {t_grp / 4:.2f} ms per image-token")
    for e in reversed(lanes):
        e.close()


def test_config5_next_mistral7b_shapes_fp8_K8_across_attention_buckets(E):
    L, T0 = 2928, 3060
    eng = E.DropoutEngine(E.MISTRAL_7B, family=FAMILY_NEXT, max_seq=3500, max_visual=L, weight_format="fp8", kv_format="fp16", seed=506)
    eng.load_synthetic(5, 0.02)
    assert eng.device_bytes < 16e9                                       # 7.1 GB of fp8 matrices + cache + scratch
    emb = (torch.randn(T0, 4096, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
    lib = eng.lib
    n_new = 290                                                          # T crosses 3072 and 3328: two launch-shape changes
    t0 = time.perf_counter()
    eng.prefill(emb, 16, L)
    eng.torch_stream.synchronize()
    t_prefill = (time.perf_counter() - t0) * 1e3
    u = eng.vision_uncert_dict()
    assert np.isfinite(u["epis_uncert_per_token"]).all() and (u["epis_uncert_per_token"] > -1e-6).all()
    a = eng.generate(n_new, mprobs=PROBS8)
    assert len(a) == n_new and eng.T() == T0 + n_new - 1
    st = eng.last_step()
    assert not (st["drop"] & st["keep"][None]).any()                     # kept tokens are never dropped (llavanext.py:805-807)
    nd = st["masked_numbers"]
    # p in [0.1, mprob] per token (llavanext.py:779-808), masks reset per member (:546): mprob = 0.1 drops ~10 %, 0.8 more
    assert 0.05 * L < nd[0] < 0.15 * L and nd[0] < nd[7] < 0.8 * L and all(nd[k] <= nd[k + 1] + 0.05 * L for k in range(7))
    sums = eng.kv_sums().copy()
    # replay: eager launches, two-sweep steps — the same tokens, cache and rng stream as graph replays of speculative steps
    lib.dd_set_tuning(8, 0)
    lib.dd_set_tuning(14, 0)
    try:
        eng.rng.manual_seed(506)
        eng.prefill(emb, 16, L)
        b = eng.generate(n_new, mprobs=PROBS8)
    finally:
        lib.dd_set_tuning(8, 1)
        lib.dd_set_tuning(14, 1)
    assert a == b
    np.testing.assert_array_equal(eng.kv_sums(), sums)
    # two lanes (fp8 grouped sweeps) bit-identical to the solo run
    lane = E.DropoutEngine(E.MISTRAL_7B, family=FAMILY_NEXT, max_seq=3500, max_visual=L, weight_format="fp8", kv_format="fp16", seed=506,
                           share_weights_with=eng)
    emb2 = (torch.randn(T0 - 100, 4096, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()
    eng.rng.manual_seed(506)
    eng.prefill(emb, 16, L)
    lane.prefill(emb2, 16, L - 100)
    got = E.EngineGroup([eng, lane]).generate(40, mprobs=PROBS8)
    assert got[0] == a[:40]
    eng.rng.manual_seed(506)
    eng.prefill(emb, 16, L)
    eng.decode_step(PROBS8)
    eng.decode_step(PROBS8)
    t_step = _timed_steps(eng, PROBS8, 30)
    print(f"\n[config 5] LLaVA-NeXT-Mistral-7B shapes, L=2928, T~3060, GQA 4, K=8, fp8 weights, fp16 KV: prefill {t_prefill:.0f} ms, "
          f"{t_step:.2f} ms per ensemble step (one sequence)")
    lane.close()
    eng.close()


def test_config5_weight_format_reaches_the_drop_in_class(E):
    """settings['weight_format'] = 'fp8' before from_hf_model / from_pretrained builds an fp8 engine behind
    CustomLlavaNextForConditionalGeneration (models/llavanext.py:505-514 is what it replaces); tokens equal the oracle's
    on the dequantised weights."""
    from transformers import CLIPVisionConfig, LlavaNextConfig, LlavaNextForConditionalGeneration, MistralConfig
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd.llavanext import CustomLlavaNextForConditionalGeneration
    from dropoutdecoding_amd.vlm import lm_state_dict_from_hf
    torch.manual_seed(1)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                          image_size=56, patch_size=14, projection_dim=16)
    tc = MistralConfig(vocab_size=512, hidden_size=512, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=128, max_position_embeddings=1024, sliding_window=None,
                       tie_word_embeddings=False)
    cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-2,
                          vision_feature_select_strategy="default", image_grid_pinpoints=[[56, 112], [112, 56], [112, 112]])
    hf = LlavaNextForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.copy_(p.to(torch.bfloat16).float())
        for n, p in hf.named_parameters():
            if "language_model" in n or "lm_head" in n:
                p.mul_(2.0)
    sd = {k: v.detach().float().cpu() for k, v in lm_state_dict_from_hf(hf).items()}
    deq = {}
    for k, v in sd.items():                       # what the fp8 engine computes with: per-row e4m3fn quantisation of every matrix
        if v.dim() == 2 and "embed_tokens" not in k:
            q, s = E.quantize_fp8(v)
            deq[k] = E.dequantize_fp8(q, s)
        else:
            deq[k] = bf16_round(v)
    old = dict(ddc.settings)
    try:
        ddc.settings["voting_numbers"] = [0.1, 0.3, 0.5, 0.7]
        ddc.settings["use_random"] = [False]
        ddc.settings["weight_format"] = "fp8"
        ddc._module_imported(506)
        m = CustomLlavaNextForConditionalGeneration.from_hf_model(hf, max_new_tokens=16, max_visual=256)
        assert m.engine.weight_format == "fp8"
        pv = torch.randn(1, 5, 3, 56, 56, generator=torch.Generator().manual_seed(3))
        sizes = torch.tensor([[100, 100]])
        vis = m._visual_embeds(pixel_values=pv, image_sizes=sizes)
        Lv = vis.shape[0]
        ids = torch.tensor([[1, 17] + [511] * Lv + [45, 6, 7, 99]])
        out = m.generate(input_ids=ids, pixel_values=pv, image_sizes=sizes, max_new_tokens=6, eos_token_id=[])
        emb, start = m._merge(ids.cuda(), vis)
        rp = getattr(tc, "rope_parameters", None) or {}
        rc = RefCfg(512, 512, 512, 2, 4, 2, 128, tc.rms_norm_eps, float(rp.get("rope_theta", getattr(tc, "rope_theta", 10000.0))))
        want = RefDecoder(FAMILY_NEXT, rc, deq, [0.1, 0.3, 0.5, 0.7], seed=506).generate(emb.cpu(), start, Lv, 6)
        assert out[0, ids.shape[1]:].tolist() == want
    finally:
        ddc.settings.clear()
        ddc.settings.update(old)


def test_llava7b_speculation_policies_on_keep_sets_that_are_never_empty(E):
    """LLaVA-1.5-7B shapes, K = 8, with 64 lm_head rows scaled up so that every step's keep set (models/llava.py:443-482)
    holds tens of visual tokens — what a trained checkpoint does where the image shows what is being said, and what random
    weights never do.  There the speculative step's check fails at (nearly) every step; the three policies must produce the
    same tokens and cache, and the adaptive one must cost about what the plain two-sweep step costs (printed)."""
    eng = E.DropoutEngine(E.LLAVA15_7B, family=FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16")
    eng.load_synthetic(0, 0.02)
    g = torch.Generator(device="cuda").manual_seed(3)
    W = torch.randn(32064, 4096, device="cuda", generator=g) * 0.02
    W[1000:1064] *= 8.0
    eng._load(E.T_LM_HEAD, 0, W)
    del W
    emb = (torch.randn(608, 4096, generator=torch.Generator().manual_seed(1)) * 0.5).cuda()
    n_new, out = 97, {}
    for mode in ("never", "always", "adaptive"):
        eng.set_speculation(mode)
        eng.rng.manual_seed(24)
        eng.prefill(emb, 5, 576)
        eng.generate(4, mprobs=PROBS8)                       # graphs captured, clocks up
        eng.spec_stats(reset=True)
        eng.torch_stream.synchronize()
        t0 = time.perf_counter()
        toks = eng.generate(n_new, mprobs=PROBS8)
        eng.torch_stream.synchronize()
        ms = (time.perf_counter() - t0) / (n_new - 4) * 1e3
        st = eng.last_step()
        out[mode] = (toks, eng.kv_sums().copy(), ms, eng.spec_stats(), int(st["keep"].sum()))
    eng.set_speculation("default")
    for mode in ("always", "adaptive"):
        assert out[mode][0] == out["never"][0], mode
        np.testing.assert_array_equal(out[mode][1], out["never"][1])
    assert out["never"][4] >= 5                              # the last step kept several visual tokens
    sa, sd = out["always"][3], out["adaptive"][3]
    assert sa["hit_rate"] < 0.2 and sd["plain"] > 0.5 * (n_new - 4) and sd["switched_off"] >= 1
    assert out["adaptive"][2] < out["always"][2]             # falling back pays on such a checkpoint
    assert out["adaptive"][2] < 1.10 * out["never"][2]       # and costs at most the first misses and the probes on top of the plain step
    print(f"\n[keep sets never empty, LLaVA-1.5-7B shapes, K=8, fp16 KV, T~660] ms per ensemble step: plain two-sweep {out['never'][2]:.2f}, "
          f"speculating always {out['always'][2]:.2f} (hit rate {sa['hit_rate']:.2f}), adaptive {out['adaptive'][2]:.2f} "
          f"({sd['plain']} plain + {sd['held'] + sd['rerun']} probe steps, {sd['rerun']} re-runs)")
    eng.close()


def test_configs_4_and_5_through_their_drop_in_classes_at_full_size():
    """`from_synthetic` of the InstructBLIP and LLaVA-NeXT wrappers (what `bench.py --config 4 / 5` runs): the real front-end shapes on
    own kernels — EVA ViT-g/14 + Q-Former -> 32 visual tokens; CLIP-L/14-336 over the 5 anyres tiles of a 672 x 672 image -> 2928 visual
    tokens — feeding the drop-in `generate()`; return layouts, span bookkeeping, and a lane decoding the same inputs to the same ids."""
    from dropoutdecoding_amd import build, config as ddc
    build.build()
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration
    from dropoutdecoding_amd.llavanext import CustomLlavaNextForConditionalGeneration
    from dropoutdecoding_amd.vlm import generate_group
    saved = dict(ddc.settings)
    try:
        ddc.settings["voting_numbers"] = PROBS8
        g = torch.Generator().manual_seed(0)
        m = CustomInstructBlipForConditionalGeneration.from_synthetic(max_new_tokens=16)
        assert m.tower_hip is not None and m.qformer_hip is not None
        kw = dict(input_ids=torch.randint(3, 31000, (1, 12), generator=g), pixel_values=torch.randn(1, 3, 224, 224, generator=g),
                  qformer_input_ids=torch.randint(1000, 30000, (1, 9), generator=g), qformer_attention_mask=torch.ones(1, 9, dtype=torch.long))
        out = m.generate(**kw, max_new_tokens=5, eos_token_id=[])
        assert out.shape == (1, 6) and int(out[0, 0]) == 2 and m.start_image_pos == [0] and m.end_image_pos == [31] and m.start_generation_pos == 44
        lanes = [m.spawn_lane(), m.spawn_lane()]
        kw2 = dict(kw, pixel_values=torch.randn(1, 3, 224, 224, generator=g))
        outs = generate_group(lanes, [kw, kw2], max_new_tokens=5, eos_token_id=[])          # the tower batched over the two images
        assert outs[0].tolist() == out.tolist()
        assert outs[1].tolist() == m.spawn_lane().generate(**kw2, max_new_tokens=5, eos_token_id=[]).tolist()
        del m, lanes
        torch.cuda.empty_cache()
        n = CustomLlavaNextForConditionalGeneration.from_synthetic(max_new_tokens=16)
        assert n.tower_hip is not None and n.engine.weight_format == "fp8"
        ids = torch.randint(3, 31000, (1, 20), generator=g)
        ids[0, 4] = 32000
        out = n.generate(input_ids=ids, pixel_values=torch.randn(1, 5, 3, 336, 336, generator=g), image_sizes=torch.tensor([[672, 672]]),
                         max_new_tokens=4, eos_token_id=[])
        assert out.shape == (1, 24) and out[0, :20].tolist() == ids[0].tolist()
        assert n.start_image_pos == [4] and n.end_image_pos == [4 + 2928 - 1] and n.start_generation_pos == 19 + 2928
        assert n.image_features[1].shape == (1, 2928, 10)
        # round 6: the anyres tiles of several images in ONE tower call (three 5-tile images = 15 of the 16 tiles a call takes, the fourth
        # in a call of its own): every image's visual tokens bit for bit those of its own call, and the ids of generate_group with them
        kws = [dict(input_ids=ids, pixel_values=torch.randn(1, 5, 3, 336, 336, generator=g), image_sizes=torch.tensor([[672, 672]])) for _ in range(4)]
        vis = n._visual_embeds_batch([{k: v for k, v in kw.items() if k != "input_ids"} for kw in kws])
        for kw, v in zip(kws, vis):
            one = n._visual_embeds(pixel_values=kw["pixel_values"], image_sizes=kw["image_sizes"])
            assert v.shape == one.shape == (2928, 4096)
            assert torch.equal(v, one)
        lanes = [n.spawn_lane(), n.spawn_lane()]
        outs = generate_group(lanes, kws[:2], max_new_tokens=3, eos_token_id=[])
        for kw, o in zip(kws[:2], outs):
            assert o.tolist() == n.spawn_lane().generate(**kw, max_new_tokens=3, eos_token_id=[]).tolist()
    finally:
        ddc.settings.clear()
        ddc.settings.update(saved)
