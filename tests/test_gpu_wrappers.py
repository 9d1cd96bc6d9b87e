"""generate() surface of the drop-in wrappers on tiny random HF models, and the K-shard phase kernels, on the GPU."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, bf16_round, random_weights


@pytest.fixture(scope="module")
def built():
    from dropoutdecoding_amd import build
    build.build()
    return True


def _ref_weights_from_engine_sd(sd):
    return {k: bf16_round(v.detach().float().cpu()) for k, v in sd.items()}


def test_llava_wrapper_generate_matches_oracle(built):
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration
    from dropoutdecoding_amd.vlm import lm_state_dict_from_hf
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                          image_size=56, patch_size=14, projection_dim=16)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-2,
                      vision_feature_select_strategy="default")
    hf = LlavaForConditionalGeneration(cfg).eval()
    for p_ in hf.parameters():                               # bf16-valued weights on both sides (vision tower too)
        p_.copy_(p_.to(torch.bfloat16).float())
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "language_model" in n or "lm_head" in n:
                p.mul_(2.5)
    sd = _ref_weights_from_engine_sd(lm_state_dict_from_hf(hf))
    ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
    ddc._module_imported(24)
    m = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
    ids1 = torch.tensor([[1, 17, 511, 45, 6, 7, 99]])                      # one placeholder (transformers 4.44 processors)
    idsL = torch.tensor([[1, 17] + [511] * 16 + [45, 6, 7, 99]])           # run of L placeholders (5.x processors)
    pv = torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    out1 = m.generate(input_ids=ids1, attention_mask=torch.ones_like(ids1), pixel_values=pv, max_new_tokens=8,
                      num_beams=1, pad_token_id=0, eos_token_id=[])
    assert out1.shape == (1, ids1.shape[1] + 8) and out1[0, :7].tolist() == ids1[0].tolist()
    # oracle on the same merged embeddings
    assert m.tower_hip is not None                           # CLIP tower + projector on own kernels
    vis = m._visual_embeds(pixel_values=pv)
    hs = m.vision_tower(pv.cuda(), output_hidden_states=True).hidden_states[-2][:, 1:]
    vis_torch = m.multi_modal_projector(hs)[0]                # the third-party path the reference uses (llava.py:233-246)
    assert float((vis - vis_torch).abs().max()) <= 1e-3 * float(vis_torch.abs().max())
    emb, start = m._merge(ids1.cuda(), vis)
    assert start == 2 and emb.shape[0] == 6 + 16
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, tc.rms_norm_eps, 10000.0)
    ref = RefDecoder(FAMILY_LLAVA, rc, sd, [0.3, 0.5, 0.7], seed=24)
    want = ref.generate(emb.cpu(), start, 16, 8)
    assert out1[0, 7:].tolist() == want
    assert m.start_image_pos == [2] and m.end_image_pos == [17] and m.start_generation_pos == 22
    np.testing.assert_allclose(m.vision_uncert_dict["epis_uncert_per_token"].cpu().numpy()[0], ref.epi.numpy(), rtol=5e-3, atol=1e-6)
    assert m.image_features[1].shape == (1, 16, 5)
    # second image on the same model object: the rng stream continues (never re-seeded per image)
    outL = m.generate(input_ids=idsL, pixel_values=pv, max_new_tokens=8, eos_token_id=[])
    want2 = ref.generate(emb.cpu(), start, 16, 8)
    assert outL[0, idsL.shape[1]:].tolist() == want2
    with pytest.raises(ValueError):
        m.generate(input_ids=torch.tensor([[1, 2, 3]]), pixel_values=pv, max_new_tokens=2)      # no image token
    with pytest.raises(ValueError):
        m.generate(input_ids=torch.tensor([[1, 511, 511, 3]]), pixel_values=pv, max_new_tokens=2)  # wrong count
    # --original: stock greedy
    m.original = True
    og = m.generate(input_ids=ids1, pixel_values=pv, max_new_tokens=6, eos_token_id=[])
    ref0 = RefDecoder(FAMILY_LLAVA, rc, sd, [], dropout=False)
    assert og[0, 7:].tolist() == ref0.generate(emb.cpu(), start, 16, 6)
    # EOS stops generation
    m.original = False
    full = RefDecoder(FAMILY_LLAVA, rc, sd, [0.3, 0.5, 0.7], seed=5).generate(emb.cpu(), start, 16, 8)
    eos = full[4]
    w3 = full[:full.index(eos) + 1]
    m.engine.rng.manual_seed(5)
    o3 = m.generate(input_ids=ids1, pixel_values=pv, max_new_tokens=8, eos_token_id=eos)
    assert o3[0, 7:].tolist() == w3 and w3[-1] == eos and len(w3) <= 5
    # several questions about one image (POPE): with settings['reuse_image_prefix'] the second prompt keeps the image
    # prefix of the first and prefills only its own text — same ids out as without it
    qs = [torch.tensor([[1, 17, 511, 45, 6, 7, 99]]), torch.tensor([[1, 17, 511, 88, 3]]), torch.tensor([[1, 17, 511, 9, 9, 9, 12, 40]])]
    m.engine.rng.manual_seed(11)
    plain = [m.generate(input_ids=q, pixel_values=pv, max_new_tokens=4, eos_token_id=[]) for q in qs]
    ddc.settings["reuse_image_prefix"] = True
    try:
        m.engine.rng.manual_seed(11)
        calls = {"n": 0}
        orig_vis = m._visual_embeds
        m._visual_embeds = lambda **kw: (calls.__setitem__("n", calls["n"] + 1), orig_vis(**kw))[1]
        reused = [m.generate(input_ids=q, pixel_values=pv, max_new_tokens=4, eos_token_id=[]) for q in qs]
        other = m.generate(input_ids=qs[1], pixel_values=pv + 0.5, max_new_tokens=4, eos_token_id=[])     # another image
        m._visual_embeds = orig_vis
    finally:
        ddc.settings.pop("reuse_image_prefix")
    assert [r.tolist() for r in reused] == [p_.tolist() for p_ in plain]
    assert calls["n"] == 1       # vision tower: never for the three questions (the prefix of the last plain call is still
                                 # cached: same image), once for the new image
    assert other.shape == (1, 5 + 4) and m.start_generation_pos == 2 + 16 + 2
    # POPE-style one-token answer with the ensemble on the first token (the `# if True:` toggle, llava.py:336-337)
    ddc.settings["first_step_ensemble"] = True
    try:
        m.engine.rng.manual_seed(9)
        o4 = m.generate(input_ids=ids1, pixel_values=pv, max_new_tokens=3, eos_token_id=[])
    finally:
        ddc.settings.pop("first_step_ensemble")
    ref4 = RefDecoder(FAMILY_LLAVA, rc, sd, [0.3, 0.5, 0.7], seed=9, first_step_ensemble=True)
    assert o4[0, 7:].tolist() == ref4.generate(emb.cpu(), start, 16, 3)
    assert m.engine.last_step()["winner"] == ref4.records[-1].winner
    # settings['rng_stream'] = 'gpu': the same model drawing from torch's GPU generator (the reference run on a GPU), lanes too
    ddc.settings["rng_stream"] = "gpu"
    try:
        mg = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
    finally:
        ddc.settings.pop("rng_stream")
    assert mg.engine.rng_stream == "gpu" and mg.spawn_lane().engine.rng_stream == "gpu"
    og = mg.generate(input_ids=ids1, pixel_values=pv, max_new_tokens=8, eos_token_id=[])
    refg = RefDecoder(FAMILY_LLAVA, rc, sd, [0.3, 0.5, 0.7], seed=24, rng_stream="gpu")
    assert og[0, 7:].tolist() == refg.generate(emb.cpu(), start, 16, 8) != want


def test_instructblip_merge_and_output_format(built):
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration as IB
    from dropoutdecoding_amd import lm
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-6, 10000.0)
    w = random_weights(rc, 5, 0.05)
    eng = lm.DropoutEngine(lm.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-6, 10000.0), family=lm.FAMILY_IBLIP, max_seq=128, max_visual=32)
    eng.load_state_dict(w)
    m = IB(eng, w["model.embed_tokens.weight"].cuda().to(torch.bfloat16), hf_front=None, eos_token_id=None, config=None)
    vis = torch.randn(32, 256, generator=torch.Generator().manual_seed(1)).cuda() * 0.7
    m._visual_embeds = lambda **kw: vis
    ids = torch.tensor([[1, 17, 45, 6, 7, 99]])
    out = m.generate(input_ids=ids, pixel_values=torch.zeros(1), max_new_tokens=6, eos_token_id=[])
    assert out.shape == (1, 7) and int(out[0, 0]) == 2                    # BOS(2) + new ids only (instructblip.py:686-695)
    emb = torch.cat([vis.cpu(), w["model.embed_tokens.weight"][ids[0]]], 0)
    ref = RefDecoder(FAMILY_IBLIP, rc, w, [0.3, 0.5, 0.7])
    assert out[0, 1:].tolist() == ref.generate(emb, 0, 32, 6)
    assert m.start_image_pos == [0] and m.end_image_pos == [31]


@pytest.mark.parametrize("split", [(4, 8), (1, 8), (3, 5)])
def test_kshard_phase_kernels_two_engines(built, split):
    """Two engines on one GPU play two ranks: members [0,s) and [s,K); records are summed by hand (what all-reduce does)."""
    from dropoutdecoding_amd import lm
    s_, K = split
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 11, 0.05)
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]
    emb = torch.randn(30, 256, generator=torch.Generator().manual_seed(2)).cuda()
    engs = []
    for _ in range(3):
        e = lm.DropoutEngine(lm.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=lm.FAMILY_LLAVA, max_seq=128,
                             max_visual=32, seed=42)
        e.load_state_dict(w)
        e.prefill(emb, 3, 20)
        engs.append(e)
    a, b, single = engs
    bufs = [e.new_xchg_buffers() for e in (a, b)]
    for step in range(6):
        single.decode_step(probs)
        for e in (a, b):
            e.step_base(probs)
        a.step_members(0, s_)
        b.step_members(s_, K)
        a.export_ids(0, s_, bufs[0][0])
        b.export_ids(s_, K, bufs[1][0])
        torch.cuda.synchronize()                       # the engines run on their own streams; this sum plays all-reduce
        ids = bufs[0][0] + bufs[1][0]
        torch.cuda.synchronize()
        for e in (a, b):
            e.import_ids(ids)
        a.export_winner(0, s_, bufs[0][1])
        b.export_winner(s_, K, bufs[1][1])
        torch.cuda.synchronize()
        rec = bufs[0][1] + bufs[1][1]
        torch.cuda.synchronize()
        for e in (a, b):
            e.import_winner(rec)
            e.step_commit()
        st = single.last_step()
        for e in (a, b):
            assert e.tokens() == single.tokens(), step
            np.testing.assert_array_equal(e.logits(), single.logits())
            np.testing.assert_allclose(e.kv_sums(), single.kv_sums(), rtol=0, atol=0)
            assert e.last_step()["winner"] == st["winner"]


def test_llavanext_wrapper_with_hf_tiny_model(built):
    """anyres front-end through HF's own get_image_features (tile split, CLIP, projector, unpad + newline packing)."""
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
    for p in hf.parameters():
        p.copy_(p.to(torch.bfloat16).float())
    for n, p in hf.named_parameters():
        if "language_model" in n or "lm_head" in n:
            p.mul_(2.0)
    sd = _ref_weights_from_engine_sd(lm_state_dict_from_hf(hf))
    ddc.settings["voting_numbers"] = [0.1, 0.3, 0.5, 0.7]
    ddc.settings["use_random"] = [False]
    ddc._module_imported(506)
    m = CustomLlavaNextForConditionalGeneration.from_hf_model(hf, max_new_tokens=16, max_visual=256)
    pv = torch.randn(1, 5, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    sizes = torch.tensor([[100, 100]])
    assert m.tower_hip is not None
    vis = m._visual_embeds(pixel_values=pv, image_sizes=sizes)
    hf_out = m._hf.get_image_features(pv.cuda(), sizes.cuda(), vision_feature_layer=-2, vision_feature_select_strategy="default")
    hf_feats = hf_out.pooler_output if hasattr(hf_out, "pooler_output") else hf_out
    hf_feats = torch.cat(list(hf_feats), 0) if isinstance(hf_feats, (list, tuple)) else hf_feats
    assert vis.shape == hf_feats.shape
    assert float((vis - hf_feats).abs().max()) <= 1e-3 * float(hf_feats.abs().max())    # own tower vs the HF modules
    L = vis.shape[0]
    assert L == 88                                                     # 100x100 image -> 88 visual tokens
    ids = torch.tensor([[1, 17] + [511] * L + [45, 6, 7, 99]])
    out = m.generate(input_ids=ids, pixel_values=pv, image_sizes=sizes, max_new_tokens=6, eos_token_id=[])
    emb, start = m._merge(ids.cuda(), vis)
    rc = RefCfg(512, 512, 512, 2, 4, 2, 128, tc.rms_norm_eps, float(LMtheta(tc)))
    ref = RefDecoder(FAMILY_NEXT, rc, sd, [0.1, 0.3, 0.5, 0.7], seed=506)
    assert out[0, ids.shape[1]:].tolist() == ref.generate(emb.cpu(), start, L, 6)
    assert m.image_features[1].shape == (1, L, 10)                     # top-10 ids (llavanext.py:652)
    # three NeXT images as lanes (the tower sees each image's five tiles as one batch; prompts prefilled per lane — 3 x 128 rows
    # is under the batched prefill's threshold): each lane == its own generate()
    from dropoutdecoding_amd.vlm import generate_group
    lanes = [m.spawn_lane() for _ in range(3)]           # fresh lanes: each starts its rng stream like a fresh process, as `solo` below
    pvs = [torch.randn(1, 5, 3, 56, 56, generator=torch.Generator().manual_seed(30 + i)) for i in range(3)]
    ins = [dict(input_ids=ids, pixel_values=p_, image_sizes=sizes) for p_ in pvs]
    outs = generate_group(lanes, ins, max_new_tokens=6, eos_token_id=[])
    for i, (p_, o) in enumerate(zip(pvs, outs)):
        solo = m.spawn_lane()
        assert o.tolist() == solo.generate(input_ids=ids, pixel_values=p_, image_sizes=sizes, max_new_tokens=6, eos_token_id=[]).tolist(), f"lane {i}"


def LMtheta(tc):
    rp = getattr(tc, "rope_parameters", None) or {}
    return rp.get("rope_theta", getattr(tc, "rope_theta", 10000.0))


def test_instructblip_wrapper_with_hf_tiny_model(built):
    """ViT + Q-Former + language_projection through the HF modules (reference models/instructblip.py:607-633), then
    the engine with InstructBLIP semantics; returned ids = BOS(2) ‖ new ids."""
    from transformers import (InstructBlipConfig, InstructBlipForConditionalGeneration, InstructBlipQFormerConfig,
                              InstructBlipVisionConfig, LlamaConfig)
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration
    from dropoutdecoding_amd.vlm import lm_state_dict_from_hf
    torch.manual_seed(2)
    vc = InstructBlipVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                  image_size=28, patch_size=14)
    qc = InstructBlipQFormerConfig(vocab_size=100, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                   intermediate_size=64, encoder_hidden_size=32, cross_attention_frequency=1)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False)
    cfg = InstructBlipConfig(vision_config=vc.to_dict(), qformer_config=qc.to_dict(), text_config=tc.to_dict(),
                             num_query_tokens=32)
    hf = InstructBlipForConditionalGeneration(cfg).eval()
    for n, p in hf.named_parameters():
        if "language_model" in n:
            p.mul_(2.5)
        if "language_projection" in n:
            p.mul_(8.0)
    # HF initialises the Q-Former queries to zero: 32 identical visual tokens would make epi pure rounding noise
    _in = getattr(hf, "model", hf)
    (_in if hasattr(_in, "query_tokens") else hf).query_tokens.normal_(0, 1.0, generator=torch.Generator().manual_seed(9))
    sd = _ref_weights_from_engine_sd(lm_state_dict_from_hf(hf))
    ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
    ddc._module_imported(5217)
    m = CustomInstructBlipForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
    pv = torch.randn(1, 3, 28, 28, generator=torch.Generator().manual_seed(3))
    qids = torch.tensor([[3, 9, 27, 4]])
    ids = torch.tensor([[1, 17, 45, 6, 7, 99]])
    out = m.generate(pixel_values=pv, qformer_input_ids=qids, qformer_attention_mask=torch.ones_like(qids), input_ids=ids,
                     attention_mask=torch.ones_like(ids), max_new_tokens=7, eos_token_id=[])
    assert out.shape == (1, 8) and int(out[0, 0]) == 2
    vis = m._visual_embeds(pixel_values=pv, qformer_input_ids=qids, qformer_attention_mask=torch.ones_like(qids))
    assert vis.shape == (32, 256)
    emb, start = m._merge(ids.cuda(), vis)
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, tc.rms_norm_eps, 10000.0)
    ref = RefDecoder(FAMILY_IBLIP, rc, sd, [0.3, 0.5, 0.7])
    assert out[0, 1:].tolist() == ref.generate(emb.cpu(), 0, 32, 7)
    assert m.start_image_pos == [0] and m.end_image_pos == [31] and m.start_generation_pos == 38


def test_instructblip_front_end_on_own_kernels(built):
    """A vision config with heads of 88 (EVA ViT-g's shape) and a Q-Former with heads of 64: the wrapper must route the whole
    visual front-end (tower -> Q-Former -> language_projection, reference models/instructblip.py:607-633) through the library's
    kernels, and the result must match the HF modules it replaces."""
    from transformers import (InstructBlipConfig, InstructBlipForConditionalGeneration, InstructBlipQFormerConfig,
                              InstructBlipVisionConfig, LlamaConfig)
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration
    torch.manual_seed(4)
    vc = InstructBlipVisionConfig(hidden_size=704, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8,
                                  image_size=56, patch_size=14)
    qc = InstructBlipQFormerConfig(vocab_size=100, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                   intermediate_size=256, encoder_hidden_size=704, cross_attention_frequency=2)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False)
    cfg = InstructBlipConfig(vision_config=vc.to_dict(), qformer_config=qc.to_dict(), text_config=tc.to_dict(),
                             num_query_tokens=32)
    hf = InstructBlipForConditionalGeneration(cfg).eval()
    _in = getattr(hf, "model", hf)
    (_in if hasattr(_in, "query_tokens") else hf).query_tokens.normal_(0, 1.0, generator=torch.Generator().manual_seed(9))
    for n, p in hf.named_parameters():
        if "language_model" not in n:
            p.copy_((p * (2.0 if p.dim() > 1 else 1.0)).to(torch.bfloat16).float())       # bf16-valued on both sides
    ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
    ddc._module_imported(5217)
    m = CustomInstructBlipForConditionalGeneration.from_hf_model(hf, max_new_tokens=8)
    assert m.tower_hip is not None and m.qformer_hip is not None
    pv = torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    qids = torch.tensor([[3, 9, 27, 4, 0, 0]])
    qmask = torch.tensor([[1, 1, 1, 1, 0, 0]])
    own = m._visual_embeds(pixel_values=pv, qformer_input_ids=qids, qformer_attention_mask=qmask).float()
    tower, qf = m.tower_hip, m.qformer_hip
    m.tower_hip = m.qformer_hip = None
    want = m._visual_embeds(pixel_values=pv, qformer_input_ids=qids, qformer_attention_mask=qmask).float()
    m.tower_hip, m.qformer_hip = tower, qf
    assert own.shape == want.shape == (32, 256)
    err = float((own - want).abs().max() / want.abs().max())
    print(f"\n[InstructBLIP front-end] own kernels vs HF modules: {err:.2e}")
    assert err < 2e-3
    ids = torch.tensor([[1, 17, 45, 6, 7, 99]])
    out = m.generate(pixel_values=pv, qformer_input_ids=qids, qformer_attention_mask=qmask, input_ids=ids,
                     attention_mask=torch.ones_like(ids), max_new_tokens=5, eos_token_id=[])
    assert out.shape == (1, 6) and int(out[0, 0]) == 2
    # two InstructBLIP images as lanes (leaked mask bits, hidden-state vote per lane) == their own generate()
    from dropoutdecoding_amd.vlm import generate_group
    lanes = [m.spawn_lane() for _ in range(2)]
    pvs = [torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(60 + i)) for i in range(2)]
    ins = [dict(pixel_values=p_, qformer_input_ids=qids, qformer_attention_mask=qmask, input_ids=ids) for p_ in pvs]
    outs = generate_group(lanes, ins, max_new_tokens=5, eos_token_id=[])
    for i, (kw, o) in enumerate(zip(ins, outs)):
        assert o.tolist() == m.spawn_lane().generate(**kw, max_new_tokens=5, eos_token_id=[]).tolist(), f"lane {i}"


def test_generate_group_three_images_at_once(built):
    """spawn_lane() + generate_group(): three images decoded together over one set of weights; each equals its own solo
    generate() and the oracle started from a fresh rng stream (one reference process per lane)."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration
    from dropoutdecoding_amd.vlm import generate_group, lm_state_dict_from_hf
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                          image_size=56, patch_size=14, projection_dim=16)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-2,
                      vision_feature_select_strategy="default")
    hf = LlavaForConditionalGeneration(cfg).eval()
    for p_ in hf.parameters():
        p_.copy_(p_.to(torch.bfloat16).float())
    for n, p in hf.named_parameters():
        if "language_model" in n or "lm_head" in n:
            p.mul_(2.5)
    sd = _ref_weights_from_engine_sd(lm_state_dict_from_hf(hf))
    ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
    ddc._module_imported(24)
    m0 = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
    lanes = [m0, m0.spawn_lane(), m0.spawn_lane()]
    assert lanes[1].engine is not m0.engine and lanes[1].engine.weight_owner is m0.engine
    prompts = [torch.tensor([[1, 17, 511, 45, 6, 7, 99]]), torch.tensor([[1, 511, 8, 9]]), torch.tensor([[1, 3, 4, 5, 511, 77]])]
    pvs = [torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    outs = generate_group(lanes, [dict(input_ids=p, pixel_values=v, attention_mask=torch.ones_like(p)) for p, v in zip(prompts, pvs)],
                          max_new_tokens=7, eos_token_id=[])
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, tc.rms_norm_eps, 10000.0)
    for i, (m, p, v, o) in enumerate(zip(lanes, prompts, pvs, outs)):
        assert o.shape == (1, p.shape[1] + 7) and o[0, :p.shape[1]].tolist() == p[0].tolist()
        emb, start = m._merge(p.cuda(), m._visual_embeds(pixel_values=v))
        want = RefDecoder(FAMILY_LLAVA, rc, sd, [0.3, 0.5, 0.7], seed=24).generate(emb.cpu(), start, 16, 7)
        assert o[0, p.shape[1]:].tolist() == want, f"lane {i}"
        assert m.start_image_pos == [start] and m.image_features[1].shape == (1, 16, 5)
    # and the same image through the solo path of a fresh lane
    solo = m0.spawn_lane()
    o1 = solo.generate(input_ids=prompts[1], pixel_values=pvs[1], max_new_tokens=7, eos_token_id=[])
    assert o1.tolist() == outs[1].tolist()
    with pytest.raises(ValueError):
        generate_group(lanes, [dict(input_ids=prompts[0], pixel_values=pvs[0])] * 2)
    # GroupPipeline: batches back to back, the next batch's vision tower + prefill overlapped with the current decode.
    # Lane b of set s sees batches s, s+2, ... and its rng stream continues across them — like one reference process.
    from dropoutdecoding_amd.vlm import GroupPipeline
    m1 = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
    pipe = GroupPipeline(m1, lanes=2)
    imgs = [(prompts[i % 3], torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(40 + i))) for i in range(6)]
    batches = [[dict(input_ids=p, pixel_values=v) for p, v in imgs[2 * b:2 * b + 2]] for b in range(3)]
    got = [o for outs in pipe.run(batches, max_new_tokens=6, eos_token_id=[]) for o in outs]
    assert len(got) == 6
    for lane_imgs in ([0, 4], [1, 5], [2], [3]):            # (set 0, lane 0), (set 0, lane 1), (set 1, lane 0), (set 1, lane 1)
        solo = m1.spawn_lane()
        for i in lane_imgs:
            want = solo.generate(input_ids=imgs[i][0], pixel_values=imgs[i][1], max_new_tokens=6, eos_token_id=[])
            assert got[i].tolist() == want.tolist(), f"image {i}"
    # an oversized batch is refused, also when it is the first one
    with pytest.raises(ValueError):
        list(pipe.run([[dict(input_ids=prompts[0], pixel_values=pvs[0])] * 3], max_new_tokens=4, eos_token_id=[]))
