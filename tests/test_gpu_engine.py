"""The HIP decode-step engine vs the golden end-to-end vectors (the reference's forward) and vs the oracle.

Tolerances (DESIGN.md "Numerics"): token ids, mask flags, member argmax ids, winner index — exact;
logits within 1e-3 of the fp32 reference relative to the largest |logit| (north star: "1e-3 relative fp32").
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, rel=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rel * np.abs(b).max()


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _engine(E, g, family, use_random=False, seed=None, max_seq=256):
    v, d, f, nl, nh, nkv, hd = [int(x) for x in g["cfg"]]
    rcfg = RefCfg(v, d, f, nl, nh, nkv, hd, float(g["rms_eps"]), float(g["rope_theta"]))
    w = random_weights(rcfg, int(g["wseed"]), float(g["std"]))
    cfg = E.LMConfig(v, d, f, nl, nh, nkv, hd, float(g["rms_eps"]), float(g["rope_theta"]))
    eng = E.DropoutEngine(cfg, family=family, max_seq=max_seq, max_visual=int(g["span_len"]) + 8, seed=seed,
                          use_random=use_random)
    eng.load_state_dict(w)
    return eng, rcfg, w


CASES = [("g5_llava_k3.npz", FAMILY_LLAVA), ("g5_llava_k8.npz", FAMILY_LLAVA), ("g5_next_k4.npz", FAMILY_NEXT),
         ("g5_next_norestore_k2.npz", FAMILY_NEXT)]


@pytest.mark.parametrize("name,family", CASES)
@pytest.mark.parametrize("rng", ["injected", "mt19937"])
def test_end_to_end_golden(E, golden_dir, name, family, rng):
    g = _load(golden_dir, name)
    use_random = bool(int(g["use_random"]))
    eng, rcfg, w = _engine(E, g, family, use_random, seed=int(g["rseed"]))
    probs = [float(p) for p in g["probs"]]
    s0, L = int(g["span_start"]), int(g["span_len"])
    eng.prefill(torch.from_numpy(g["embeds"]).cuda(), s0, L)
    assert close(eng.logits(), g["prefill_logits_last"])
    assert close(eng.image_logits(), g["prefill_image_logits"])
    u = eng.vision_uncert_dict()
    np.testing.assert_allclose(u["epis_uncert_per_token"], g["epis_uncert_per_token"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(u["alea_uncert_per_token"], g["alea_uncert_per_token"], rtol=1e-3)
    np.testing.assert_allclose(u["variance_per_token"], g["variance_per_token"], rtol=5e-3)
    assert set(map(tuple, np.sort(eng.topk()[1], 1))) == set(map(tuple, np.sort(g["topk_ids"], 1)))
    toks = eng.tokens()
    assert toks == [int(g["tokens"][0])]
    for s in range(len(g["tokens"]) - 1):
        uni = torch.from_numpy(g["uniforms"][s]).cuda() if rng == "injected" else None
        eng.decode_step(probs, uniforms=uni)
        st = eng.last_step()
        assert int(np.argmax(eng.base_logits())) == int(g["step_base_argmax"][s])
        assert close(eng.base_logits(), g["step_base_logits"][s])
        np.testing.assert_array_equal(st["drop"], g["step_drop"][s].astype(bool), err_msg=f"step {s}")
        assert st["member_argmax"].tolist() == g["step_member_argmax"][s].tolist()
        assert st["winner"] == int(g["step_winner"][s])
        assert close(eng.logits(), g["step_logits"][s])
        if "step_masked_numbers" in g.files:
            assert st["masked_numbers"].tolist() == g["step_masked_numbers"][s].tolist()
    assert eng.tokens() == g["tokens"].tolist()
    assert eng.T() == int(g["kv_len"])
    kv = eng.kv_sums()
    np.testing.assert_allclose(kv[:, 0], g["kv_k_sum"], atol=2e-2)
    np.testing.assert_allclose(kv[:, 1], g["kv_v_sum"], atol=2e-2)
    eng.close()


def test_end_to_end_golden_instructblip(E, golden_dir):
    g = _load(golden_dir, "g5_iblip_k3.npz")
    eng, rcfg, w = _engine(E, g, FAMILY_IBLIP)
    probs = [float(p) for p in g["probs"]]
    eng.prefill(torch.from_numpy(g["embeds"]).cuda(), 0, int(g["span_len"]))
    np.testing.assert_allclose(eng.vision_uncert_dict()["epis_uncert_per_token"], g["epis_uncert_per_token"], rtol=2e-3, atol=1e-6)
    for s in range(len(g["tokens"]) - 1):
        eng.decode_step(probs)
        st = eng.last_step()
        np.testing.assert_array_equal(st["drop"], g["step_drop"][s].astype(bool), err_msg=f"step {s}")
        assert st["member_argmax"].tolist() == g["step_member_argmax"][s].tolist()     # Q3: hidden-state argmax
    assert eng.tokens() == g["tokens"].tolist()
    eng.close()


def test_original_greedy_matches_oracle(E, golden_dir):
    """`--original` (K=0): stock greedy decode, BASELINE configs[0]."""
    g = _load(golden_dir, "g5_llava_k3.npz")
    eng, rcfg, w = _engine(E, g, FAMILY_LLAVA)
    emb = torch.from_numpy(g["embeds"])
    s0, L = int(g["span_start"]), int(g["span_len"])
    ref = RefDecoder(FAMILY_LLAVA, rcfg, w, [], dropout=False)
    want = ref.generate(emb, s0, L, 10)
    eng.prefill(emb.cuda(), s0, L)
    got = eng.generate(10, dropout=False)
    assert got == want
    assert close(eng.logits(), ref.records[-1].logits)
    eng.close()


CHAIR_K4 = [0.1, 0.3, 0.5, 0.7]      # BASELINE config 2: the reference's shipped LLaVA-1.5 list (chair_test/chair_test.py:170)


@pytest.mark.parametrize("family,K", [(FAMILY_LLAVA, 8), (FAMILY_NEXT, 3), (FAMILY_IBLIP, 4), (FAMILY_LLAVA, 1), (FAMILY_LLAVA, "chair4"),
                                      (FAMILY_NEXT, "chair4")])
def test_longer_decode_vs_oracle_first_divergence(E, golden_dir, family, K):
    """24 steps against the oracle on seeded inputs; reports the first divergence and the margin there."""
    gname = {FAMILY_LLAVA: "g5_llava_k3.npz", FAMILY_NEXT: "g5_next_k4.npz", FAMILY_IBLIP: "g5_iblip_k3.npz"}[family]
    g = _load(golden_dir, gname)
    probs = CHAIR_K4 if K == "chair4" else ([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K] if K > 1 else [0.5])
    eng, rcfg, w = _engine(E, g, family, seed=77)
    ref = RefDecoder(family, rcfg, w, probs, seed=77)
    emb = torch.from_numpy(g["embeds"])
    s0, L = int(g["span_start"]), int(g["span_len"])
    n_new = 25
    want = ref.generate(emb, s0, L, n_new)
    eng.prefill(emb.cuda(), s0, L)
    for s in range(n_new - 1):
        eng.decode_step(probs)
        st = eng.last_step()
        r = ref.records[s]
        top2 = np.sort(r.logits)[-2:]
        info = f"step {s}: oracle margin {top2[1] - top2[0]:.3g}"
        np.testing.assert_array_equal(st["drop"], r.drop, err_msg=info)
        assert st["member_argmax"].tolist() == r.member_argmax, info
        assert st["winner"] == r.winner, info
        assert close(eng.logits(), r.logits), info
    assert eng.tokens() == want
    eng.close()


def test_prefill_rejects_bad_span(E, golden_dir):
    g = _load(golden_dir, "g5_llava_k3.npz")
    eng, _, _ = _engine(E, g, FAMILY_LLAVA)
    emb = torch.from_numpy(g["embeds"]).cuda()
    with pytest.raises(ValueError):                       # reference llava.py:134-138: ValueError on count mismatch
        eng.prefill(emb, 30, 36)
    with pytest.raises(ValueError):
        eng.prefill(emb[:, :128], 1, 36)
    with pytest.raises(Exception):
        eng.decode_step([0.3])                            # decode before prefill
    eng.close()


def test_properties_mid_size_synthetic(E):
    """Size-independent properties on a wider model (d=1024, GQA 8/4... ) with synthetic weights:
    determinism, mprob<=0.1-with-keep... and K members with nothing dropped reproduce the un-masked pass."""
    cfg = E.LMConfig(4096, 1024, 2816, 4, 8, 8, 128, 1e-5, 10000.0)
    eng = E.DropoutEngine(cfg, family=FAMILY_IBLIP, max_seq=512, max_visual=64, seed=1)
    eng.load_synthetic(3, 0.03)
    gen = torch.Generator().manual_seed(0)
    emb = torch.randn(80, 1024, generator=gen).cuda()
    eng.prefill(emb, 4, 64)
    first = eng.tokens()
    # InstructBLIP rule drops tokens with epi >= quantile(1 - p): p -> 0 keeps everything except the maximum
    a = eng.generate(12, mprobs=[0.3, 0.5, 0.7])
    eng.prefill(emb, 4, 64)
    b = eng.generate(12, mprobs=[0.3, 0.5, 0.7])
    assert a == b and a[0] == first[0]                     # deterministic replay
    eng.prefill(emb, 4, 64)
    c = eng.generate(12, dropout=False)
    eng2 = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=512, max_visual=64, seed=1)
    eng2.load_synthetic(3, 0.03)
    eng2.prefill(emb, 4, 64)
    # uniforms = 1.0 can never be < p, so no member drops anything: every member == the un-masked pass
    ones = torch.ones(8, 64).cuda()
    for _ in range(11):
        eng2.decode_step([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8], uniforms=ones)
        st = eng2.last_step()
        assert st["drop"].sum() == 0 and st["winner"] == 0 and len(set(st["member_argmax"].tolist())) == 1
        assert close(eng2.logits(), eng2.base_logits(), 1e-4)     # same math, different row packing (1 vs 8 rows)
    assert eng2.tokens() == c
    eng.close()
    eng2.close()


def _rand_case(E, family, rcfg, T0, s0, L, probs, n_new, seed=3, wseed=21, std=0.05, max_seq=None):
    from oracle.decode_ref import RefDecoder
    w = random_weights(rcfg, wseed, std)
    cfg = E.LMConfig(rcfg.vocab_size, rcfg.hidden_size, rcfg.intermediate_size, rcfg.num_layers, rcfg.num_heads,
                     rcfg.num_kv_heads, rcfg.head_dim, rcfg.rms_eps, rcfg.rope_theta)
    eng = E.DropoutEngine(cfg, family=family, max_seq=max_seq or (T0 + n_new + 8), max_visual=L, seed=seed)
    eng.load_state_dict(w)
    emb = torch.randn(T0, rcfg.hidden_size, generator=torch.Generator().manual_seed(seed)) * 0.8
    ref = RefDecoder(family, rcfg, w, probs, seed=seed)
    want = ref.generate(emb, s0, L, n_new)
    eng.prefill(emb.cuda(), s0, L)
    assert close(eng.logits(), ref.prefill_logits[-1].numpy())
    for s in range(n_new - 1):
        eng.decode_step(probs)
        st, r = eng.last_step(), ref.records[s]
        top2 = np.sort(r.logits)[-2:]
        info = f"step {s}: oracle margin {top2[1] - top2[0]:.3g}"
        np.testing.assert_array_equal(st["drop"], r.drop, err_msg=info)
        assert st["member_argmax"].tolist() == r.member_argmax, info
        assert st["winner"] == r.winner, info
        assert close(eng.logits(), r.logits), info
    assert eng.tokens() == want
    eng.close()


def test_two_packed_sweeps_K12(E):
    """K > 8: members run as two packed sweeps (8 + 4); masks are cumulative across all 12 (LLaVA-1.5 rule)."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    _rand_case(E, FAMILY_LLAVA, rc, 40, 3, 30, [0.1 + 0.05 * i for i in range(12)], 6)


def test_three_packed_sweeps_K20(E):
    """K up to DD_MAX_MEMBERS = 64 (round 5; models/llava.py:340 takes any list): 20 members = three packed sweeps (8 + 8 + 4), three bit planes of
    drop flags, cumulative masks across all 20 — tokens, masks, votes against the oracle."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    _rand_case(E, FAMILY_LLAVA, rc, 40, 3, 30, [0.1 + 0.035 * i for i in range(20)], 5)


def test_eight_packed_sweeps_K64_the_limit(E):
    """DD_MAX_MEMBERS itself — 64 members = eight packed sweeps, eight bit planes of drop flags, a full 64-entry
    probability table; the members' logits buffer starts at 16 rows and grows on the first such step (dd_engine.hip ensure_member_rows),
    with a K = 5 step BEFORE (graphs captured against the small buffer must not be replayed) and the NeXT rule's reset masks after."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    cfg = E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 21, 0.05)
    probs64 = [0.1 + 0.0125 * i for i in range(64)]
    for family in (FAMILY_LLAVA, FAMILY_NEXT):
        eng = E.DropoutEngine(cfg, family=family, max_seq=64, max_visual=30, seed=3)
        eng.load_state_dict(w)
        emb = torch.randn(40, 256, generator=torch.Generator().manual_seed(3)) * 0.8
        eng.prefill(emb.cuda(), 3, 30)
        eng.decode_step(probs64[:5])
        eng.decode_step(probs64[:5])                          # (replayed from the graph cache)
        first = eng.tokens()
        eng.rng.manual_seed(3)
        eng.prefill(emb.cuda(), 3, 30)
        ref = RefDecoder(family, rc, w, probs64, seed=3)
        want = ref.generate(emb, 3, 30, 5)
        for s in range(4):
            eng.decode_step(probs64)
            st, r = eng.last_step(), ref.records[s]
            np.testing.assert_array_equal(st["drop"], r.drop, err_msg=f"{family} step {s}")
            assert st["masked_numbers"].tolist() == r.masked_numbers
            assert st["member_argmax"].tolist() == r.member_argmax and st["winner"] == r.winner, (family, s)
            assert close(eng.logits(), r.logits), (family, s)
        assert eng.tokens() == want
        eng.rng.manual_seed(3)
        eng.prefill(emb.cuda(), 3, 30)                        # and back to a short list on the grown buffer
        eng.decode_step(probs64[:5])
        eng.decode_step(probs64[:5])
        assert eng.tokens() == first
        with pytest.raises(Exception):
            eng.decode_step(probs64 + [0.5])                  # 65 members
        eng.close()


def test_gqa_group4_long_context_many_key_tiles(E):
    """Mistral-style GQA (8 q heads / 2 kv heads), theta 1e6, 300-token prefix: 5 key tiles per kv head, visual span
    crossing tile boundaries."""
    rc = RefCfg(512, 1024, 512, 2, 8, 2, 128, 1e-5, 1000000.0)
    _rand_case(E, FAMILY_NEXT, rc, 300, 7, 250, [0.2, 0.5, 0.8], 5, std=0.03)


def test_vocab_not_multiple_of_16_instructblip(E):
    """Vicuna's V=32001 style: padded vocabulary rows must never win an argmax or leak into the scorer."""
    rc = RefCfg(509, 256, 512, 2, 2, 2, 128, 1e-6, 10000.0)
    _rand_case(E, FAMILY_IBLIP, rc, 40, 0, 32, [0.3, 0.5, 0.7], 6)


def test_kv_capacity_and_state_errors(E):
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    cfg = E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=64, max_visual=16)
    eng.load_state_dict(random_weights(rc, 1, 0.05))
    eng.prefill(torch.randn(60, 256).cuda(), 2, 16)
    n = 0
    with pytest.raises(Exception, match="KV cache full"):
        for n in range(10):
            eng.decode_step([0.3])
    assert 1 <= n <= 4
    with pytest.raises(ValueError):
        eng.prefill(torch.randn(70, 256).cuda(), 2, 16)        # longer than the KV capacity
    with pytest.raises(ValueError):
        E.DropoutEngine(E.LMConfig(512, 200, 512, 2, 2, 2, 128), family=FAMILY_LLAVA)   # hidden not a multiple of 256
    eng.close()


def test_instructblip_positions_from_leaked_mask_444_rule(E, golden_dir):
    """SURVEY Q2 switch: under transformers 4.44 the decode position is cumsum(mask)-1 of the leaked mask."""
    g = _load(golden_dir, "g5_iblip_k3.npz")
    v, d, f, nl, nh, nkv, hd = [int(x) for x in g["cfg"]]
    rcfg = RefCfg(v, d, f, nl, nh, nkv, hd, float(g["rms_eps"]), float(g["rope_theta"]))
    w = random_weights(rcfg, int(g["wseed"]), float(g["std"]))
    probs = [0.3, 0.5, 0.7]
    emb = torch.from_numpy(g["embeds"])
    ref = RefDecoder(FAMILY_IBLIP, rcfg, w, probs, iblip_positions="mask")
    want = ref.generate(emb, 0, 32, 8)
    ref_cache = RefDecoder(FAMILY_IBLIP, rcfg, w, probs, iblip_positions="cache").generate(emb, 0, 32, 8)
    eng = E.DropoutEngine(E.LMConfig(v, d, f, nl, nh, nkv, hd, float(g["rms_eps"]), float(g["rope_theta"])),
                          family=FAMILY_IBLIP, max_seq=128, max_visual=32, iblip_positions="mask")
    eng.load_state_dict(w)
    eng.prefill(emb.cuda(), 0, 32)
    got = eng.generate(8, mprobs=probs)
    assert got == want
    assert close(eng.logits(), ref.records[-1].logits)
    eng.close()


def test_full_size_llava15_7b_properties(E):
    """BASELINE config size (LLaVA-1.5-7B shapes, 576 visual + 32 prompt tokens, K=8), synthetic weights.
    Size-independent properties at the full depth — the oracle comparisons of these widths are
    tests/test_gpu_7b_shapes_vs_oracle.py (two layers, every lane and step form against its own RefDecoder) and, once per round on
    the GPU box, tests/test_gpu_full_depth_oracle.py (all 32 layers; profiles/r06_full_size_oracle.log):
    replay determinism, members == un-masked pass when nothing is dropped (the vote is then unanimous and the
    token equals stock greedy), masks obey the reference's invariants, K-sharded phases == the single call."""
    eng = E.DropoutEngine(E.LLAVA15_7B, family=FAMILY_LLAVA, max_seq=704, max_visual=576, seed=5217)
    eng.load_synthetic(1, 0.02)
    emb = torch.randn(608, 4096, generator=torch.Generator().manual_seed(0)).cuda()
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    eng.prefill(emb, 5, 576)
    u = eng.vision_uncert_dict()
    epi = u["epis_uncert_per_token"][0]
    assert np.isfinite(epi).all() and (epi > -1e-6).all()              # KL(p_l || mean p) >= 0
    assert np.isfinite(u["alea_uncert_per_token"]).all() and (u["alea_uncert_per_token"] >= 0).all()
    a = eng.generate(6, mprobs=probs)
    st = eng.last_step()
    nd = st["masked_numbers"]
    assert (np.diff(nd) >= -int(st["keep"].sum())).all()               # cumulative masks (llava.py:344), modulo kept tokens
    assert not (st["drop"] & st["keep"][None]).any()                   # kept tokens are never dropped (llava.py:660)
    assert 0.03 * 576 < nd[0] < 0.2 * 576 and nd[-1] > nd[0]           # p in [0.1, mprob]
    eng.rng.manual_seed(5217)
    eng.prefill(emb, 5, 576)
    assert eng.generate(6, mprobs=probs) == a                          # deterministic replay under a fixed seed
    # nothing dropped -> every member reproduces the un-masked pass -> token == stock greedy
    eng.prefill(emb, 5, 576)
    ones = torch.ones(8, 576).cuda()
    for _ in range(4):
        eng.decode_step(probs, uniforms=ones)
        s2 = eng.last_step()
        assert s2["drop"].sum() == 0 and s2["winner"] == 0 and len(set(s2["member_argmax"].tolist())) == 1
        assert close(eng.logits(), eng.base_logits(), 2e-4)
    with_members = eng.tokens()
    eng.prefill(emb, 5, 576)
    assert eng.generate(5, dropout=False) == with_members
    # phased (K-shard) API == single call, bit for bit
    eng.rng.manual_seed(9)
    eng.prefill(emb, 5, 576)
    for _ in range(3):
        eng.decode_step(probs)
    ref_toks, ref_logits = eng.tokens(), eng.logits()
    eng.rng.manual_seed(9)
    eng.prefill(emb, 5, 576)
    ids, rec = eng.new_xchg_buffers()
    torch.cuda.synchronize()
    for _ in range(3):
        eng.step_base(probs)
        eng.step_members(0, 8)
        eng.export_ids(0, 8, ids); eng.import_ids(ids)
        eng.export_winner(0, 8, rec); eng.import_winner(rec)
        eng.step_commit()
    assert eng.tokens() == ref_toks
    np.testing.assert_array_equal(eng.logits(), ref_logits)
    # BASELINE config 2: K = 4 with the reference's shipped list [0.1, 0.3, 0.5, 0.7] (chair_test/chair_test.py:170) at this size:
    # the same invariants, replay determinism, and speculative == two-sweep steps bit for bit
    eng.rng.manual_seed(24)
    eng.prefill(emb, 5, 576)
    k4 = eng.generate(8, mprobs=CHAIR_K4)
    st = eng.last_step()
    nd = st["masked_numbers"]
    assert st["drop"].shape == (4, 576) and not (st["drop"] & st["keep"][None]).any()
    assert (np.diff(nd) >= -int(st["keep"].sum())).all() and 0.03 * 576 < nd[0] < 0.2 * 576 and nd[-1] > nd[0]
    assert len(set(st["member_argmax"].tolist()) | {st["voted"]}) <= 4 and st["member_argmax"][st["winner"]] == st["voted"]
    k4_logits, k4_sums = eng.logits(), eng.kv_sums().copy()
    eng.set_speculation("never")
    eng.rng.manual_seed(24)
    eng.prefill(emb, 5, 576)
    assert eng.generate(8, mprobs=CHAIR_K4) == k4
    np.testing.assert_array_equal(eng.logits(), k4_logits)
    np.testing.assert_array_equal(eng.kv_sums(), k4_sums)
    eng.set_speculation("default")
    eng.close()


def test_fp8_weight_storage_next_family(E):
    """BASELINE config 5 style: matrices stored as OCP fp8 e4m3fn + per-row scales, expanded exactly to bf16 in
    registers.  The oracle runs on the DEQUANTISED weights (scale * q in fp32), so the same tolerances apply."""
    from dropoutdecoding_amd.lm import dequantize_fp8, quantize_fp8
    rc = RefCfg(512, 512, 512, 2, 4, 2, 128, 1e-5, 1000000.0)
    w = random_weights(rc, 33, 0.05)
    wq = {}
    for k, v in w.items():
        if v.dim() == 2 and "embed_tokens" not in k:
            q, s = quantize_fp8(v)
            wq[k] = dequantize_fp8(q, s)
        else:
            wq[k] = v
    probs = [0.2, 0.4, 0.6, 0.8]
    eng = E.DropoutEngine(E.LMConfig(512, 512, 512, 2, 4, 2, 128, 1e-5, 1000000.0), family=FAMILY_NEXT, max_seq=160,
                          max_visual=88, seed=4, weight_format="fp8")
    eng.load_state_dict(w)
    emb = torch.randn(100, 512, generator=torch.Generator().manual_seed(4)) * 0.8
    ref = RefDecoder(FAMILY_NEXT, rc, wq, probs, seed=4)
    want = ref.generate(emb, 6, 88, 10)
    eng.prefill(emb.cuda(), 6, 88)
    assert close(eng.logits(), ref.prefill_logits[-1].numpy())
    np.testing.assert_allclose(eng.vision_uncert_dict()["epis_uncert_per_token"][0], ref.epi.numpy(), rtol=5e-3, atol=1e-6)
    for s in range(9):
        eng.decode_step(probs)
        st, r = eng.last_step(), ref.records[s]
        np.testing.assert_array_equal(st["drop"], r.drop, err_msg=f"step {s}")
        assert st["member_argmax"].tolist() == r.member_argmax and st["winner"] == r.winner, s
        assert close(eng.logits(), r.logits), s
    assert eng.tokens() == want
    with pytest.raises(ValueError):
        eng._load(E.T_WQ, 0, w["model.layers.0.self_attn.q_proj.weight"])      # bf16 matrix into an fp8 engine
    eng.close()


def test_fp8_conversion_is_ocp_e4m3fn(E):
    """Every finite e4m3fn byte through the kernel's fp8 -> bf16 expansion (a 256x256 identity-like probe)."""
    from dropoutdecoding_amd.lm import dequantize_fp8
    # one matrix row per byte value: W[n, k] = byte n at k == 0, zero elsewhere; x = e_0 picks it out through lm_head
    V, d = 256, 256
    rc = RefCfg(V, d, 256, 1, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 1, 0.05)
    eng = E.DropoutEngine(E.LMConfig(V, d, 256, 1, 2, 2, 128, 1e-5, 10000.0), family=FAMILY_LLAVA, max_seq=64, max_visual=8,
                          weight_format="fp8")
    eng.load_state_dict(w)
    q = torch.zeros(V, d, dtype=torch.uint8)
    q[:, 0] = torch.arange(256, dtype=torch.uint8)
    q[127, 0] = 0
    q[255, 0] = 0                                        # the two NaN encodings
    scale = torch.ones(V)
    import ctypes as C
    from dropoutdecoding_amd import _lib
    _lib.check(eng.lib.dd_lm_load_tensor_fp8(eng._h, E.T_LM_HEAD, 0, q.data_ptr(), scale.data_ptr(), V, d, 0), "load")
    eng.prefill(torch.randn(10, d).cuda(), 1, 8)
    # logits = lm_head @ hidden: column 0 of lm_head times hidden[0]; compare ratios to the torch decode of the bytes
    hid = eng.hidden()
    want = dequantize_fp8(q, scale)[:, 0].numpy() * hid[0]
    got = eng.logits()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-6 * np.abs(want).max())
    eng.close()


def test_mid_scale_llava_shapes_vs_oracle(E):
    """SURVEY's mid-scale configuration (d=1024, 8 layers, the real V=32064, L=576 visual + 32 prompt tokens, K=8):
    the largest size the oracle finishes in about a minute; exercises every multi-tile path (11 key tiles per head,
    2004 vocabulary tiles, 576-token mask rows) against the reference-faithful CPU restatement."""
    rc = RefCfg(32064, 1024, 2816, 8, 8, 8, 128, 1e-5, 10000.0)
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    w = random_weights(rc, 5, 0.03)
    cfg = E.LMConfig(32064, 1024, 2816, 8, 8, 8, 128, 1e-5, 10000.0)
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=640, max_visual=576, seed=5217)
    eng.load_state_dict(w)
    emb = torch.randn(608, 1024, generator=torch.Generator().manual_seed(9)) * 0.8
    ref = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=5217)
    want = ref.generate(emb, 5, 576, 6)
    eng.prefill(emb.cuda(), 5, 576)
    assert close(eng.logits(), ref.prefill_logits[-1].numpy())
    u = eng.vision_uncert_dict()
    np.testing.assert_allclose(u["epis_uncert_per_token"][0], ref.epi.numpy(), rtol=5e-3, atol=1e-6)
    assert set(map(tuple, np.sort(eng.topk()[1], 1))) == set(map(tuple, np.sort(ref.topk_ids.numpy(), 1)))
    for s in range(5):
        eng.decode_step(probs)
        st, r = eng.last_step(), ref.records[s]
        p = 0.1 + (np.array(probs)[:, None] - 0.1) * ((ref.epi.numpy() - ref.epi.numpy().min()) / np.ptp(ref.epi.numpy()))[None]
        margin = np.abs(r.uniforms - p).min()
        info = f"step {s}: min |r - p| = {margin:.2e}"
        np.testing.assert_array_equal(st["keep"], r.keep, err_msg=info)
        if margin > 1e-5:                                   # a uniform within 1e-5 of its threshold may legitimately flip
            np.testing.assert_array_equal(st["drop"], r.drop, err_msg=info)
            assert st["masked_numbers"].tolist() == r.masked_numbers, info
        assert st["member_argmax"].tolist() == r.member_argmax, info
        assert st["winner"] == r.winner, info
        assert close(eng.logits(), r.logits), info
    assert eng.tokens() == want
    eng.close()


def test_graph_replay_equals_eager_launches(E):
    """Decode steps replayed from the hipGraph cache (the default) == the same steps launched kernel by kernel:
    tokens, masks and logits bit for bit, across a 64-token attention-tile boundary (new graph) and a change of K."""
    from dropoutdecoding_amd import _lib
    L = _lib.load()
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 17, 0.05)
    emb = torch.randn(58, 256, generator=torch.Generator().manual_seed(3)).cuda()
    outs = []
    for graph in (0, 1):
        L.dd_set_tuning(8, graph)
        eng = E.DropoutEngine(E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=FAMILY_LLAVA, max_seq=160,
                              max_visual=40, seed=11)
        eng.load_state_dict(w)
        eng.prefill(emb, 4, 40)
        rec = []
        for s in range(14):                                   # T: 58 -> 72 crosses the 64-key tile boundary
            probs = [0.3, 0.5, 0.7] if s < 9 else [0.2, 0.4, 0.6, 0.8]
            eng.decode_step(probs)
            st = eng.last_step()
            rec.append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], eng.logits().copy()))
        outs.append((eng.tokens(), rec, eng.kv_sums().copy()))
        eng.close()
    L.dd_set_tuning(8, 1)
    assert outs[0][0] == outs[1][0]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    for (d0, a0, w0, l0), (d1, a1, w1, l1) in zip(outs[0][1], outs[1][1]):
        np.testing.assert_array_equal(d0, d1)
        assert a0 == a1 and w0 == w1
        np.testing.assert_array_equal(l0, l1)


def _variant_case(E, family, rc, T0, s0, L, probs, n_new, seed=5, eng_kw=None, ref_kw=None, first=False, std=0.05):
    """Engine vs oracle with the dormant variants switched on (same checks as _rand_case)."""
    w = random_weights(rc, 23, std)
    cfg = E.LMConfig(rc.vocab_size, rc.hidden_size, rc.intermediate_size, rc.num_layers, rc.num_heads, rc.num_kv_heads,
                     rc.head_dim, rc.rms_eps, rc.rope_theta)
    eng = E.DropoutEngine(cfg, family=family, max_seq=T0 + n_new + 8, max_visual=L, seed=seed, **(eng_kw or {}))
    eng.load_state_dict(w)
    emb = torch.randn(T0, rc.hidden_size, generator=torch.Generator().manual_seed(seed)) * 0.8
    ref = RefDecoder(family, rc, w, probs, seed=seed, first_step_ensemble=first, **(ref_kw or {}))
    want = ref.generate(emb, s0, L, n_new)
    eng.prefill(emb.cuda(), s0, L, first_step_ensemble=first, mprobs=probs)
    avg = bool((eng_kw or {}).get("use_avg"))
    recs = ([ref.first_record] if first else []) + ref.records
    for s, r in enumerate(recs):
        if s > 0 or not first:
            eng.decode_step(probs)
        st = eng.last_step()
        info = f"record {s}"
        np.testing.assert_array_equal(st["drop"], r.drop, err_msg=info)
        np.testing.assert_array_equal(st["masked_numbers"], r.masked_numbers, err_msg=info)
        if not avg:                                    # the mean replaces member 0's row, so its argmax is the mean's
            assert st["member_argmax"].tolist() == r.member_argmax, info
        assert st["winner"] == r.winner, info
        assert close(eng.logits(), r.logits), info
        if first and s == 0:      # the masked prompt pass must really differ from the un-masked one
            assert np.abs(r.logits - r.base_logits).max() > 1e-2 * np.abs(r.base_logits).max()
            assert not close(eng.logits(), r.base_logits, 1e-3)
    assert eng.tokens() == want
    sums = eng.kv_sums()
    eng.close()
    return want, sums


@pytest.mark.parametrize("family,K", [(FAMILY_LLAVA, 3), (FAMILY_NEXT, 4), (FAMILY_LLAVA, 8)])
def test_first_token_ensemble_toggle(E, family, K):
    """`# if True:` at llava.py:336 — the ensemble also picks the first token: every member re-runs
    the whole prompt with its masked columns from an empty cache; the winner's cache continues (incl. a winner != 0)."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    probs = [0.1 + 0.1 * i for i in range(K)]
    _variant_case(E, family, rc, 70, 3, 50, probs, 5, first=True)


def test_first_token_ensemble_needs_a_text_token_before_the_span(E):
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    eng = E.DropoutEngine(E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=FAMILY_IBLIP, max_seq=64, max_visual=32)
    eng.load_state_dict(random_weights(rc, 1, 0.05))
    with pytest.raises(ValueError, match="position 0"):
        eng.prefill(torch.randn(40, 256).cuda(), 0, 32, first_step_ensemble=True, mprobs=[0.3, 0.5])
    eng.close()


@pytest.mark.parametrize("family", [FAMILY_LLAVA, FAMILY_NEXT])
def test_average_instead_of_vote(E, family):
    """settings['use_avg'] / select_by_average (llava.py:37-52): logits := fp32 mean over members, member 0's cache."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    _variant_case(E, family, rc, 50, 2, 40, [0.2, 0.4, 0.6, 0.8, 0.9], 6, eng_kw=dict(use_avg=True), ref_kw=dict(use_avg=True))


@pytest.mark.parametrize("family,s0", [(FAMILY_LLAVA, 4), (FAMILY_IBLIP, 0)])
def test_epis_no_overlap_method(E, family, s0):
    """The dormant `epis_no_overlap` branch (llava.py:663-683 cumulative, instructblip.py:486-505 reset + random)."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    _variant_case(E, family, rc, 48, s0, 32, [0.3, 0.5, 0.7], 6, eng_kw=dict(mask_method="epis_no_overlap"),
                  ref_kw=dict(mask_method="epis_no_overlap"))


def _lane_setup(E, family, rc, shapes, seed=7, max_seq=160, **kw):
    """engines[0] owns the weights; the others are lanes over them.  shapes: (T0, span_start, L) per sequence."""
    w = random_weights(rc, 31, 0.05)
    cfg = E.LMConfig(rc.vocab_size, rc.hidden_size, rc.intermediate_size, rc.num_layers, rc.num_heads, rc.num_kv_heads,
                     rc.head_dim, rc.rms_eps, rc.rope_theta)
    Lmax = max(s[2] for s in shapes)
    engines = []
    for i in range(len(shapes)):
        e = E.DropoutEngine(cfg, family=family, max_seq=max_seq, max_visual=Lmax, seed=seed,
                            share_weights_with=engines[0] if engines else None, **kw)
        engines.append(e)
    engines[0].load_state_dict(w)
    embs = [torch.randn(T0, rc.hidden_size, generator=torch.Generator().manual_seed(100 + i)) * 0.8
            for i, (T0, _, _) in enumerate(shapes)]
    return w, engines, embs


@pytest.mark.parametrize("family,shapes", [
    (FAMILY_LLAVA, [(40, 3, 30), (70, 5, 50), (33, 1, 30), (66, 2, 60)]),       # lengths either side of a 64-key tile
    (FAMILY_LLAVA, [(34 + 3 * i, 1 + i % 3, 30) for i in range(11)]),           # > 8 lanes: base rows in two operand planes
    (FAMILY_LLAVA, [(34 + 2 * i, 1 + i % 3, 30) for i in range(21)]),           # > 16 lanes: four planes, two attention launches
    (FAMILY_LLAVA, [(34 + (5 * i) % 37, 1 + i % 3, 30) for i in range(37)]),    # > 32 lanes: eight planes, two sampler launches
    (FAMILY_NEXT, [(90, 4, 80), (50, 4, 40)]),
    (FAMILY_IBLIP, [(40, 0, 32), (45, 0, 32), (38, 0, 32)]),                    # leaked mask bits per lane
])
def test_group_step_each_lane_equals_the_oracle_and_a_solo_run(E, family, shapes):
    """dd_lm_group_step: the base passes of all lanes are one sweep; every lane must still produce exactly what it
    produces alone (bit-identical logits) and what the oracle produces (each lane = one reference process, own rng)."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    probs = [0.2, 0.4, 0.6, 0.8]
    w, engines, embs = _lane_setup(E, family, rc, shapes)
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    grp = E.EngineGroup(engines)
    n_steps = 6
    recs = [[] for _ in engines]
    for s in range(n_steps):
        grp.decode_step(probs)
        for i, e in enumerate(engines):
            st = e.last_step()
            recs[i].append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], e.logits().copy(), e.base_logits().copy()))
    toks = [e.tokens() for e in engines]
    sums = [e.kv_sums().copy() for e in engines]
    for i, ((T0, s0, L), emb) in enumerate(zip(shapes, embs)):
        ref = RefDecoder(family, rc, w, probs, seed=7)
        want = ref.generate(emb, s0, L, n_steps + 1)
        assert toks[i] == want, f"lane {i}"
        for s, r in enumerate(ref.records):
            np.testing.assert_array_equal(recs[i][s][0], r.drop, err_msg=f"lane {i} step {s}")
            assert recs[i][s][1] == r.member_argmax and recs[i][s][2] == r.winner
            assert close(recs[i][s][3], r.logits) and close(recs[i][s][4], r.base_logits)
    # solo runs on the same handles (each lane alone, eager + graph paths): bit-identical
    for i, (e, emb, (T0, s0, L)) in enumerate(zip(engines, embs, shapes)):
        e.rng.manual_seed(7)
        e.prefill(emb.cuda(), s0, L)
        for s in range(n_steps):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.logits(), recs[i][s][3])
            np.testing.assert_array_equal(e.base_logits(), recs[i][s][4])
        assert e.tokens() == toks[i]
        np.testing.assert_array_equal(e.kv_sums(), sums[i])
    for e in reversed(engines):
        e.close()


def test_group_generate_with_uneven_stops_and_stock_greedy(E):
    """Lanes stop at different times (EOS / n_new); the group goes on with fewer rows.  K = 0 (`--original`) in a group
    appends each lane's base-row K/V."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    shapes = [(40, 3, 30), (52, 5, 40), (47, 2, 40)]
    probs = [0.3, 0.5, 0.7]
    w, engines, embs = _lane_setup(E, FAMILY_LLAVA, rc, shapes)
    wants = [RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=7).generate(emb, s0, L, 10) for emb, (T0, s0, L) in zip(embs, shapes)]
    eos = wants[1][3]                                        # lane 1 stops early on this id (others only if they emit it)
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    got = E.EngineGroup(engines).generate(10, eos=[eos], mprobs=probs)
    for g, wnt in zip(got, wants):
        cut = wnt[:wnt.index(eos) + 1] if eos in wnt else wnt
        assert g == cut
    assert len(got[1]) <= 4
    # stock greedy
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    got0 = E.EngineGroup(engines).generate(8, dropout=False)
    for g, emb, (T0, s0, L) in zip(got0, embs, shapes):
        assert g == RefDecoder(FAMILY_LLAVA, rc, w, [], dropout=False).generate(emb, s0, L, 8)
    with pytest.raises(ValueError):
        other = E.DropoutEngine(E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=FAMILY_LLAVA, max_seq=160, max_visual=40)
        E.EngineGroup([engines[0], other])
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("kv_format", ["fp32", "fp16"])
def test_full_size_lanes_equal_solo_runs_bitwise(E, kv_format):
    """BASELINE size (LLaVA-1.5-7B shapes, K = 8): 11 lanes decoded as a group (fused base pass in two operand planes, one
    8-sequence = 64-row member sweep through the slice-pair kernels, one 2-sequence sweep and one single: every row width at
    full depth) == each lane alone, bit for bit — logits, masks, tokens, KV checksums.  With the fp16 cache the 64-row and the
    16-row sweep run concurrently on two streams (dd_engine.hip group_step_eager), with the fp32 cache one after the other."""
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    shapes = [(608, 5, 576), (640, 9, 576), (600, 1, 576), (615, 20, 576), (609, 5, 576), (700, 60, 576),   # tiles 10 / 11
              (610, 3, 576), (633, 7, 576), (655, 11, 576), (602, 2, 576), (690, 33, 576)]
    engs = []
    for i in range(len(shapes)):
        engs.append(E.DropoutEngine(E.LLAVA15_7B, family=FAMILY_LLAVA, max_seq=768, max_visual=576, seed=5217, kv_format=kv_format,
                                    share_weights_with=engs[0] if engs else None))
    engs[0].load_synthetic(1, 0.02)
    embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(50 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.prefill(x, s0, L)
    grp = E.EngineGroup(engs)
    n_steps, rec = 3, [[] for _ in engs]
    for s in range(n_steps):
        grp.decode_step(probs)
        for i, e in enumerate(engs):
            st = e.last_step()
            rec[i].append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], e.logits().copy(), e.base_logits().copy()))
    toks = [e.tokens() for e in engs]
    sums = [e.kv_sums().copy() for e in engs]
    assert len({tuple(t) for t in toks}) > 1                          # the lanes really decode different things
    for i, (e, x, (T0, s0, L)) in enumerate(zip(engs, embs, shapes)):
        e.rng.manual_seed(5217)
        e.prefill(x, s0, L)
        for s in range(n_steps):
            e.decode_step(probs)
            st = e.last_step()
            np.testing.assert_array_equal(st["drop"], rec[i][s][0], err_msg=f"lane {i} step {s}")
            assert st["member_argmax"].tolist() == rec[i][s][1] and st["winner"] == rec[i][s][2]
            np.testing.assert_array_equal(e.logits(), rec[i][s][3], err_msg=f"lane {i} step {s}")
            np.testing.assert_array_equal(e.base_logits(), rec[i][s][4], err_msg=f"lane {i} step {s}")
        assert e.tokens() == toks[i]
        np.testing.assert_array_equal(e.kv_sums(), sums[i])
    for e in reversed(engs):
        e.close()


def test_full_size_forty_lanes_equal_solo_runs_bitwise(E):
    """BASELINE size, 40 lanes in one group (the un-masked rows as ONE 64-row sweep: eight operand planes, three attention
    launches, lm_head over 64 rows; five 64-row member sweeps; the mask sampler in two launches): lanes from every part of the
    group == the same lane decoded alone, bit for bit."""
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    n = 40
    shapes = [(600 + (7 * i) % 90, 1 + (3 * i) % 20, 576) for i in range(n)]
    engs = []
    for i in range(n):
        engs.append(E.DropoutEngine(E.LLAVA15_7B, family=FAMILY_LLAVA, max_seq=704, max_visual=576, seed=5217,
                                    share_weights_with=engs[0] if engs else None, kv_format="fp16"))
    engs[0].load_synthetic(1, 0.02)
    embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(50 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.prefill(x, s0, L)
    grp = E.EngineGroup(engs)
    n_steps, rec = 3, [[] for _ in engs]
    for s in range(n_steps):
        grp.decode_step(probs)
        for i, e in enumerate(engs):
            st = e.last_step()
            rec[i].append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], e.logits().copy(), e.base_logits().copy()))
    toks = [e.tokens() for e in engs]
    sums = [e.kv_sums().copy() for e in engs]
    assert len({tuple(t) for t in toks}) > 1
    for i in (0, 7, 8, 31, 32, 33, 39):
        e, x, (T0, s0, L) = engs[i], embs[i], shapes[i]
        e.rng.manual_seed(5217)
        e.prefill(x, s0, L)
        for s in range(n_steps):
            e.decode_step(probs)
            st = e.last_step()
            np.testing.assert_array_equal(st["drop"], rec[i][s][0], err_msg=f"lane {i} step {s}")
            assert st["member_argmax"].tolist() == rec[i][s][1] and st["winner"] == rec[i][s][2]
            np.testing.assert_array_equal(e.logits(), rec[i][s][3], err_msg=f"lane {i} step {s}")
            np.testing.assert_array_equal(e.base_logits(), rec[i][s][4], err_msg=f"lane {i} step {s}")
        assert e.tokens() == toks[i]
        np.testing.assert_array_equal(e.kv_sums(), sums[i])
    with pytest.raises(ValueError):
        E.EngineGroup(engs + [engs[0]] * 25)
    for e in reversed(engs):
        e.close()


def test_lanes_mid_scale_against_the_oracle(E):
    """d = 1024, d_ff = 2816 (K = 2816: 11 k-steps per wave, main loop + tail of the grouped GEMV), GQA 2, 4 layers, V = 4099:
    five lanes (one 4-sequence sweep + one plain 8-row sweep) against the oracle, one fresh rng stream per lane."""
    rc = RefCfg(4099, 1024, 2816, 4, 8, 4, 128, 1e-5, 10000.0)
    probs = [0.1, 0.3, 0.5, 0.7, 0.9]
    shapes = [(90, 4, 80), (70, 2, 64), (130, 10, 100), (66, 1, 60), (100, 3, 90)]
    w, engines, embs = _lane_setup(E, FAMILY_LLAVA, rc, shapes, max_seq=192)
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    toks = E.EngineGroup(engines).generate(5, mprobs=probs)
    for i, (emb, (T0, s0, L)) in enumerate(zip(embs, shapes)):
        ref = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=7)
        assert toks[i] == ref.generate(emb, s0, L, 5), f"lane {i}"
        st = engines[i].last_step()
        np.testing.assert_array_equal(st["drop"], ref.records[-1].drop)
        assert close(engines[i].logits(), ref.records[-1].logits)
    for e in reversed(engines):
        e.close()


def test_fp8_lanes_grouped_sweeps(E):
    """fp8 weight storage through the grouped GEMV (tiles expanded once, used for every operand plane): 11 lanes (fused
    16-row base pass, 4 + 4 + 2 + 1 member sweeps) against the oracle on the dequantised weights, and a mid-size shape
    whose k loops run full length bit for bit against solo runs."""
    from dropoutdecoding_amd.lm import dequantize_fp8, quantize_fp8
    rc = RefCfg(512, 512, 512, 2, 4, 2, 128, 1e-5, 1000000.0)
    w = random_weights(rc, 33, 0.05)
    wq = {k: (dequantize_fp8(*quantize_fp8(v)) if v.dim() == 2 and "embed_tokens" not in k else v) for k, v in w.items()}
    probs = [0.2, 0.4, 0.6, 0.8]
    shapes = [(60 + 5 * i, 3 + i % 4, 40) for i in range(11)]
    cfg = E.LMConfig(512, 512, 512, 2, 4, 2, 128, 1e-5, 1000000.0)
    engs = []
    for i in range(len(shapes)):
        engs.append(E.DropoutEngine(cfg, family=FAMILY_NEXT, max_seq=192, max_visual=40, seed=4, weight_format="fp8",
                                    share_weights_with=engs[0] if engs else None))
    engs[0].load_state_dict(w)
    embs = [torch.randn(T0, 512, generator=torch.Generator().manual_seed(70 + i)) * 0.8 for i, (T0, _, _) in enumerate(shapes)]
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.prefill(x.cuda(), s0, L)
    toks = E.EngineGroup(engs).generate(6, mprobs=probs)
    for i, (x, (T0, s0, L)) in enumerate(zip(embs, shapes)):
        ref = RefDecoder(FAMILY_NEXT, rc, wq, probs, seed=4)
        assert toks[i] == ref.generate(x, s0, L, 6), f"lane {i}"
        assert close(engs[i].logits(), ref.records[-1].logits)
    for e in reversed(engs):
        e.close()
    # full-length k loops: d = 2048, d_ff = 5632 (22 64-k steps per wave + tail), 5 lanes, synthetic fp8 weights
    big = E.LMConfig(4099, 2048, 5632, 2, 16, 16, 128, 1e-5, 10000.0)
    shapes = [(70, 3, 60), (80, 5, 64), (66, 1, 60), (90, 9, 70), (75, 2, 64)]
    engs = []
    for i in range(len(shapes)):
        engs.append(E.DropoutEngine(big, family=FAMILY_LLAVA, max_seq=128, max_visual=70, seed=9, weight_format="fp8",
                                    share_weights_with=engs[0] if engs else None))
    engs[0].load_synthetic(3, 0.03)
    embs = [torch.randn(T0, 2048, generator=torch.Generator().manual_seed(90 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.prefill(x, s0, L)
    grp = E.EngineGroup(engs)
    rec = [[] for _ in engs]
    for s in range(3):
        grp.decode_step(probs)
        for i, e in enumerate(engs):
            rec[i].append((e.last_step()["drop"].copy(), e.logits().copy(), e.base_logits().copy()))
    toks = [e.tokens() for e in engs]
    for i, (e, x, (T0, s0, L)) in enumerate(zip(engs, embs, shapes)):
        e.rng.manual_seed(9)
        e.prefill(x, s0, L)
        for s in range(3):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.last_step()["drop"], rec[i][s][0])
            np.testing.assert_array_equal(e.logits(), rec[i][s][1], err_msg=f"lane {i} step {s}")
            np.testing.assert_array_equal(e.base_logits(), rec[i][s][2], err_msg=f"lane {i} step {s}")
        assert e.tokens() == toks[i]
    for e in reversed(engs):
        e.close()


def test_long_generation_across_attention_launch_buckets(E):
    """300 tokens per lane starting at 200 / 240 / 250 / 490 keys: the grouped step is re-captured whenever a lane's key-tile
    bucket (4 tiles = 256 keys) changes, lanes cross at different steps.  Tokens against the oracle, logits bit for bit
    against solo runs (which cross the same buckets through dd_lm_decode_step's own graph cache)."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    probs = [0.2, 0.5, 0.8]
    shapes = [(200, 3, 150), (240, 5, 200), (250, 1, 240), (490, 7, 400)]
    w, engines, embs = _lane_setup(E, FAMILY_LLAVA, rc, shapes, max_seq=832)
    n_new = 300
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    toks = E.EngineGroup(engines).generate(n_new, mprobs=probs)
    finals = [e.logits().copy() for e in engines]
    sums = [e.kv_sums().copy() for e in engines]
    for i, (e, emb, (T0, s0, L)) in enumerate(zip(engines, embs, shapes)):
        assert len(toks[i]) == n_new
        if i in (0, 3):                                  # the oracle is slow at these lengths: two lanes are enough
            assert toks[i] == RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=7).generate(emb, s0, L, n_new), f"lane {i}"
        e.rng.manual_seed(7)
        e.prefill(emb.cuda(), s0, L)
        assert e.generate(n_new, mprobs=probs) == toks[i], f"lane {i} solo"
        np.testing.assert_array_equal(e.logits(), finals[i])
        np.testing.assert_array_equal(e.kv_sums(), sums[i])
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("rows_path", [1, 0])
@pytest.mark.parametrize("family,s0", [(FAMILY_LLAVA, 4), (FAMILY_NEXT, 2)])
def test_truncate_and_extend_equal_a_full_prefill(E, family, s0, rows_path):
    """dd_lm_truncate + dd_lm_prefill_extend (several questions about one image): prefill(prefix) then extend(tail) gives
    bit for bit the logits, first token, KV cache and following ensemble steps of prefill(prefix ‖ tail); a second tail
    after truncating back reuses the same prefix; the oracle decodes the full prompts."""
    from dropoutdecoding_amd import _lib
    _lib.load().dd_set_tuning(11, rows_path)                 # 1: chunks of <= 32 rows go through the decode GEMVs (default)
    same = np.testing.assert_array_equal if not rows_path else (
        lambda a, b, err_msg="": np.testing.assert_allclose(a, b, rtol=0, atol=2e-5 * float(np.abs(b).max()), err_msg=err_msg))
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 41, 0.05)
    cfg = E.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    probs = [0.3, 0.5, 0.7]
    L, P = 70, s0 + 70                                       # the prefix ends with the visual span (crosses a 64-key tile)
    g = torch.Generator().manual_seed(12)
    prefix = torch.randn(P, 256, generator=g) * 0.8
    tails = [torch.randn(n, 256, generator=g) * 0.8 for n in (9, 1, 23, 40)]     # 8- / 16- / 32-row kernels, and the GEMM path
    full = E.DropoutEngine(cfg, family=family, max_seq=256, max_visual=L, seed=3)
    part = E.DropoutEngine(cfg, family=family, max_seq=256, max_visual=L, seed=3)
    full.load_state_dict(w)
    part.load_state_dict(w)
    part.prefill(prefix.cuda(), s0, L)
    epi0 = part.vision_uncert_dict()["epis_uncert_per_token"].copy()
    ref_rng = None
    for ti, tail in enumerate(tails):
        prompt = torch.cat([prefix, tail], 0)
        full.prefill(prompt.cuda(), s0, L)
        part.truncate(P)
        part.prefill_extend(tail.cuda())
        (np.testing.assert_array_equal if tail.shape[0] > 32 else same)(part.logits(), full.logits(), err_msg=f"tail {ti}")
        np.testing.assert_array_equal(part.vision_uncert_dict()["epis_uncert_per_token"], epi0)
        ref = RefDecoder(family, rc, w, probs, seed=3)
        if ref_rng is not None:
            ref.rng = ref_rng                                 # the stream continues across prompts, as on the engines
        want = ref.generate(prompt, s0, L, 5)
        ref_rng = ref.rng
        for s in range(4):
            full.decode_step(probs)
            part.decode_step(probs)
            np.testing.assert_array_equal(part.last_step()["drop"], full.last_step()["drop"])
            (np.testing.assert_array_equal if tail.shape[0] > 32 else same)(part.logits(), full.logits(), err_msg=f"tail {ti} step {s}")
        assert part.tokens() == full.tokens() == want
        np.testing.assert_allclose(part.kv_sums(), full.kv_sums(), rtol=0, atol=0 if (not rows_path or tail.shape[0] > 32) else 1e-3)
    with pytest.raises(ValueError):
        part.truncate(P - 1)                                  # would cut into the visual span
    with pytest.raises(ValueError):
        part.prefill_extend(tails[0].cuda())                  # tokens already generated: truncate first
    _lib.load().dd_set_tuning(11, 1)
    full.close()
    part.close()


@pytest.mark.parametrize("kw,K", [(dict(), 12), (dict(use_avg=True), 4), (dict(mask_method="epis_no_overlap"), 3)])
def test_group_step_side_paths(E, kw, K):
    """Group steps off the main road: K > 8 (members fall back to per-sequence packed sweeps), select_by_average (commit per
    sequence), the no-overlap mask rule in the batched mask kernel — each lane against the oracle and a solo run."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    probs = [0.1 + 0.06 * i for i in range(K)]
    shapes = [(40, 3, 30), (52, 5, 40), (47, 2, 40)]
    w, engines, embs = _lane_setup(E, FAMILY_LLAVA, rc, shapes, **kw)
    for e, emb, (T0, s0, L) in zip(engines, embs, shapes):
        e.prefill(emb.cuda(), s0, L)
    toks = E.EngineGroup(engines).generate(5, mprobs=probs)
    finals = [e.logits().copy() for e in engines]
    for i, (e, emb, (T0, s0, L)) in enumerate(zip(engines, embs, shapes)):
        ref = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=7, **kw)
        assert toks[i] == ref.generate(emb, s0, L, 5), f"lane {i}"
        assert close(finals[i], ref.records[-1].logits)
        e.rng.manual_seed(7)
        e.prefill(emb.cuda(), s0, L)
        assert e.generate(5, mprobs=probs) == toks[i]
        np.testing.assert_array_equal(e.logits(), finals[i])
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("dims,T0", [((512, 256, 512, 2, 2, 2), 1100),          # 16-tile matrices: part of one 32-tile column block
                                     ((2048, 1024, 2816, 2, 8, 4), 1300)])      # several column blocks, K = 2816 (88 k-steps), GQA 2
def test_prefill_gemm_block_shapes_and_orders_give_the_same_bits(E, dims, T0):
    """The prefill GEMM has three forms — 128 x 128 blocks in row-major block order, the same in the XCD-aware order, and the
    128 x 512 LDS-staged block used from 1024 rows on (k_gemm_big) — that issue the same MFMA sequence per accumulator tile:
    image logits, last-row logits, the KV cache and the following ensemble steps must be bit-identical, and agree with the oracle."""
    from dropoutdecoding_amd import _lib
    lib = _lib.load()
    V, d, dff, nl, H, Hkv = dims
    rc = RefCfg(V, d, dff, nl, H, Hkv, 128, 1e-5, 10000.0)
    w = random_weights(rc, 77, 0.05)
    cfg = E.LMConfig(V, d, dff, nl, H, Hkv, 128, 1e-5, 10000.0)
    L, s0 = 96, 3
    probs = [0.3, 0.5, 0.7]
    x = torch.randn(T0, d, generator=torch.Generator().manual_seed(5)) * 0.8
    e = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=T0 + 64, max_visual=L, seed=3)
    e.load_state_dict(w)
    outs = []
    try:
        for order, big in ((0, 0), (1, 0), (0, 1024), (1, 1024), (1, 128)):
            lib.dd_set_tuning(15, order)
            lib.dd_set_tuning(16, big)
            e.rng.manual_seed(3)
            e.prefill(x.cuda(), s0, L)
            rec = [e.image_logits().copy(), e.logits().copy(), e.kv_sums().copy()]
            for _ in range(2):
                e.decode_step(probs)
                rec.append(e.logits().copy())
            rec.append(np.asarray(e.tokens()))
            outs.append(rec)
    finally:
        lib.dd_set_tuning(15, 1)
        lib.dd_set_tuning(16, 1024)
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            np.testing.assert_array_equal(a, b)
    ref = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=3)
    assert ref.generate(x, s0, L, 3) == outs[0][-1].tolist()
    assert close(outs[0][-2], ref.records[-1].logits)
    e.close()


@pytest.mark.parametrize("family,wfmt", [(FAMILY_LLAVA, "bf16"), (FAMILY_IBLIP, "bf16"), (FAMILY_LLAVA, "fp16")])
def test_prefill_group_equals_one_prefill_per_sequence(E, family, wfmt):
    """dd_lm_prefill_group: the prompts of several lanes through the layers as ONE matrix (each sequence padded to whole 128-row
    blocks, the QKV epilogue writing each row to its own sequence's cache).  Every lane must end up bit for bit as its own
    prefill() leaves it — image logits, scores, top-k ids, first token, KV cache — and decode the same tokens afterwards;
    ragged lengths, spans at different offsets, a batch small enough to fall back to single prefills."""
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    shapes = [(300, 3, 200), (257, 0, 32), (384, 9, 200), (129, 1, 100), (290, 5, 64)]       # 5 x 384 rows = 1920 >= 1024: batched
    if family == FAMILY_IBLIP:
        shapes = [(T0, 0, min(L, 32)) for T0, _, L in shapes]
    probs = [0.3, 0.5, 0.7]
    w, engines, embs = _lane_setup(E, family, rc, shapes, max_seq=448, weight_format=wfmt, kv_format="fp16" if wfmt == "fp16" else "fp32")
    embs = [x.cuda() for x in embs]
    solo = []
    for e, x, (T0, s0, L) in zip(engines, embs, shapes):
        e.rng.manual_seed(7)
        e.prefill(x, s0, L)
        rec = [e.image_logits().copy(), e.logits().copy(), e.kv_sums().copy(), e.vision_uncert_dict()["epis_uncert_per_token"].copy(),
               np.asarray(e.topk()[1]).copy()]
        for _ in range(3):
            e.decode_step(probs)
        rec += [e.logits().copy(), np.asarray(e.tokens())]
        solo.append(rec)
    for sub in (slice(0, 5), slice(1, 3)):                   # 5 sequences: one batch; 2 short ones: 2 x 384 < 1024 rows -> fallback
        ee, xx, ss = engines[sub], embs[sub], shapes[sub]
        for e in ee:
            e.rng.manual_seed(7)
        E.prefill_group(ee, xx, [(s0, L) for _, s0, L in ss])
        for e, want in zip(ee, solo[sub]):
            got = [e.image_logits().copy(), e.logits().copy(), e.kv_sums().copy(), e.vision_uncert_dict()["epis_uncert_per_token"].copy(),
                   np.asarray(e.topk()[1]).copy()]
            for _ in range(3):
                e.decode_step(probs)
            got += [e.logits().copy(), np.asarray(e.tokens())]
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b)
    with pytest.raises(Exception):
        E.prefill_group(engines[:2], embs[:2], [(0, 10), (250, 200)])         # span does not fit
    with pytest.raises(ValueError):
        E.prefill_group(engines[:2], embs[:1], [(0, 10)])
    for e in reversed(engines):
        e.close()
