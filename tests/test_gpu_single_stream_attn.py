"""Round 6, two kernel forms that must not change a bit: the decode attention's merge launches request the key tiles' outputs with their
first loads (attn_combine_core PRE; dd_tools_set_tuning key 55 = 0 restores the loads behind the two barriers) and the 8-row qkv GEMV
walks its tile groups in a loop with the rows' operand in registers (k_gemv_loop; key 54 = 0 restores k_gemv).  The same MFMA chains and
the same sums in the same order, so every logit, mask, token, KV row and the rng stream must be BIT-identical to the round-5 forms.
Reference anchors: the attention inside the LM forward the reference calls at models/llava.py:294-303, 350-359 (mask: 346-349).
LLaMA-7B / Mistral-7B widths, two layers; one sequence on its own (contexts that start inside a key tile, cross tile boundaries, cross a
multiple of four tiles = a new launch shape, and exceed the sixteen tiles the early loads cover) and sixteen lanes in the rider form (the
merge of 64 member rows + riding rows in one launch).  The oracle comparison of the default path is the rest of the suite.
(The third form tried this round — one sequence's whole attention in one launch, a wave per key tile — was bit-identical too and
slower: profiles/r06_lab/attn_one_launch_ab.log; it is not in the tree.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _run(eng, T, emb, s0, L, probs, steps, knobs, spec, graph, seed=11):
    lib = eng.lib
    for k, v in knobs.items():
        T.dd_tools_set_tuning(k, v)
    lib.dd_set_tuning(14, spec)
    lib.dd_set_tuning(8, 1 if graph else 0)
    try:
        eng.rng.manual_seed(seed)
        eng.prefill(emb, s0, L)
        recs = []
        for _ in range(steps):
            eng.decode_step(probs)
            st = eng.last_step()
            recs.append((st["drop"].copy(), st["masked_numbers"].tolist(), st["member_argmax"].tolist(), st["winner"], st["keep"].copy(),
                         eng.logits().copy(), eng.base_logits().copy()))
        return recs, eng.tokens(), eng.kv_sums().copy(), eng.rng.rand(32).cpu().numpy()
    finally:
        for k in knobs:
            T.dd_tools_set_tuning(k, 1)
        lib.dd_set_tuning(14, 2)
        lib.dd_set_tuning(8, 1)


def _same(got, ref, what):
    assert got[1] == ref[1], what
    for s, (a, b) in enumerate(zip(got[0], ref[0])):
        np.testing.assert_array_equal(a[0], b[0], err_msg=f"{what}: drop masks, step {s}")
        assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3], f"{what}: step {s}"
        np.testing.assert_array_equal(a[4], b[4], err_msg=f"{what}: keep set, step {s}")
        np.testing.assert_array_equal(a[5], b[5], err_msg=f"{what}: winner logits, step {s}")
        np.testing.assert_array_equal(a[6], b[6], err_msg=f"{what}: base logits, step {s}")
    np.testing.assert_array_equal(got[2], ref[2], err_msg=f"{what}: KV checksums")
    np.testing.assert_array_equal(got[3], ref[3], err_msg=f"{what}: rng stream")


@pytest.mark.parametrize("name,dims,family,K,T0,L,steps", [
    ("llama-7b widths, one partly filled tile", (4096, 11008, 32, 32), "llava-1.5", 8, 41, 24, 4),
    ("llama-7b widths, 250 -> 262 keys: the fourth tile fills, a fifth opens (new launch shape)", (4096, 11008, 32, 32), "llava-1.5", 8, 250, 200, 12),
    ("llama-7b widths, K = 3, 700 keys: eleven tiles", (4096, 11008, 32, 32), "llava-1.5", 3, 700, 576, 5),
    ("llama-7b widths, 1020 -> 1030 keys: past the sixteen tiles the early loads cover", (4096, 11008, 32, 32), "llava-1.5", 8, 1020, 576, 10),
    ("mistral-7b widths (GQA 4), 130 keys", (4096, 14336, 32, 8), "llava-next", 4, 130, 96, 5),
])
def test_early_merge_loads_and_looping_qkv_gemv_leave_the_same_bits(E, name, dims, family, K, T0, L, steps):
    from dropoutdecoding_amd import _lib
    T = _lib.load_tools()
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    eng = E.DropoutEngine(cfg, family=family, max_seq=T0 + 64, max_visual=L, seed=11, lib=T)
    eng.load_synthetic(seed=5, std=0.02)
    emb = (torch.randn(T0, d, generator=torch.Generator().manual_seed(3)) * 0.5).cuda()
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]
    s0 = 3
    try:
        # the round-5 forms: merge launches with late loads, k_gemv for qkv
        ref = _run(eng, T, emb, s0, L, probs, steps, {54: 0, 55: 0}, spec=0, graph=False)
        for knobs in ({54: 0, 55: 1}, {54: 1, 55: 0}, {54: 1, 55: 1}):
            for spec, graph in ((0, False), (0, True), (1, True)):
                got = _run(eng, T, emb, s0, L, probs, steps, knobs, spec=spec, graph=graph)
                _same(got, ref, f"{name}: knobs {knobs}, spec {spec}, graph {graph}")
    finally:
        eng.close()


def test_rider_merge_with_early_loads_leaves_the_same_bits(E):
    """Sixteen lanes, K = 8, LLaMA-7B widths: the rider sweeps' merge launch (k_attn_combine_ride: 64 member rows + the riding rows) with
    the early loads against the late loads — logits of every lane, tokens and KV checksums equal."""
    from dropoutdecoding_amd import _lib
    T = _lib.load_tools()
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L, n = 24, 16
    engines = []
    for i in range(n):
        engines.append(E.DropoutEngine(cfg, family="llava-1.5", max_seq=L + 96, max_visual=L, seed=50 + i,
                                       share_weights_with=engines[0] if engines else None, lib=T))
    engines[0].load_synthetic(seed=3, std=0.02)
    gen = torch.Generator().manual_seed(21)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [(2 + (i % 3), L) for i in range(n)]
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    outs = []
    try:
        for pre in (0, 1, 0, 1):
            T.dd_tools_set_tuning(55, pre)
            for i, (e, emb, (s0, Lv)) in enumerate(zip(engines, embs, spans)):
                e.rng.manual_seed(50 + i)
                e.prefill(emb, s0, Lv)
            grp = E.EngineGroup(engines)
            logits = []
            for _ in range(4):
                grp.decode_step(probs)
                logits.append([(e.logits().copy(), e.base_logits().copy()) for e in engines])
            outs.append((logits, [e.tokens() for e in engines], [e.kv_sums().copy() for e in engines]))
    finally:
        T.dd_tools_set_tuning(55, 1)
    for o in outs[1:]:
        assert o[1] == outs[0][1]
        for sa, sb in zip(o[0], outs[0][0]):
            for (la, ba), (lb, bb) in zip(sa, sb):
                np.testing.assert_array_equal(la, lb)
                np.testing.assert_array_equal(ba, bb)
        for x, y in zip(o[2], outs[0][2]):
            np.testing.assert_array_equal(x, y)
    for e in reversed(engines):
        e.close()
