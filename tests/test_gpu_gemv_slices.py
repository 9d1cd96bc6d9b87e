"""The slice-resident 16 / 32 / 64-row decode GEMVs (csrc/dd_gemv_slices.h + k_gemv_finish) against the K-split-over-waves
kernels they replace on the lanes path: same k order, same MFMA chains, same reduction order, so every logit, token and
KV row must be BIT-identical — and with it everything the solo 8-row kernel produces for the same sequence.
Shapes are the 7B families' (K = 4096, 11008, 14336), two layers deep so the test stays small."""
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


def _group(E, cfg, n, family, L, seed=3, **kw):
    engines = []
    for i in range(n):
        e = E.DropoutEngine(cfg, family=family, max_seq=L + 96, max_visual=L, seed=50 + i,
                            share_weights_with=engines[0] if engines else None, **kw)
        engines.append(e)
    engines[0].load_synthetic(seed=seed, std=0.02)
    return engines


def _run(E, engines, embs, spans, probs, steps, slices, graph):
    lib = engines[0].lib
    lib.dd_set_tuning(13, 1 if slices else 0)
    lib.dd_set_tuning(8, 1 if graph else 0)
    for i, (e, emb, (s0, L)) in enumerate(zip(engines, embs, spans)):
        e.rng.manual_seed(50 + i)
        e.prefill(emb, s0, L)
    grp = E.EngineGroup(engines)
    out = []
    for _ in range(steps):
        grp.decode_step(probs)
        out.append([(e.logits().copy(), e.base_logits().copy()) for e in engines])
    toks = [e.tokens() for e in engines]
    sums = [e.kv_sums().copy() for e in engines]
    lib.dd_set_tuning(13, 1)
    lib.dd_set_tuning(8, 1)
    return out, toks, sums


@pytest.mark.parametrize("name,dims,n_lanes", [
    ("llama-7b-shapes, 4 lanes: 32 member rows", (4096, 11008, 32, 32), 4),
    ("llama-7b-shapes, 2 lanes: 16 member rows", (4096, 11008, 32, 32), 2),
    ("llama-7b-shapes, 20 lanes: base rows in four planes", (4096, 11008, 32, 32), 20),
    ("mistral-7b-shapes (GQA 4, d_ff 14336), 5 lanes", (4096, 14336, 32, 8), 5),
    ("llama-7b-shapes, 9 lanes: one 64-row member pass (eight operand planes) + one 8-row pass", (4096, 11008, 32, 32), 9),
    ("mistral-7b-shapes, 8 lanes: 64 member rows, K = 14336 in chunks", (4096, 14336, 32, 8), 8),
])
def test_slice_kernels_bit_identical_to_wave_split_kernels(E, name, dims, n_lanes):
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    L = 24
    engines = _group(E, cfg, n_lanes, "llava-1.5", L)
    gen = torch.Generator().manual_seed(9)
    T0s = [L + 6 + (i % 5) for i in range(n_lanes)]
    embs = [(torch.randn(T0, d, generator=gen) * 0.5).cuda() for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    ref, rtoks, rsums = _run(E, engines, embs, spans, probs, 3, slices=False, graph=False)
    for graph in (False, True):
        got, gtoks, gsums = _run(E, engines, embs, spans, probs, 3, slices=True, graph=graph)
        assert gtoks == rtoks, name
        for s in range(3):
            for i in range(n_lanes):
                np.testing.assert_array_equal(got[s][i][0], ref[s][i][0], err_msg=f"{name}: member logits, step {s} lane {i}")
                np.testing.assert_array_equal(got[s][i][1], ref[s][i][1], err_msg=f"{name}: base logits, step {s} lane {i}")
        for a, b in zip(gsums, rsums):
            np.testing.assert_array_equal(a, b)
    # and a lane decoded alone (8-row kernels) gives the same bits as in the group
    e = engines[1]
    e.rng.manual_seed(51)
    e.prefill(embs[1], *spans[1])
    for s in range(3):
        e.decode_step(probs)
        np.testing.assert_array_equal(e.logits(), ref[s][1][0])
    assert e.tokens() == rtoks[1]
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("name,dff,n_lanes", [("fp8, mistral-7b shapes, 2 lanes: 16 rows, K = 14336 through the slice kernel too", 14336, 2),
                                              ("fp8, mistral-7b shapes, 4 lanes: 32 rows", 14336, 4),
                                              ("fp8, mistral-7b shapes, 9 lanes: one 64-row pass + one 8-row pass", 14336, 9),
                                              ("fp8, llama-7b shapes, 4 lanes (K = 11008 stays on the wave-split kernel)", 11008, 4)])
def test_fp8_slice_kernels_bit_identical_to_wave_split_kernels(E, name, dff, n_lanes):
    """BASELINE config 5's weight format on the slice-resident kernels (k_gemv_slices_fp8: the fp8 -> bf16 expansion done once per
    weight load for all operand planes; row scales in the finishing kernel): every logit and KV row as the wave-split fp8 kernels
    and the 8-row fp8 kernel produce them."""
    cfg = E.LMConfig(2048, 4096, dff, 2, 32, 8 if dff == 14336 else 32, 128, 1e-5, 10000.0)
    L = 24
    engines = _group(E, cfg, n_lanes, "llava-next", L, weight_format="fp8", kv_format="fp16")
    gen = torch.Generator().manual_seed(9)
    T0s = [L + 6 + (i % 5) for i in range(n_lanes)]
    embs = [(torch.randn(T0, 4096, generator=gen) * 0.5).cuda() for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    ref, rtoks, rsums = _run(E, engines, embs, spans, probs, 3, slices=False, graph=False)
    got, gtoks, gsums = _run(E, engines, embs, spans, probs, 3, slices=True, graph=True)
    assert gtoks == rtoks, name
    for s in range(3):
        for i in range(n_lanes):
            np.testing.assert_array_equal(got[s][i][0], ref[s][i][0], err_msg=f"{name}: member logits, step {s} lane {i}")
            np.testing.assert_array_equal(got[s][i][1], ref[s][i][1], err_msg=f"{name}: base logits, step {s} lane {i}")
    for a, b in zip(gsums, rsums):
        np.testing.assert_array_equal(a, b)
    e = engines[1]                                         # alone: two-sweep 8-row kernels, then the speculative 16-row step
    for mode in ("never", "always"):
        e.set_speculation(mode)
        e.rng.manual_seed(51)
        e.prefill(embs[1], *spans[1])
        for s in range(3):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.logits(), ref[s][1][0], err_msg=f"{name}: solo ({mode}), step {s}")
        assert e.tokens() == rtoks[1]
    e.set_speculation("default")
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("n_lanes", [2, 4])
def test_progressive_stage_in_leaves_the_same_bits(E, n_lanes):
    """the progressive stage-in of the operand planes (dd_gemv_slices.h PROG, tools key 49: the default at two and four
    planes, where it pays 3 %) is claimed to give the same bits as the blocking stage-in it replaced — here both run: 2 lanes (16 member
    rows, NG = 2) and 4 lanes (32 rows, NG = 4), LLaMA-7B widths, every logit, token and KV checksum equal."""
    from dropoutdecoding_amd import _lib
    T = _lib.load_tools()
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L = 24
    engines = _group(E, cfg, n_lanes, "llava-1.5", L, lib=T)
    gen = torch.Generator().manual_seed(21)
    embs = [(torch.randn(L + 6 + i, d, generator=gen) * 0.5).cuda() for i in range(n_lanes)]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    outs = []
    try:
        for prog in (1, 0, 1):
            T.dd_tools_set_tuning(49, prog)
            outs.append(_run(E, engines, embs, spans, probs, 4, slices=True, graph=False))
    finally:
        T.dd_tools_set_tuning(49, 1)
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
