"""K <= 4 (the reference's own settings: voting_numbers [0.1, 0.3, 0.5, 0.7], [0.3, 0.5, 0.7], ... — chair_test/chair_test.py:165-170,
models/config.py:2): the members of a sequence fill half an operand plane, so a 64-row member sweep carries SIXTEEN sequences, two per
plane (csrc/dd_engine.hip lm_sweep_groups `packed`, GemvArgs / AttnDecodeArgs `half_planes`).  Tokens, logits, masks, KV checksums and
rng streams must be bit-identical to the one-sequence-per-plane sweeps and to every sequence decoded alone.  Whole groups of fourteen (28,
42, 56 lanes) take the rider form on top: seven half planes + TWO riding planes with their partners' un-masked rows.  7B-family shapes,
two layers."""
import numpy as np
import pytest
import torch

from test_gpu_rider import _group, _same

pytestmark = pytest.mark.gpu

K4 = [0.1, 0.3, 0.5, 0.7]
K3 = [0.3, 0.5, 0.7]


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


@pytest.fixture(scope="module")
def T():
    from dropoutdecoding_amd import _lib
    return _lib.load_tools()


def _run(E, T, engines, embs, spans, probs, steps, half, graph, eos=None):
    T.dd_tools_set_tuning(30, 1 if half else 0)
    T.dd_tools_set_tuning(26, 1 if half else 0)  # reference run: the classic form, one sequence per plane; half planes: whole groups of
                                                 # fourteen take the rider form (seven half planes + two riding planes), other line-ups the classic one
    T.dd_tools_set_tuning(8, 1 if graph else 0)
    for i, (e, emb, (s0, L)) in enumerate(zip(engines, embs, spans)):
        e.rng.manual_seed(50 + i)
        e.prefill(emb, s0, L)
        e.set_eos(eos if eos is not None else [])
    grp = E.EngineGroup(engines)
    out = []
    for s in range(steps):
        grp.decode_step(probs)
        rec = []
        for e in engines:
            st = e.last_step()
            rec.append((e.logits().copy(), e.base_logits().copy(), st["drop"].copy(), st["keep"].copy()))
        out.append(rec)
    toks = [e.tokens() for e in engines]
    sums = [e.kv_sums().copy() for e in engines]
    tails = [e.rng.rand(8).cpu().numpy().copy() for e in engines]
    T.dd_tools_set_tuning(30, 1)
    T.dd_tools_set_tuning(26, 1)
    T.dd_tools_set_tuning(8, 1)
    return out, toks, sums, tails


@pytest.mark.parametrize("name,family,dims,n_lanes,probs,kw", [
    ("llama-7b shapes, 16 lanes, K = 4: one sweep of sixteen", "llava-1.5", (4096, 11008, 32, 32), 16, K4, {}),
    ("llama-7b shapes, 32 lanes, K = 3 (one dead row per half plane): two sweeps on two branches", "llava-1.5", (4096, 11008, 32, 32), 32, K3, {}),
    ("llama-7b shapes, 27 lanes, K = 4: sixteen + eight + a pair + one", "llava-1.5", (4096, 11008, 32, 32), 27, K4, {}),
    ("mistral-7b shapes (GQA 4), 16 lanes, K = 4, LLaVA-NeXT rule", "llava-next", (4096, 14336, 32, 8), 16, K4, {}),
    ("InstructBLIP rule (vote on the hidden rows of a half plane, leaked zeros), 16 lanes, K = 4", "instructblip", (4096, 11008, 32, 32), 16, K4, {}),
    ("llama-7b shapes, 16 lanes, K = 2, Philox stream", "llava-1.5", (4096, 11008, 32, 32), 16, [0.5, 0.3], {"rng_stream": "gpu"}),
    ("rider form, 28 lanes, K = 4: one ring of two groups of fourteen", "llava-1.5", (4096, 11008, 32, 32), 28, K4, {}),
    ("rider form, 56 lanes, K = 3: two branches, rings of two", "llava-1.5", (4096, 11008, 32, 32), 56, K3, {}),
    ("rider form, 42 lanes, K = 4: one ring of three, GQA, LLaVA-NeXT rule", "llava-next", (4096, 14336, 32, 8), 42, K4, {}),
    ("rider form, 28 lanes, K = 4, InstructBLIP rule with masked positions", "instructblip", (4096, 11008, 32, 32), 28, K4, {"iblip_positions": "mask"}),
])
def test_half_plane_sweeps_equal_plain_sweeps_and_solo_runs(E, T, name, family, dims, n_lanes, probs, kw):
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    L, steps = 24, 5
    engines = _group(E, T, cfg, n_lanes, family, L, **kw)
    gen = torch.Generator().manual_seed(21)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n_lanes)]
    spans = [((0 if family == "instructblip" else 2 + (i % 3)), L) for i in range(n_lanes)]
    ref = _run(E, T, engines, embs, spans, probs, steps, half=False, graph=False)
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, probs, steps, half=True, graph=graph)
        _same(got, ref, f"{name} (graph {graph})")
    for li in (1, 8, 15, n_lanes - 1):      # second sequence of a plane, first of another, a rider of the second riding plane, the last lane
        e = engines[li]
        e.set_speculation("never")
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        for s in range(steps):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.logits(), ref[0][s][li][0], err_msg=f"{name}: solo lane {li} step {s}")
        assert e.tokens() == ref[1][li]
        e.set_speculation("default")
    for e in reversed(engines):
        e.close()


def test_half_plane_sweeps_with_sequences_that_end(E, T):
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L, n = 24, 32
    engines = _group(E, T, cfg, n, "llava-1.5", L)
    gen = torch.Generator().manual_seed(23)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [(2 + (i % 3), L) for i in range(n)]
    plain = _run(E, T, engines, embs, spans, K4, 6, half=False, graph=False)
    eos = sorted({plain[1][3][2], plain[1][18][3]})
    ref = _run(E, T, engines, embs, spans, K4, 6, half=False, graph=False, eos=eos)
    assert any(len(t) < 7 for t in ref[1])
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, K4, 6, half=True, graph=graph, eos=eos)
        _same(got, ref, f"EOS run (graph {graph})")
    for e in reversed(engines):
        e.close()


def test_generate_keeps_groups_of_fourteen_while_sequences_end(E, T):
    """EngineGroup.generate at K = 4 with EOS ids and 28 lanes: ended sequences stay in the line-up while they fill the last group of
    fourteen; tokens and rng streams as with one sequence per plane in the classic form, and as each sequence alone."""
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L, n, n_new = 24, 28, 9
    engines = _group(E, T, cfg, n, "llava-1.5", L)
    gen = torch.Generator().manual_seed(29)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [(2 + (i % 3), L) for i in range(n)]

    def run(half, eos):
        T.dd_tools_set_tuning(30, 1 if half else 0)
        T.dd_tools_set_tuning(26, 1 if half else 0)
        for i, e in enumerate(engines):
            e.rng.manual_seed(50 + i)
            e.prefill(embs[i], *spans[i])
        toks = E.EngineGroup(engines).generate(n_new, eos=eos, mprobs=K4, lookahead=3)
        tails = [e.rng.rand(8).cpu().numpy().copy() for e in engines]
        T.dd_tools_set_tuning(30, 1)
        T.dd_tools_set_tuning(26, 1)
        return toks, tails

    free, _ = run(False, None)
    eos = sorted({free[2][2], free[15][4], free[20][6]})
    want, wtails = run(False, eos)
    lens = sorted(len(t) for t in want)
    assert lens[0] < n_new and lens[-1] == n_new, lens
    got, gtails = run(True, eos)
    assert got == want
    for a, b in zip(gtails, wtails):
        np.testing.assert_array_equal(a, b)
    for li in (2, 15, 27):
        e = engines[li]
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        assert e.generate(n_new, eos=eos, mprobs=K4) == want[li], f"lane {li} alone"
    for e in reversed(engines):
        e.close()
