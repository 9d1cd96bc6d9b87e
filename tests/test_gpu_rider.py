"""The rider form of the group step (csrc/dd_engine.hip group_step_rider): with 16, 24, 32 ... sequences in groups of eight the
un-masked rows need no sweep of their own — they ride in a ninth operand plane of another group's member sweep (72-row kernels,
csrc/dd_gemv.hip try_slices9), the ring leaders' rows a step ahead.  Everything a step produces — tokens, member / un-masked logits,
keep sets and masks, KV checksums, rng streams — must be BIT-identical to the classic group step (one fused un-masked pass + the
member sweeps) and to every sequence decoded alone.  7B-family shapes (the nine-plane kernels exist for those), two layers deep.
The switch between the two forms is an experiment knob (tools key 26), so the engines live in libdropdec_tools.so here."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

K8 = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
K4 = [0.1, 0.3, 0.5, 0.7]


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


def _group(E, T, cfg, n, family, L, **kw):
    engines = []
    for i in range(n):
        engines.append(E.DropoutEngine(cfg, family=family, max_seq=L + 96, max_visual=L, seed=50 + i, kv_format="fp16", lib=T,
                                       share_weights_with=engines[0] if engines else None, **kw))
    engines[0].load_synthetic(seed=3, std=0.02)
    return engines


def _run(E, T, engines, embs, spans, probs, steps, rider, graph, eos=None, read_every_step=True):
    T.dd_tools_set_tuning(26, 1 if rider else 0)
    T.dd_tools_set_tuning(30, 0)                # (K <= 4 would otherwise take the half-plane form: tests/test_gpu_half_planes.py)
    T.dd_tools_set_tuning(8, 1 if graph else 0)
    for i, (e, emb, (s0, L)) in enumerate(zip(engines, embs, spans)):
        e.rng.manual_seed(50 + i)
        e.prefill(emb, s0, L)
        e.set_eos(eos if eos is not None else [])
    grp = E.EngineGroup(engines)
    out = []
    for s in range(steps):
        grp.decode_step(probs)
        if read_every_step or s == steps - 1:
            rec = []
            for e in engines:
                st = e.last_step()
                rec.append((e.logits().copy(), e.base_logits().copy(), st["drop"].copy(), st["keep"].copy()))
            out.append(rec)
    toks = [e.tokens() for e in engines]
    sums = [e.kv_sums().copy() for e in engines]
    tails = [e.rng.rand(8).cpu().numpy().copy() for e in engines]
    T.dd_tools_set_tuning(26, 1)
    T.dd_tools_set_tuning(30, 1)
    T.dd_tools_set_tuning(8, 1)
    return out, toks, sums, tails


def _same(a, b, what):
    assert a[1] == b[1], f"{what}: tokens"
    assert len(a[0]) == len(b[0])
    for s, (ra, rb) in enumerate(zip(a[0], b[0])):
        for i, (x, y) in enumerate(zip(ra, rb)):
            for j, nm in enumerate(("member logits", "un-masked logits", "masks", "keep set")):
                np.testing.assert_array_equal(x[j], y[j], err_msg=f"{what}: {nm}, step {s} lane {i}")
    for i, (x, y) in enumerate(zip(a[2], b[2])):
        np.testing.assert_array_equal(x, y, err_msg=f"{what}: KV checksums, lane {i}")
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        np.testing.assert_array_equal(x, y, err_msg=f"{what}: rng stream after the run, lane {i}")


@pytest.mark.parametrize("n_lanes", [16, 32])
def test_rider_step_fp8_mistral_shapes(E, T, n_lanes):
    """BASELINE config 5's weight format in the rider form: nine operand planes through k_gemv_slices_fp8<9, ...> (K = 4096) and the chunked
    k_gemv_slices_fp8c<9, ...> (down_proj, K = 14336) — bit for bit the classic group step (64-row fp8 passes + a fused un-masked pass)
    and every sequence decoded alone (the 8-row fp8 kernel)."""
    d, dff, H, Hkv = 4096, 14336, 32, 8
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 1000000.0)
    L = 40
    engines = _group(E, T, cfg, n_lanes, "llava-next", L, weight_format="fp8")
    gen = torch.Generator().manual_seed(11)
    T0s = [L + 6 + (i % 5) for i in range(n_lanes)]
    embs = [(torch.randn(T0, d, generator=gen) * 0.5).cuda() for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    steps = 5
    ref = _run(E, T, engines, embs, spans, K8, steps, rider=False, graph=False)
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, K8, steps, rider=True, graph=graph)
        _same(got, ref, f"fp8 rider, {n_lanes} lanes (graph {graph})")
    for li in (0, 9, n_lanes - 1):
        e = engines[li]
        e.set_speculation("never")
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        for s in range(steps):
            e.decode_step(K8)
            np.testing.assert_array_equal(e.logits(), ref[0][s][li][0], err_msg=f"fp8 rider: solo lane {li} step {s}")
            np.testing.assert_array_equal(e.base_logits(), ref[0][s][li][1], err_msg=f"fp8 rider: solo lane {li} step {s} (un-masked)")
        assert e.tokens() == ref[1][li]
        e.set_speculation("default")
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("name,dims,n_lanes,probs", [
    ("llama-7b shapes, 16 lanes: one ring of two groups", (4096, 11008, 32, 32), 16, K8),
    ("llama-7b shapes, 32 lanes: two branches, rings of two", (4096, 11008, 32, 32), 32, K8),
    ("llama-7b shapes, 24 lanes: one ring of three, K = 4 (the riding plane has more live rows than the member planes)", (4096, 11008, 32, 32), 24, K4),
    ("mistral-7b shapes (GQA 4, d_ff 14336), 16 lanes", (4096, 14336, 32, 8), 16, K8),
])
def test_rider_step_equals_classic_group_step_and_solo_runs(E, T, name, dims, n_lanes, probs):
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    L = 24
    engines = _group(E, T, cfg, n_lanes, "llava-1.5", L)
    gen = torch.Generator().manual_seed(9)
    T0s = [L + 6 + (i % 5) for i in range(n_lanes)]
    embs = [(torch.randn(T0, d, generator=gen) * 0.5).cuda() for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    steps = 6
    ref = _run(E, T, engines, embs, spans, probs, steps, rider=False, graph=False)
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, probs, steps, rider=True, graph=graph)
        _same(got, ref, f"{name} (graph {graph})")
    # without a read-back between the steps (the graph replays back to back)
    got = _run(E, T, engines, embs, spans, probs, steps, rider=True, graph=True, read_every_step=False)
    assert got[1] == ref[1]
    for i in range(n_lanes):
        np.testing.assert_array_equal(got[0][-1][i][0], ref[0][-1][i][0])
        np.testing.assert_array_equal(got[2][i], ref[2][i])
    # sequences decoded alone: a ring leader, a rider of the first sweep, the last lane
    for li in (0, 9, n_lanes - 1):
        e = engines[li]
        e.set_speculation("never")
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        for s in range(steps):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.logits(), ref[0][s][li][0], err_msg=f"{name}: solo lane {li} step {s}")
            np.testing.assert_array_equal(e.base_logits(), ref[0][s][li][1], err_msg=f"{name}: solo lane {li} step {s} (un-masked)")
        assert e.tokens() == ref[1][li]
        e.set_speculation("default")
    for e in reversed(engines):
        e.close()


def test_rider_step_with_sequences_that_end_and_line_ups_that_change(E, T):
    """EOS inside the run (finished sequences keep riding, their rows are ignored), a lane re-prefilled between two steps (the ring
    leaders' parked rows are dropped: the classic fused pass starts the next ring), and the lanes handed over in another order."""
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L, n = 24, 16
    engines = _group(E, T, cfg, n, "llava-1.5", L)
    gen = torch.Generator().manual_seed(11)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [(2 + (i % 3), L) for i in range(n)]
    plain = _run(E, T, engines, embs, spans, K8, 6, rider=False, graph=False)
    # tokens that end some sequences early: the third token of lanes 2 (a ring leader) and 11 (a rider)
    eos = sorted({plain[1][2][2], plain[1][11][2]})
    ref = _run(E, T, engines, embs, spans, K8, 6, rider=False, graph=False, eos=eos)
    assert any(len(t) < 7 for t in ref[1]), "the chosen EOS ids end no sequence"
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, K8, 6, rider=True, graph=graph, eos=eos)
        _same(got, ref, f"EOS run (graph {graph})")

    def mixed(rider):
        T.dd_tools_set_tuning(26, 1 if rider else 0)
        for i, e in enumerate(engines):
            e.rng.manual_seed(50 + i)
            e.prefill(embs[i], *spans[i])
            e.set_eos([])
        grp = E.EngineGroup(engines)
        for _ in range(3):
            grp.decode_step(K8)
        torch.cuda.synchronize()                        # (the rng is seeded on torch's current stream, the steps run on the engines')
        engines[1].rng.manual_seed(777)                 # a ring leader starts over: its parked rows belong to the old sequence
        engines[1].prefill(embs[5], *spans[5])
        grp.decode_step(K8)
        grp.decode_step(K8)
        grp2 = E.EngineGroup(engines[8:] + engines[:8])  # the groups swap roles
        grp2.decode_step(K8)
        grp2.decode_step(K8)
        engines[3].decode_step(K8)                       # a sequence steps alone in between
        grp2.decode_step(K8)
        res = ([e.tokens() for e in engines], [e.logits().copy() for e in engines], [e.kv_sums().copy() for e in engines])
        T.dd_tools_set_tuning(26, 1)
        return res

    a, b = mixed(False), mixed(True)
    assert a[0] == b[0]
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        np.testing.assert_array_equal(x, y)
    for e in reversed(engines):
        e.close()


def test_generate_keeps_whole_groups_while_sequences_end(E, T):
    """EngineGroup.generate with EOS ids: sequences that ended stay in the line-up as long as they fill the last group of eight (their steps
    are no-ops on the device), so the step keeps its rider form; every sequence's tokens and rng stream are those of the classic form and
    of the sequence generated alone."""
    d = 4096
    cfg = E.LMConfig(2048, d, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    L, n, n_new = 24, 24, 9
    engines = _group(E, T, cfg, n, "llava-1.5", L)
    gen = torch.Generator().manual_seed(13)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [(2 + (i % 3), L) for i in range(n)]

    def run(rider, eos):
        T.dd_tools_set_tuning(26, 1 if rider else 0)
        for i, e in enumerate(engines):
            e.rng.manual_seed(50 + i)
            e.prefill(embs[i], *spans[i])
        toks = E.EngineGroup(engines).generate(n_new, eos=eos, mprobs=K8, lookahead=3)
        tails = [e.rng.rand(8).cpu().numpy().copy() for e in engines]
        T.dd_tools_set_tuning(26, 1)
        return toks, tails

    free, _ = run(False, None)
    eos = sorted({free[2][2], free[13][4], free[20][6]})          # ends lanes 2, 13, 20 (and whoever else emits one of them) at different steps
    want, wtails = run(False, eos)
    lens = sorted(len(t) for t in want)
    assert lens[0] < n_new and lens[-1] == n_new, lens
    got, gtails = run(True, eos)
    assert got == want
    for a, b in zip(gtails, wtails):
        np.testing.assert_array_equal(a, b)
    for li in (2, 13, 23):
        e = engines[li]
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        assert e.generate(n_new, eos=eos, mprobs=K8) == want[li], f"lane {li} alone"
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("name,family,dims,kw", [
    ("InstructBLIP rule (quantile masks, vote on the hidden state, the last member's zeros leak into the next un-masked row)", "instructblip",
     (4096, 11008, 32, 32), {}),
    ("InstructBLIP with masked positions (the riding row's RoPE position depends on the leaked zeros)", "instructblip", (4096, 11008, 32, 32),
     {"iblip_positions": "mask"}),
    ("LLaVA-NeXT rule (masks reset every step), GQA", "llava-next", (4096, 14336, 32, 8), {}),
    ("LLaVA-1.5 rule, draws from the Philox stream", "llava-1.5", (4096, 11008, 32, 32), {"rng_stream": "gpu"}),
])
def test_rider_step_other_families(E, T, name, family, dims, kw):
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    L, n, steps = 24, 16, 5
    engines = _group(E, T, cfg, n, family, L, **kw)
    gen = torch.Generator().manual_seed(17)
    embs = [(torch.randn(L + 6 + (i % 5), d, generator=gen) * 0.5).cuda() for i in range(n)]
    spans = [((0 if family == "instructblip" else 2 + (i % 3)), L) for i in range(n)]
    ref = _run(E, T, engines, embs, spans, K8, steps, rider=False, graph=False)
    for graph in (False, True):
        got = _run(E, T, engines, embs, spans, K8, steps, rider=True, graph=graph)
        _same(got, ref, f"{name} (graph {graph})")
    for li in (3, 12):
        e = engines[li]
        e.set_speculation("never")
        e.rng.manual_seed(50 + li)
        e.prefill(embs[li], *spans[li])
        for s in range(steps):
            e.decode_step(K8)
            np.testing.assert_array_equal(e.logits(), ref[0][s][li][0], err_msg=f"{name}: solo lane {li} step {s}")
        assert e.tokens() == ref[1][li]
        e.set_speculation("default")
    for e in reversed(engines):
        e.close()


def test_full_size_64_lanes_repeat_bitwise(E):
    """LLaVA-1.5-7B shapes, all 32 layers, 64 lanes (the bench's line-up: four rings on four branches, stages): three repetitions of
    (prefill, 40 steps) from the same seeds give the same tokens and the same KV checksums — the check tools/stress_lanes.py runs for
    thousands of steps (branch-local mask sampling failed it about once in 2,000 steps)."""
    n, steps = 64, 40
    engines = []
    for i in range(n):
        engines.append(E.DropoutEngine(E.LLAVA15_7B, family=E.FAMILY_LLAVA, max_seq=704, max_visual=576, kv_format="fp16",
                                       share_weights_with=engines[0] if engines else None))
    engines[0].load_synthetic(0, 0.02)
    embs = [torch.randn(608, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(n)]
    first = None
    for rep in range(3):
        torch.cuda.synchronize()
        for e in engines:
            e.rng.manual_seed(24)
        torch.cuda.synchronize()
        for e, x in zip(engines, embs):
            e.prefill(x, 5, 576)
        grp = E.EngineGroup(engines)
        for _ in range(steps):
            grp.decode_step(K8)
        got = ([e.tokens() for e in engines], [e.kv_sums().copy() for e in engines])
        if first is None:
            first = got
            continue
        assert got[0] == first[0], f"repetition {rep}: tokens"
        for i, (a, b) in enumerate(zip(got[1], first[1])):
            np.testing.assert_array_equal(a, b, err_msg=f"repetition {rep}: KV checksums, lane {i}")
    for e in reversed(engines):
        e.close()


@pytest.mark.parametrize("weight_format,family,dims", [("bf16", "llava-1.5", (4096, 11008, 32, 32)), ("fp8", "llava-next", (4096, 14336, 32, 8))])
def test_rstd_workgroup_and_store_placement_do_not_change_a_bit(E, T, weight_format, family, dims):
    """Round 5, last day: the slice kernels' rstd runs in a workgroup of its own behind the streaming ones (SliceArgs::rstd_wg, tools
    key 50) and the slice-pair kernels write their partial sums after the stream (key 36 bit 8 = the old mid-stream placement).  Where things are
    computed and when they are stored is not arithmetic: 16 lanes in the rider form, both placements, every output compared."""
    d, dff, H, Hkv = dims
    cfg = E.LMConfig(2048, d, dff, 2, H, Hkv, 128, 1e-5, 10000.0)
    L = 24
    n_lanes = 16
    engines = _group(E, T, cfg, n_lanes, family, L, **({"weight_format": "fp8"} if weight_format == "fp8" else {}))
    gen = torch.Generator().manual_seed(21)
    T0s = [L + 6 + (i % 5) for i in range(n_lanes)]
    embs = [(torch.randn(T0, d, generator=gen) * 0.5).cuda() for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n_lanes)]
    try:
        new = _run(E, T, engines, embs, spans, K8, 4, rider=True, graph=True)
        T.dd_tools_set_tuning(50, 0)
        T.dd_tools_set_tuning(36, 8)
        old = _run(E, T, engines, embs, spans, K8, 4, rider=True, graph=True)
        _same(old, new, f"rstd in workgroup 0 + mid-stream stores vs the product's placements ({weight_format})")
        T.dd_tools_set_tuning(36, 0)
        half = _run(E, T, engines, embs, spans, K8, 4, rider=True, graph=False)
        _same(half, new, f"rstd in workgroup 0 only ({weight_format})")
    finally:
        T.dd_tools_set_tuning(50, 1)
        T.dd_tools_set_tuning(36, 0)
        for e in reversed(engines):
            e.close()
