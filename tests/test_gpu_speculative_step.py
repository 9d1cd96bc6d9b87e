"""Speculative single-sequence step (dd_engine.hip decode_step_spec): the K members ride in the same sweep as the un-masked
row with masks sampled for an empty keep set, and are re-run only when the real keep set (models/llava.py:603, 660) would
have restored a token one of them dropped.  Exactness: every result of every step — tokens, masks, masked_numbers, member
argmax, winner, logits (bitwise), KV rows, and the rng stream afterwards — equals the two-sweep step's, for the three
families, with both outcomes of the speculation occurring."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights

RC = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _run(E, eng, emb, s0, L, probs, steps, spec, graph, seed, sync=False):
    lib = eng.lib
    lib.dd_set_tuning(14, int(spec))            # 0 never, 1 always, 2 adaptive (the library's default policy)
    lib.dd_set_tuning(8, 1 if graph else 0)
    try:
        eng.rng.manual_seed(seed)
        eng.prefill(emb.cuda(), s0, L)
        recs, oks = [], []
        for _ in range(steps):
            held = eng.decode_step_sync(probs) if sync else None
            if not sync:
                eng.decode_step(probs)
            st = eng.last_step()
            recs.append((st["drop"].copy(), st["masked_numbers"].tolist(), st["member_argmax"].tolist(), st["winner"], st["keep"].copy(),
                         eng.logits().copy(), eng.base_logits().copy()))
            oks.append(held if int(spec) == 2 else (eng.spec_ok() if spec else -1))     # adaptive: -1 = a plain two-sweep step
            assert held is None or held == oks[-1]
        return recs, eng.tokens(), eng.kv_sums().copy(), eng.rng.rand(32).cpu().numpy(), oks
    finally:
        lib.dd_set_tuning(14, 2)
        lib.dd_set_tuning(8, 1)


@pytest.mark.parametrize("family,K,use_random", [(FAMILY_LLAVA, 8, False), (FAMILY_LLAVA, 3, False), (FAMILY_NEXT, 4, False),
                                                 (FAMILY_NEXT, 4, True), (FAMILY_IBLIP, 8, False)])
def test_speculative_step_equals_two_sweep_step(E, family, K, use_random):
    w = random_weights(RC, 31, 0.05)
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L = 32 if family == FAMILY_IBLIP else 40
    s0 = 0 if family == FAMILY_IBLIP else 3
    eng = E.DropoutEngine(cfg, family=family, max_seq=192, max_visual=L, seed=7, use_random=use_random)
    eng.load_state_dict(w)
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.8
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]
    steps = 40
    ref = _run(E, eng, emb, s0, L, probs, steps, spec=False, graph=False, seed=7)
    for graph, sync in ((False, False), (True, False), (False, True), (True, True)):   # sync: the host decides the fallback
        got = _run(E, eng, emb, s0, L, probs, steps, spec=True, graph=graph, seed=7, sync=sync)
        assert got[1] == ref[1]
        for s, (a, b) in enumerate(zip(got[0], ref[0])):
            np.testing.assert_array_equal(a[0], b[0], err_msg=f"drop masks, step {s}")
            assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3], f"step {s}"
            np.testing.assert_array_equal(a[4], b[4], err_msg=f"keep set, step {s}")
            np.testing.assert_array_equal(a[5], b[5], err_msg=f"winner logits, step {s}")
            np.testing.assert_array_equal(a[6], b[6], err_msg=f"base logits, step {s}")
        np.testing.assert_array_equal(got[2], ref[2])
        np.testing.assert_array_equal(got[3], ref[3])          # the rng stream stands where the two-sweep step leaves it
        oks = got[4]
        if family != FAMILY_IBLIP and not use_random:
            assert 0 < sum(oks) < steps, f"both outcomes must occur for the test to mean something: {oks}"
        if use_random:                                          # epis_no_overlap: the keep set is not used, never a re-run
            assert sum(oks) == steps
    # and the oracle agrees (tokens; the golden / oracle suites cover the rest through the same default path)
    want = RefDecoder(family, RC, w, probs, seed=7, use_random=use_random).generate(emb, s0, L, steps + 1)
    assert ref[1] == want
    eng.close()


def test_epis_kl_mode_end_to_end_against_the_oracle(E):
    """The dormant "epis_kl" method of InstructBLIP (models/instructblip.py:123, 464-485, 559-578): stochastic masks with the
    reset, the 10 % of visual tokens whose prefill distribution is closest to the step's restored.  Engine (speculative and
    two-sweep steps, and as a lane of a group) against the oracle restating the reference's functions."""
    w = random_weights(RC, 31, 0.05)
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L, s0, probs = 32, 0, [0.2, 0.4, 0.6, 0.8]
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(78)) * 0.8
    ref = RefDecoder(FAMILY_IBLIP, RC, w, probs, seed=7, mask_method="epis_kl")
    want = ref.generate(emb, s0, L, 21)
    eng = E.DropoutEngine(cfg, family=FAMILY_IBLIP, max_seq=192, max_visual=L, seed=7, mask_method="epis_kl")
    eng.load_state_dict(w)
    lib = eng.lib
    for spec in (1, 0):
        lib.dd_set_tuning(14, spec)
        try:
            eng.rng.manual_seed(7)
            eng.prefill(emb.cuda(), s0, L)
            for s in range(20):
                eng.decode_step(probs)
                st, r = eng.last_step(), ref.records[s]
                np.testing.assert_array_equal(st["keep"], r.keep, err_msg=f"KL keep set, step {s} (spec={spec})")
                np.testing.assert_array_equal(st["drop"], r.drop, err_msg=f"masks, step {s} (spec={spec})")
                assert st["member_argmax"].tolist() == r.member_argmax and st["winner"] == r.winner
            assert eng.tokens() == want
        finally:
            lib.dd_set_tuning(14, 1)
    assert int(ref.records[0].keep.sum()) == 3                 # int(0.1 * 32)
    lane = E.DropoutEngine(cfg, family=FAMILY_IBLIP, max_seq=192, max_visual=L, seed=9, mask_method="epis_kl", share_weights_with=eng)
    emb2 = torch.randn(L + 5, RC.hidden_size, generator=torch.Generator().manual_seed(79)) * 0.8
    eng.rng.manual_seed(7)
    eng.prefill(emb.cuda(), s0, L)
    lane.prefill(emb2.cuda(), s0, L)
    got = E.EngineGroup([eng, lane]).generate(21, mprobs=probs)
    assert got[0] == want
    assert got[1] == RefDecoder(FAMILY_IBLIP, RC, w, probs, seed=9, mask_method="epis_kl").generate(emb2, s0, L, 21)
    lane.close()
    eng.close()


@pytest.mark.parametrize("family,use_random", [(FAMILY_LLAVA, False), (FAMILY_NEXT, False), (FAMILY_NEXT, True)])
def test_philox_stream_end_to_end(E, family, use_random):
    """rng_stream="gpu": the draws of llava.py:650 come from the GPU generator's Philox stream (the reference run on a GPU).
    Two-sweep step, speculative step (with its backup / re-run of the draws), graph replays and a 3-lane group all give the
    oracle's tokens and leave the stream at the oracle's offset."""
    w = random_weights(RC, 31, 0.05)
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L, s0, probs, steps = 40, 3, [0.3, 0.5, 0.7, 0.9], 30
    eng = E.DropoutEngine(cfg, family=family, max_seq=192, max_visual=L, seed=7, use_random=use_random, rng_stream="gpu")
    eng.load_state_dict(w)
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.8
    ref = RefDecoder(family, RC, w, probs, seed=7, use_random=use_random, rng_stream="gpu")
    want = ref.generate(emb, s0, L, steps + 1)
    tail = ref.rng.rand_f32(32)
    cpu_tokens = RefDecoder(family, RC, w, probs, seed=7, use_random=use_random).generate(emb, s0, L, steps + 1)
    assert want != cpu_tokens, "the two generators must lead to different sequences for the test to mean something"
    oks_seen = []
    for spec, graph, sync in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        _, toks, _, after, oks = _run(E, eng, emb, s0, L, probs, steps, spec=spec, graph=graph, seed=7, sync=sync)
        assert toks == want, (spec, graph, sync)
        np.testing.assert_array_equal(after, tail)
        oks_seen += [o for o in oks if o >= 0]
    if not use_random:
        assert 0 < sum(oks_seen) < len(oks_seen), "both outcomes of the speculation must occur"
    # lanes: each lane its own Philox stream (seeds differ), one group step for all
    seeds = [7, 8, 9]
    lanes = [eng] + [E.DropoutEngine(cfg, family=family, max_seq=192, max_visual=L, seed=s, use_random=use_random,
                                     rng_stream="gpu", share_weights_with=eng) for s in seeds[1:]]
    for e, s in zip(lanes, seeds):
        e.rng.manual_seed(s)
        e.prefill(emb.cuda(), s0, L)
    got = E.EngineGroup(lanes).generate(steps + 1, mprobs=probs)
    for i, s in enumerate(seeds):
        r = RefDecoder(family, RC, w, probs, seed=s, use_random=use_random, rng_stream="gpu")
        assert got[i] == r.generate(emb, s0, L, steps + 1), f"lane {i}"
        np.testing.assert_array_equal(lanes[i].rng.rand(8).cpu().numpy(), r.rng.rand_f32(8))
    for e in reversed(lanes):
        e.close()


def test_generate_uses_the_host_decided_step_and_matches_the_queued_loop(E):
    """DropoutEngine.generate(): the one-sequence loop goes through dd_lm_decode_step_sync; tokens, the rng stream afterwards
    and the EOS stop equal the queued loop's (sync_steps = False) and the oracle's."""
    w = random_weights(RC, 31, 0.05)
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L, s0, probs = 40, 3, [0.2, 0.4, 0.6, 0.8]
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=192, max_visual=L, seed=7)
    eng.load_state_dict(w)
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.8
    want = RefDecoder(FAMILY_LLAVA, RC, w, probs, seed=7).generate(emb, s0, L, 40)
    eos = want[17]
    cut = want[:want.index(eos) + 1]
    outs = []
    for sync in (True, False, True):
        eng.sync_steps = sync
        eng.rng.manual_seed(7)
        eng.prefill(emb.cuda(), s0, L)
        full = eng.generate(40, mprobs=probs)
        eng.prefill(emb.cuda(), s0, L)                 # second image on the same stream, stopped by an EOS
        short = eng.generate(40, eos=[eos], mprobs=probs)
        outs.append((full, short, eng.rng.rand(16).cpu().numpy()))
    assert outs[0][0] == want
    ref2 = RefDecoder(FAMILY_LLAVA, RC, w, probs, seed=7)
    ref2.generate(emb, s0, L, 40)
    second = ref2.generate(emb, s0, L, 40)
    assert outs[0][1] == second[:second.index(eos) + 1] if eos in second else second
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[1] == outs[0][1]
        np.testing.assert_array_equal(o[2], outs[0][2])
    eng.close()


def test_adaptive_policy_on_a_keep_set_that_is_rarely_empty(E):
    """On a real checkpoint the keep set (models/llava.py:443-482: visual tokens whose top-k ids contain the un-masked pass'
    argmax) is routinely non-empty and some member has almost always dropped one of its tokens, so the speculation fails.
    Built here by construction: 24 lm_head rows scaled up, so the un-masked argmax is always one of those ids and they fill
    the top-k lists of the visual tokens — about 5 / 24 of the visual tokens are kept at every step.  The adaptive policy must then fall back to plain two-sweep steps (and re-probe), the
    'always' policy must re-run nearly every step, and every result must equal the two-sweep step's and the oracle's."""
    w = dict(random_weights(RC, 31, 0.05))
    w["lm_head.weight"] = w["lm_head.weight"].clone()
    w["lm_head.weight"][64:88] *= 4.0
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    L, s0, steps = 40, 3, 150
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=256, max_visual=L, seed=7)
    eng.load_state_dict(w)
    emb = torch.randn(L + 9, RC.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.8
    ref = _run(E, eng, emb, s0, L, probs, steps, spec=0, graph=False, seed=7)
    kept = [int(r[4].sum()) for r in ref[0]]
    assert sum(k > 0 for k in kept) > 0.6 * steps, kept              # the construction works: most steps have tokens to keep
    for graph in (False, True):
        eng.spec_stats(reset=True)
        always = _run(E, eng, emb, s0, L, probs, steps, spec=1, graph=graph, seed=7, sync=True)
        st_always = eng.spec_stats(reset=True)
        adaptive = _run(E, eng, emb, s0, L, probs, steps, spec=2, graph=graph, seed=7, sync=True)
        st_adapt = eng.spec_stats(reset=True)
        for got in (always, adaptive):
            assert got[1] == ref[1]
            for s, (a, b) in enumerate(zip(got[0], ref[0])):
                np.testing.assert_array_equal(a[0], b[0], err_msg=f"drop masks, step {s}")
                assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3], f"step {s}"
                np.testing.assert_array_equal(a[4], b[4], err_msg=f"keep set, step {s}")
                np.testing.assert_array_equal(a[5], b[5], err_msg=f"winner logits, step {s}")
            np.testing.assert_array_equal(got[2], ref[2])
            np.testing.assert_array_equal(got[3], ref[3])
        assert st_always["held"] + st_always["rerun"] == steps and st_always["plain"] == 0
        assert st_always["hit_rate"] < 0.25, st_always              # below the break-even: speculating always is the slow choice
        assert all(ok == 1 for ok, k in zip(always[4], kept) if k == 0)     # an empty keep set can never fail the check
        assert st_adapt["switched_off"] >= 1 and st_adapt["plain"] >= steps // 3, st_adapt
        assert st_adapt["held"] + st_adapt["rerun"] + st_adapt["plain"] == steps
        assert st_adapt["rerun"] < st_always["rerun"] // 2                  # the probes are the only re-runs left
        assert adaptive[4].count(-1) == st_adapt["plain"]
    want = RefDecoder(FAMILY_LLAVA, RC, w, probs, seed=7).generate(emb, s0, L, steps + 1)
    assert ref[1] == want
    eng.close()
