"""Captions that end on EOS, image after image on ONE rng stream (the reference's shape: chair_test.py:274-346 decodes its
images back to back, HF's greedy loop stops each at EOS, and the global generator seeded once at import — models/llava.py:16-20 —
keeps running through `torch.rand_like` at :650).

Decode steps are enqueued ahead of the host's knowledge of the tokens, so the stop is device-side (dd_lm_set_eos,
DDState::done): the steps enqueued beyond the EOS step must not draw from the stream, emit tokens or move any state.
Checked against the oracle with ONE continuing stream and no re-seed between images: tokens and masks bit-exact for every
image, the rng state equal afterwards, independent of the look-ahead depth, of graph replay vs eager launches and of
repetition (determinism)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights

RC = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
PROBS = [0.2, 0.4, 0.6, 0.8]
N_NEW = 12


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _images(n, seed0=300, L=30):
    shapes = [(36 + 5 * i, 1 + i % 4, L) for i in range(n)]
    embs = [torch.randn(T0, RC.hidden_size, generator=torch.Generator().manual_seed(seed0 + i)) * 0.8 for i, (T0, _, _) in enumerate(shapes)]
    return shapes, embs


def _engine(E, family, w, seed, owner=None, L=30):
    cfg = E.LMConfig(RC.vocab_size, RC.hidden_size, RC.intermediate_size, RC.num_layers, RC.num_heads, RC.num_kv_heads,
                     RC.head_dim, RC.rms_eps, RC.rope_theta)
    e = E.DropoutEngine(cfg, family=family, max_seq=160, max_visual=L + 2, seed=seed, share_weights_with=owner)
    if owner is None:
        e.load_state_dict(w)
    return e


def _pick_eos(family, w, shapes, embs, seed):
    """An id that ends the FIRST image early (its 4th token) — later images stop wherever they happen to emit it."""
    free = RefDecoder(family, RC, w, PROBS, seed=seed).generate(embs[0], shapes[0][1], shapes[0][2], N_NEW)
    return free[3]


def _oracle_run(family, w, shapes, embs, seed, eos):
    """One decoder, one stream, images back to back, each stopped at EOS like HF's loop; -> per image (tokens, last drop,
    last logits) and the stream's next draws."""
    ref = RefDecoder(family, RC, w, PROBS, seed=seed)
    out = []
    for (T0, s0, L), emb in zip(shapes, embs):
        toks = ref.generate(emb, s0, L, N_NEW, eos=eos)
        last = ref.records[-1] if ref.records else None
        out.append((toks, None if last is None else last.drop.copy(), None if last is None else last.logits.copy(),
                    None if last is None else list(last.masked_numbers)))
    return out, ref.rng.rand_f32(16)


@pytest.mark.parametrize("family", [FAMILY_LLAVA, FAMILY_NEXT])
def test_back_to_back_images_with_eos_continue_one_stream(E, family):
    w = random_weights(RC, 31, 0.05)
    shapes, embs = _images(3)
    seed = 5217
    eos = _pick_eos(family, w, shapes, embs, seed)
    want, want_next = _oracle_run(family, w, shapes, embs, seed, eos)
    assert len(want[0][0]) == 4 and want[0][0][-1] == eos          # the first caption does end on EOS, early
    eng = _engine(E, family, w, seed)
    lib = eng.lib
    runs = 0
    try:
        for graph in (1, 0):
            lib.dd_set_tuning(8, graph)
            for lookahead in (1, 2, 6, 11) if graph else (6,):
                for rep in range(4 if graph else 4):           # 4 x 4 + 4 = 20 repeats in all
                    eng.rng.manual_seed(seed)
                    for i, ((T0, s0, L), emb) in enumerate(zip(shapes, embs)):
                        eng.prefill(emb.cuda(), s0, L)
                        toks = eng.generate(N_NEW, eos=[eos], mprobs=PROBS, lookahead=lookahead)
                        wt, wdrop, wlogits, wnum = want[i]
                        assert toks == wt, f"image {i} (graph={graph}, lookahead={lookahead}, rep={rep})"
                        if wdrop is not None:
                            # diagnostics after the loop describe the EOS step, not a wasted look-ahead step
                            st = eng.last_step()
                            np.testing.assert_array_equal(st["drop"], wdrop, err_msg=f"image {i}")
                            assert st["masked_numbers"].tolist() == wnum
                            assert np.abs(eng.logits() - wlogits).max() <= 1e-3 * np.abs(wlogits).max()
                        assert eng.T() == T0 + len(wt) - 1
                    got_next = eng.rng.rand(16).cpu().numpy()
                    np.testing.assert_array_equal(got_next, want_next)   # the stream stands where the reference's would
                    runs += 1
    finally:
        lib.dd_set_tuning(8, 1)
    assert runs == 20
    eng.close()


def test_eos_as_the_prefill_token_and_clearing_the_list(E):
    """The prefill's greedy token can itself be the EOS (HF stops with one token); an empty list never stops."""
    w = random_weights(RC, 31, 0.05)
    shapes, embs = _images(2)
    seed = 11
    first = RefDecoder(FAMILY_LLAVA, RC, w, PROBS, seed=seed).generate(embs[0], shapes[0][1], shapes[0][2], 1)[0]
    ref = RefDecoder(FAMILY_LLAVA, RC, w, PROBS, seed=seed)
    w0 = ref.generate(embs[0], shapes[0][1], shapes[0][2], N_NEW, eos=first)
    w1 = ref.generate(embs[1], shapes[1][1], shapes[1][2], 6, eos=None)
    assert w0 == [first]
    eng = _engine(E, FAMILY_LLAVA, w, seed)
    eng.prefill(embs[0].cuda(), shapes[0][1], shapes[0][2])
    assert eng.generate(N_NEW, eos=first, mprobs=PROBS) == w0
    # forced steps after the end change nothing either (the sequence is finished until the next prefill)
    for _ in range(3):
        eng.decode_step(PROBS)
    assert eng.tokens() == w0
    eng.prefill(embs[1].cuda(), shapes[1][1], shapes[1][2])
    assert eng.generate(6, eos=None, mprobs=PROBS) == w1            # list cleared: runs to n_new, stream untouched by image 0
    eng.close()


@pytest.mark.parametrize("family", [FAMILY_LLAVA, FAMILY_IBLIP])
def test_lanes_back_to_back_with_eos_each_lane_its_own_stream(E, family):
    """EngineGroup: every lane is one reference process (own stream, continuing over that lane's images).  Lanes finish at
    different steps; the group keeps stepping the others while the finished lane's look-ahead steps are no-ops."""
    w = random_weights(RC, 31, 0.05)
    n_lanes, rounds = 5, 2
    L = 32 if family == FAMILY_IBLIP else 30
    shapes, embs = _images(n_lanes * rounds, L=L)
    if family == FAMILY_IBLIP:
        shapes = [(T0, 0, L) for (T0, _, _) in shapes]
    seeds = [40 + i for i in range(n_lanes)]
    eos = _pick_eos(family, w, shapes, embs, seeds[0])
    wants = []
    for i in range(n_lanes):
        idx = [r * n_lanes + i for r in range(rounds)]
        wants.append(_oracle_run(family, w, [shapes[j] for j in idx], [embs[j] for j in idx], seeds[i], eos))
    assert len(wants[0][0][0][0]) == 4
    engines = []
    for i in range(n_lanes):
        engines.append(_engine(E, family, w, seeds[i], owner=engines[0] if engines else None, L=L))
    grp = E.EngineGroup(engines)
    for rep in range(4):
        for i, e in enumerate(engines):
            e.rng.manual_seed(seeds[i])
        for r in range(rounds):
            for i, e in enumerate(engines):
                T0, s0, Ls = shapes[r * n_lanes + i]
                e.prefill(embs[r * n_lanes + i].cuda(), s0, Ls)
            got = grp.generate(N_NEW, eos=[eos], mprobs=PROBS, lookahead=(6 if rep % 2 == 0 else 2))
            for i in range(n_lanes):
                wt, wdrop, wlogits, wnum = wants[i][0][r]
                assert got[i] == wt, f"lane {i} round {r} rep {rep}"
                if wdrop is not None:
                    np.testing.assert_array_equal(engines[i].last_step()["drop"], wdrop, err_msg=f"lane {i} round {r}")
        if family != FAMILY_IBLIP:        # InstructBLIP's masks are deterministic: no stream to compare
            for i, e in enumerate(engines):
                np.testing.assert_array_equal(e.rng.rand(16).cpu().numpy(), wants[i][1], err_msg=f"lane {i}")
    for e in reversed(engines):
        e.close()
