"""The oracle against the golden vectors the reference itself produced (oracle/gen_golden.py).

CPU only.  These pin the oracle; the `-m gpu` tests then compare the HIP path with the oracle.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dropout_ref as DR
from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig, random_weights
from oracle.mt19937 import TorchCpuMT19937


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_g1_uncertainty_and_topk(golden_dir):
    g = _load(golden_dir, "g1_uncertainty.npz")
    for c in range(int(g["n_cases"])):
        logits = torch.from_numpy(g[f"c{c}_logits"])
        d = DR.vision_uncertainty(logits)
        for key in ("variance_per_token", "epis_uncert_per_token", "alea_uncert_per_token", "variance",
                    "epis_uncert", "alea_uncert"):
            np.testing.assert_array_equal(d[key].numpy(), g[f"c{c}_{key}"], err_msg=f"case {c} {key}")
        vals, ids = DR.topk_tokens(logits, int(g[f"c{c}_k"]))
        np.testing.assert_array_equal(ids.numpy(), g[f"c{c}_topk_ids"])
        np.testing.assert_array_equal(vals.numpy(), g[f"c{c}_topk_vals"])


def test_g2_overlap_keep(golden_dir):
    g = _load(golden_dir, "g2_overlap.npz")
    topk_ids, start = torch.from_numpy(g["topk_ids"]), int(g["start"])
    seen = set()
    for c in range(int(g["n_cases"])):
        keep = DR.overlap_keep(torch.from_numpy(g[f"c{c}_logits"]), topk_ids)
        idx = torch.nonzero(keep).flatten().numpy() + start
        np.testing.assert_array_equal(idx, g[f"c{c}_idx"])
        seen.add(min(len(idx), 2))
    assert seen == {0, 1, 2}, "fixture must cover 0-match, 1-match (0-d in the reference) and many"


@pytest.mark.parametrize("fam,mode", [("llava", DR.MODE_LLAVA_CUMULATIVE), ("next", DR.MODE_NEXT_RESET),
                                      ("next_no_overlap", DR.MODE_NEXT_NO_OVERLAP), ("iblip", DR.MODE_IBLIP_QUANTILE)])
def test_g3_masks(golden_dir, fam, mode):
    g = _load(golden_dir, "g3_masks.npz")
    for c in range(int(g["n_cases"])):
        epi = torch.from_numpy(g[f"c{c}_epi"])
        probs = [float(p) for p in g[f"c{c}_probs"]]
        keep = DR.overlap_keep(torch.from_numpy(g[f"c{c}_step_logits"]), torch.from_numpy(g[f"c{c}_topk_ids"]))
        uni = torch.from_numpy(g[f"c{c}_uniforms"])
        drop = DR.sample_masks(epi, probs, keep, mode, uni)
        start, L = int(g[f"c{c}_start"]), epi.numel()
        ref = g[f"c{c}_{fam}_masks"]                                   # [K, T] 1 = attend
        assert (ref[:, :start] == 1).all() and (ref[:, start + L:] == 1).all()
        np.testing.assert_array_equal(drop.numpy(), ref[:, start:start + L] == 0, err_msg=f"case {c} {fam}")
        if fam == "llava":
            np.testing.assert_array_equal(drop.sum(1).numpy(), g[f"c{c}_llava_masked_numbers"])


@pytest.mark.parametrize("fam,mode", [("llava_no_overlap", DR.MODE_LLAVA_CUMULATIVE_NO_OVERLAP),
                                      ("iblip_no_overlap", DR.MODE_NEXT_NO_OVERLAP)])
def test_g7_dormant_no_overlap_masks(golden_dir, fam, mode):
    """`epis_no_overlap` (llava.py:663-683 driven cumulatively; instructblip.py:486-505 with the reset): the reference's
    own outputs on the g3 inputs."""
    g, g7 = _load(golden_dir, "g3_masks.npz"), _load(golden_dir, "g7_variants.npz")
    differs = False
    for c in range(int(g7["n_cases"])):
        epi = torch.from_numpy(g[f"c{c}_epi"])
        probs = [float(p) for p in g[f"c{c}_probs"]]
        keep = DR.overlap_keep(torch.from_numpy(g[f"c{c}_step_logits"]), torch.from_numpy(g[f"c{c}_topk_ids"]))
        drop = DR.sample_masks(epi, probs, keep, mode, torch.from_numpy(g[f"c{c}_uniforms"]))
        start, L = int(g[f"c{c}_start"]), epi.numel()
        ref = g7[f"c{c}_{fam}_masks"]
        np.testing.assert_array_equal(drop.numpy(), ref[:, start:start + L] == 0, err_msg=f"case {c} {fam}")
        with_keep = DR.sample_masks(epi, probs, keep, DR.MODE_LLAVA_CUMULATIVE if "llava" in fam else DR.MODE_NEXT_RESET,
                                    torch.from_numpy(g[f"c{c}_uniforms"]))
        differs |= bool((with_keep != drop).any())
    assert differs, "fixture must contain a case where the keep set matters"


def test_g7_select_by_average(golden_dir):
    """models/llava.py:37-52: numpy fp32 mean over the members, bit for bit."""
    g7 = _load(golden_dir, "g7_variants.npz")
    for a in range(int(g7["n_avg"])):
        rows = torch.from_numpy(g7[f"avg{a}_rows"])
        mean = torch.from_numpy(rows.numpy().mean(axis=0))
        np.testing.assert_array_equal(mean.numpy(), g7[f"avg{a}_mean"])
        seq = rows[0].clone()                                       # the order the HIP kernel uses: ((r0+r1)+r2)+... then / K
        for k in range(1, rows.shape[0]):
            seq = seq + rows[k]
        np.testing.assert_array_equal((seq / np.float32(rows.shape[0])).numpy(), g7[f"avg{a}_mean"])


def test_g3_uniforms_are_the_mt19937_stream(golden_dir):
    g = _load(golden_dir, "g3_masks.npz")
    for c in range(int(g["n_cases"])):
        uni = g[f"c{c}_uniforms"]
        rng = TorchCpuMT19937(int(g[f"c{c}_seed"]))
        mine = np.stack([rng.rand_f32(uni.shape[1]) for _ in range(uni.shape[0])])
        np.testing.assert_array_equal(mine, uni)


def test_g3_degenerate_epi_drops_nothing(golden_dir):
    g = _load(golden_dir, "g3_masks.npz")
    c = int(g["n_cases"]) - 1
    assert np.ptp(g[f"c{c}_epi"]) == 0
    assert (g[f"c{c}_llava_masks"] == 1).all() and (g[f"c{c}_next_masks"] == 1).all()


def test_g4_vote(golden_dir):
    g = _load(golden_dir, "g4_vote.npz")
    for c in range(int(g["n_cases"])):
        ids = g[f"c{c}_ids"].tolist()
        win, tok = DR.vote(ids)
        assert win == int(g[f"c{c}_winner"]) and tok == ids[win]


def test_g6_rng_stream(golden_dir):
    g = _load(golden_dir, "g6_rng.npz")
    for c in range(int(g["n_cases"])):
        rng = TorchCpuMT19937(int(g[f"c{c}_seed"]))
        mine = np.concatenate([rng.rand_f32(int(n)) for n in g[f"c{c}_sizes"]])
        np.testing.assert_array_equal(mine, g[f"c{c}_draws"])


def test_g8_philox_stream(golden_dir):
    """oracle/philox.py against (a) the published Philox4x32-10 known answers (Random123 kat_vectors: zero, all-ones and the
    pi-digits case) and (b) draws of torch's GPU generator recorded on the MI355X box (oracle/gen_golden_philox.py)."""
    from oracle.philox import TorchGpuPhilox, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        assert tuple(int(v) for v in philox4x32_10(*ctr, *key)) == want
    g = _load(golden_dir, "g8_philox.npz")
    for c in range(int(g["n_cases"])):
        rng = TorchGpuPhilox(int(g[f"c{c}_seed"]))
        got = []
        for n in g[f"c{c}_sizes"]:
            d = rng.rand_f32(int(n))
            got.append(d if d.size <= 8192 else np.concatenate([d[:2048], d[-2048:]]))
        np.testing.assert_array_equal(np.concatenate(got), g[f"c{c}_draws"])
        assert rng.offset == 4 * len(g[f"c{c}_sizes"])
    with pytest.raises(ValueError):
        TorchGpuPhilox(1, offset=2)


def _decoder_from(g, family):
    v, d, f, nl, nh, nkv, hd = [int(x) for x in g["cfg"]]
    cfg = LMConfig(v, d, f, nl, nh, nkv, hd, float(g["rms_eps"]), float(g["rope_theta"]))
    w = random_weights(cfg, int(g["wseed"]), float(g["std"]))
    seed = int(g["rseed"]) if "rseed" in g.files else 0
    use_random = bool(int(g["use_random"])) if "use_random" in g.files else False
    return RefDecoder(family, cfg, w, [float(p) for p in g["probs"]], seed=seed, use_random=use_random), cfg


@pytest.mark.parametrize("name,family", [("g5_llava_k3.npz", FAMILY_LLAVA), ("g5_llava_k8.npz", FAMILY_LLAVA),
                                         ("g5_next_k4.npz", FAMILY_NEXT), ("g5_next_norestore_k2.npz", FAMILY_NEXT)])
def test_g5_end_to_end_llava_families(golden_dir, name, family):
    g = _load(golden_dir, name)
    dec, cfg = _decoder_from(g, family)
    toks = dec.generate(torch.from_numpy(g["embeds"]), int(g["span_start"]), int(g["span_len"]), len(g["tokens"]))
    np.testing.assert_allclose(dec.prefill_logits[-1].numpy(), g["prefill_logits_last"], rtol=1e-3, atol=2e-5)
    s0, L = int(g["span_start"]), int(g["span_len"])
    np.testing.assert_allclose(dec.prefill_logits[s0:s0 + L].numpy(), g["prefill_image_logits"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(dec.epi.numpy(), g["epis_uncert_per_token"][0], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(dec.uncert["alea_uncert_per_token"][0].numpy(), g["alea_uncert_per_token"][0], rtol=1e-4)
    np.testing.assert_allclose(dec.uncert["variance_per_token"][0].numpy(), g["variance_per_token"][0], rtol=1e-3)
    assert set(map(tuple, np.sort(dec.topk_ids.numpy(), 1))) == set(map(tuple, np.sort(g["topk_ids"], 1)))
    assert toks == g["tokens"].tolist()
    for s, rec in enumerate(dec.records):
        assert rec.base_argmax == int(g["step_base_argmax"][s])
        np.testing.assert_array_equal(rec.uniforms, g["uniforms"][s])
        np.testing.assert_array_equal(rec.drop, g["step_drop"][s].astype(bool), err_msg=f"step {s}")
        assert rec.member_argmax == g["step_member_argmax"][s].tolist()
        assert rec.winner == int(g["step_winner"][s])
        np.testing.assert_allclose(rec.logits, g["step_logits"][s], rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(rec.base_logits, g["step_base_logits"][s], rtol=1e-3, atol=2e-5)
        if "step_masked_numbers" in g.files:
            assert rec.masked_numbers == g["step_masked_numbers"][s].tolist()
    assert dec.cache.length == int(g["kv_len"])
    for i in range(cfg.num_layers):
        assert abs(float(dec.cache.k[i].double().sum()) - g["kv_k_sum"][i]) < 1e-2
        assert abs(float(dec.cache.v[i].double().sum()) - g["kv_v_sum"][i]) < 1e-2


def test_g5_end_to_end_instructblip(golden_dir):
    g = _load(golden_dir, "g5_iblip_k3.npz")
    dec, cfg = _decoder_from(g, FAMILY_IBLIP)
    toks = dec.generate(torch.from_numpy(g["embeds"]), 0, int(g["span_len"]), len(g["tokens"]))
    assert toks == g["tokens"].tolist()
    np.testing.assert_allclose(dec.epi.numpy(), g["epis_uncert_per_token"][0], rtol=2e-4, atol=1e-6)
    for s, rec in enumerate(dec.records):
        np.testing.assert_array_equal(rec.drop, g["step_drop"][s].astype(bool), err_msg=f"step {s}")
        assert rec.member_argmax == g["step_member_argmax"][s].tolist()          # Q3: hidden-state argmax
        # Q2: base pass of step s sees the zeros the last member of step s-1 left behind
        leaked = np.zeros(int(g["span_len"]), bool) if s == 0 else g["step_drop"][s - 1][-1].astype(bool)
        np.testing.assert_array_equal(g["step_base_drop"][s].astype(bool), leaked)
    winners = [r.winner for r in dec.records]
    assert any(w != 0 for w in winners), "fixture should exercise a non-first winner"


def test_g7_epis_kl_keep_set_and_masks(golden_dir):
    """`epis_kl` (instructblip.py:464-485) and `lowest_percent_kl_indices` (:559-578): the reference's own outputs on the g3
    inputs plus random image logits stored with the fixture."""
    g, g7 = _load(golden_dir, "g3_masks.npz"), _load(golden_dir, "g7_variants.npz")
    seen = 0
    for c in range(int(g7["n_cases"])):
        if f"c{c}_kl_lowest" not in g7.files:
            continue
        seen += 1
        epi = torch.from_numpy(g[f"c{c}_epi"])
        probs = [float(p) for p in g[f"c{c}_probs"]]
        img = torch.from_numpy(g7[f"c{c}_kl_image_logits"])
        keep = DR.kl_keep(img, torch.from_numpy(g[f"c{c}_step_logits"]))
        assert sorted(torch.nonzero(keep).flatten().tolist()) == sorted(g7[f"c{c}_kl_lowest"].tolist()), f"case {c}"
        drop = DR.sample_masks(epi, probs, keep, DR.MODE_IBLIP_KL, torch.from_numpy(g[f"c{c}_uniforms"]))
        start, L = int(g[f"c{c}_start"]), epi.numel()
        np.testing.assert_array_equal(drop.numpy(), g7[f"c{c}_iblip_kl_masks"][:, start:start + L] == 0, err_msg=f"case {c}")
    assert seen >= 4
