"""HIP kernels of the dropout-specific functions vs the oracle and the reference's golden vectors (through the C-ABI)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dropout_ref as DR
from oracle.mt19937 import TorchCpuMT19937
from oracle.philox import TorchGpuPhilox


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="module")
def ops():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import dropout
    assert torch.cuda.is_available()
    return dropout


def test_rng_is_torch_cpu_stream(ops, golden_dir):
    g = _load(golden_dir, "g6_rng.npz")
    for c in range(int(g["n_cases"])):
        rng = ops.TorchCpuCompatRNG(int(g[f"c{c}_seed"]))
        mine = np.concatenate([rng.rand(int(n)).cpu().numpy() for n in g[f"c{c}_sizes"]])
        np.testing.assert_array_equal(mine, g[f"c{c}_draws"])
    rng = ops.TorchCpuCompatRNG(123)
    ref = TorchCpuMT19937(123)
    for n in (1, 623, 624, 625, 5000, 3, 1248):
        np.testing.assert_array_equal(rng.rand(n).cpu().numpy(), ref.rand_f32(n))
    rng.manual_seed(5217)
    np.testing.assert_array_equal(rng.rand(700).cpu().numpy(), TorchCpuMT19937(5217).rand_f32(700))


def test_philox_stream_is_torch_gpu_rand(ops, golden_dir):
    """dd_rng_create_philox == torch.manual_seed(s); torch.rand(n, device="cuda") of THIS box's torch (the generator the
    reference's rand_like at models/llava.py:650 draws from on a GPU), == oracle/philox.py, == the committed vectors."""
    cpu_state, gpu_state = torch.get_rng_state(), torch.cuda.get_rng_state()
    try:
        for seed, sizes in ((0, [5, 576, 576]), (5217, [576, 2928, 8192, 1]), ((1 << 40) + 3, [1, 1023, 300000, 524288])):
            torch.manual_seed(seed)
            rng, ref = ops.TorchGpuCompatRNG(seed), TorchGpuPhilox(seed)
            for n in sizes:
                want = torch.rand(n, device="cuda")
                mine = rng.rand(n)
                assert torch.equal(mine, want), (seed, n)
                np.testing.assert_array_equal(mine.cpu().numpy(), ref.rand_f32(n))
        torch.manual_seed(77)
        torch.rand(576, device="cuda")
        torch.rand(576, device="cuda")                         # offset 8 from here
        want = torch.rand(100, device="cuda")
        assert torch.equal(ops.TorchGpuCompatRNG(77, offset=8).rand(100), want)
        rng.manual_seed(9)                                      # torch.manual_seed on the same generator: offset back to 0
        torch.manual_seed(9)
        assert torch.equal(rng.rand(700), torch.rand(700, device="cuda"))
    finally:
        torch.set_rng_state(cpu_state)
        torch.cuda.set_rng_state(gpu_state)
    g = _load(golden_dir, "g8_philox.npz")
    for c in range(int(g["n_cases"])):
        rng = ops.TorchGpuCompatRNG(int(g[f"c{c}_seed"]))
        got = []
        for n in g[f"c{c}_sizes"]:
            d = rng.rand(int(n)).cpu().numpy()
            got.append(d if d.size <= 8192 else np.concatenate([d[:2048], d[-2048:]]))
        np.testing.assert_array_equal(np.concatenate(got), g[f"c{c}_draws"])
    with pytest.raises(Exception):
        ops.TorchGpuCompatRNG(1, offset=6)
    with pytest.raises(Exception):
        ops.TorchGpuCompatRNG(1).rand(524289)


@pytest.mark.parametrize("mode", [DR.MODE_LLAVA_CUMULATIVE, DR.MODE_NEXT_RESET])
def test_masks_from_the_philox_stream(ops, mode):
    """K members x several steps draw consecutive rand_like(epi) calls of the GPU generator, per member, like llava.py:650."""
    rs = np.random.RandomState(3)
    for L, probs in ((576, [0.3, 0.5, 0.7]), (2928, [0.5] * 5), (33, [0.1, 0.9]), (1500, [0.2, 0.4, 0.6, 0.8, 0.25, 0.33, 0.5, 0.7])):
        epi = torch.from_numpy((rs.rand(L) * 2).astype(np.float32))
        keep = torch.from_numpy(rs.rand(L) < 0.05)
        rng, ref = ops.TorchGpuCompatRNG(5217), TorchGpuPhilox(5217)
        for step in range(3):
            uni = torch.from_numpy(np.stack([ref.rand_f32(L) for _ in probs]))
            want = DR.sample_masks(epi, probs, keep, mode, uni)
            drop, nd = ops.sample_masks(epi.cuda(), probs, keep.cuda(), mode, rng=rng)
            np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), want.numpy(), err_msg=f"L={L} step={step}")
        np.testing.assert_array_equal(rng.rand(64).cpu().numpy(), ref.rand_f32(64))      # both streams end at the same offset


def test_uncertainty_golden(ops, golden_dir):
    g = _load(golden_dir, "g1_uncertainty.npz")
    for c in range(int(g["n_cases"])):
        logits = torch.from_numpy(g[f"c{c}_logits"]).cuda()
        k = int(g[f"c{c}_k"])
        d, (vals, ids) = ops.calculate_vision_uncertainty(logits, topk=k)
        for key, rtol in (("epis_uncert_per_token", 2e-5), ("alea_uncert_per_token", 2e-5), ("variance_per_token", 1e-4),
                          ("variance", 1e-4), ("epis_uncert", 2e-5), ("alea_uncert", 2e-5)):
            np.testing.assert_allclose(d[key].cpu().numpy(), g[f"c{c}_{key}"], rtol=rtol, atol=1e-9, err_msg=f"{c} {key}")
        np.testing.assert_array_equal(ids[0].cpu().numpy(), g[f"c{c}_topk_ids"][0])     # bit-exact index work
        np.testing.assert_array_equal(vals[0].cpu().numpy(), g[f"c{c}_topk_vals"][0])


@pytest.mark.parametrize("L,V,scale", [(576, 32064, 4.0), (32, 32001, 6.0), (7, 130, 2.0)])
def test_uncertainty_full_size_vs_oracle(ops, L, V, scale):
    gen = torch.Generator().manual_seed(L + V)
    logits = (torch.randn(1, L, V, generator=gen) * scale).float()
    ref = DR.vision_uncertainty(logits)
    rv, ri = DR.topk_tokens(logits, 10)
    d, (vals, ids) = ops.calculate_vision_uncertainty(logits.cuda(), topk=10)
    np.testing.assert_allclose(d["epis_uncert_per_token"].cpu().numpy(), ref["epis_uncert_per_token"].numpy(), rtol=5e-5)
    np.testing.assert_allclose(d["alea_uncert_per_token"].cpu().numpy(), ref["alea_uncert_per_token"].numpy(), rtol=5e-5)
    np.testing.assert_allclose(d["variance_per_token"].cpu().numpy(), ref["variance_per_token"].numpy(), rtol=2e-4)
    np.testing.assert_array_equal(ids[0].cpu().numpy(), ri[0].numpy())
    np.testing.assert_array_equal(vals[0].cpu().numpy(), rv[0].numpy())


def test_uncertainty_padded_rows(ops):
    """ld > V (the engine's padded vocabulary): padding columns must be ignored."""
    gen = torch.Generator().manual_seed(3)
    L, V, ld = 5, 100, 112
    buf = torch.full((L, ld), 1e30)
    buf[:, :V] = torch.randn(L, V, generator=gen) * 3
    ref = DR.vision_uncertainty(buf[None, :, :V].contiguous())
    d = ops.calculate_vision_uncertainty(buf.cuda()[:, :V])
    np.testing.assert_allclose(d["epis_uncert_per_token"].cpu().numpy(), ref["epis_uncert_per_token"].numpy(), rtol=5e-5)


def test_overlap_keep_golden(ops, golden_dir):
    g = _load(golden_dir, "g2_overlap.npz")
    topk = torch.from_numpy(g["topk_ids"]).cuda()
    for c in range(int(g["n_cases"])):
        idx, keep = ops.get_overlap_image_tokens(torch.from_numpy(g[f"c{c}_logits"]).cuda(), topk, int(g["start"]))
        np.testing.assert_array_equal(idx.cpu().numpy(), g[f"c{c}_idx"])


def test_argmax_first_maximal_index(ops):
    x = torch.zeros(3, 1000)
    x[0, 17] = x[0, 600] = 5.0
    x[1, 999] = 1.0
    x[2] = -1.0
    x[2, 0] = x[2, 5] = 0.5
    assert ops.argmax_rows(x.cuda()).tolist() == torch.argmax(x, -1).tolist() == [17, 999, 0]


@pytest.mark.parametrize("V", [32064, 32001, 4099, 7])
def test_argmax_vocabulary_rows_ties_and_unaligned_rows(ops, V):
    """The row scan reads 16 bytes per load where the row is aligned (round 6) and one float otherwise: vocabulary-sized rows, ties placed in
    different vector lanes / rounds / the scalar tail, rows that start off a 16-byte boundary (odd widths; a view with a storage offset) —
    always torch.argmax's first maximal index (models/llava.py:297, 352: `torch.argmax(logits, dim=-1)`)."""
    g = torch.Generator().manual_seed(V)
    x = torch.randn(9, V, generator=g)
    x[1, V - 1] = 9.0                                   # the last element (scalar tail when V % 4 != 0)
    x[2, 0] = x[2, V - 1] = 9.0                         # tie between the first and the last
    if V > 4200:
        x[3, 4097] = x[3, 4098] = x[3, 1025] = 9.0      # ties inside one vector and across rounds
        x[4, 4096 * 4 % V] = 9.0
    x[5] = -2.5                                         # all equal: index 0
    x[6, V // 2:] = float("-inf")
    want = torch.argmax(x, -1).tolist()
    assert ops.argmax_rows(x.cuda()).tolist() == want
    big = torch.zeros(9 * V + 3)
    big[3:] = x.reshape(-1)
    y = big.cuda()[3:].view(9, V)                       # same values on the device, every row off the 16-byte grid whatever V is
    assert y.data_ptr() % 16 != 0 and y.is_contiguous()
    assert ops.argmax_rows(y).tolist() == want


MODES = [("llava", DR.MODE_LLAVA_CUMULATIVE), ("next", DR.MODE_NEXT_RESET), ("next_no_overlap", DR.MODE_NEXT_NO_OVERLAP),
         ("iblip", DR.MODE_IBLIP_QUANTILE)]


@pytest.mark.parametrize("fam,mode", MODES)
@pytest.mark.parametrize("rng_mode", ["injected", "mt19937"])
def test_masks_golden(ops, golden_dir, fam, mode, rng_mode):
    g = _load(golden_dir, "g3_masks.npz")
    for c in range(int(g["n_cases"])):
        epi = torch.from_numpy(g[f"c{c}_epi"]).cuda()
        probs = [float(p) for p in g[f"c{c}_probs"]]
        _, keep = ops.get_overlap_image_tokens(torch.from_numpy(g[f"c{c}_step_logits"]).cuda(),
                                               torch.from_numpy(g[f"c{c}_topk_ids"]).cuda())
        if rng_mode == "injected":
            drop, nd, idx = ops.sample_masks(epi, probs, keep, mode, uniforms=torch.from_numpy(g[f"c{c}_uniforms"]).cuda(),
                                             want_indices=True)
        else:
            rng = ops.TorchCpuCompatRNG(int(g[f"c{c}_seed"]))
            drop, nd, idx = ops.sample_masks(epi, probs, keep, mode, rng=rng, want_indices=True)
        start, L = int(g[f"c{c}_start"]), epi.numel()
        ref = g[f"c{c}_{fam}_masks"][:, start:start + L] == 0
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), ref, err_msg=f"case {c} {fam} {rng_mode}")
        np.testing.assert_array_equal(nd.cpu().numpy(), ref.sum(1))
        if fam == "llava":
            np.testing.assert_array_equal(nd.cpu().numpy(), g[f"c{c}_llava_masked_numbers"])
        idx = idx.cpu().numpy()
        for k in range(len(probs)):
            want = np.nonzero(ref[k])[0]
            np.testing.assert_array_equal(idx[k, :len(want)], want)        # ascending dropped indices
            assert (idx[k, len(want):] == -1).all()


@pytest.mark.parametrize("fam,mode", [("llava_no_overlap", DR.MODE_LLAVA_CUMULATIVE_NO_OVERLAP),
                                      ("iblip_no_overlap", DR.MODE_NEXT_NO_OVERLAP)])
@pytest.mark.parametrize("rng_mode", ["injected", "mt19937"])
def test_masks_golden_dormant_no_overlap(ops, golden_dir, fam, mode, rng_mode):
    """The reference's dormant `epis_no_overlap` call (llava.py:663-683 / instructblip.py:486-505) on the g3 inputs."""
    g, g7 = _load(golden_dir, "g3_masks.npz"), _load(golden_dir, "g7_variants.npz")
    for c in range(int(g7["n_cases"])):
        epi = torch.from_numpy(g[f"c{c}_epi"]).cuda()
        probs = [float(p) for p in g[f"c{c}_probs"]]
        if rng_mode == "injected":
            drop, nd = ops.sample_masks(epi, probs, None, mode, uniforms=torch.from_numpy(g[f"c{c}_uniforms"]).cuda())
        else:
            drop, nd = ops.sample_masks(epi, probs, None, mode, rng=ops.TorchCpuCompatRNG(int(g[f"c{c}_seed"])))
        start, L = int(g[f"c{c}_start"]), epi.numel()
        ref = g7[f"c{c}_{fam}_masks"][:, start:start + L] == 0
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), ref, err_msg=f"case {c} {fam} {rng_mode}")
        np.testing.assert_array_equal(nd.cpu().numpy(), ref.sum(1))


@pytest.mark.parametrize("fam,mode", MODES + [("llava_no_overlap", DR.MODE_LLAVA_CUMULATIVE_NO_OVERLAP)])
def test_masks_random_trials_vs_oracle(ops, fam, mode):
    rs = np.random.RandomState(hash(fam) % 1000)
    for trial in range(40):
        L = int(rs.choice([1, 2, 31, 32, 33, 64, 100, 576, 1000, 2928]))
        K = int(rs.randint(1, 9))
        probs = [float(p) for p in rs.choice([0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.25, 0.33], K)]
        epi = torch.from_numpy((rs.rand(L) * 2).astype(np.float32))
        if trial % 5 == 0 and L > 4:
            epi[rs.randint(0, L, L // 2)] = float(epi[0])               # duplicates (quantile ties)
        keep = torch.from_numpy(rs.rand(L) < 0.05)
        uni = torch.from_numpy(rs.rand(K, L).astype(np.float32))
        ref = DR.sample_masks(epi, probs, keep, mode, uni)
        drop, nd = ops.sample_masks(epi.cuda(), probs, keep.cuda(), mode, uniforms=uni.cuda())
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), ref.numpy(), err_msg=f"trial {trial} L={L} probs={probs}")


@pytest.mark.parametrize("fam,mode", MODES)
def test_masks_and_vote_at_the_member_limit_K64(ops, fam, mode):
    """K = DD_MAX_MEMBERS = 64 through the stand-alone operators — 64 probabilities in the kernel's argument table, 64 rows
    of flags and counts (the cumulative rule carries a member's zeros through all 64), a 64-way vote — against the oracle; K = 65 refused."""
    rs = np.random.RandomState(64)
    for L in (33, 576, 2928):
        probs = [float(p) for p in np.linspace(0.1, 0.9, 64)]
        epi = torch.from_numpy((rs.rand(L) * 2).astype(np.float32))
        keep = torch.from_numpy(rs.rand(L) < 0.05)
        uni = torch.from_numpy(rs.rand(64, L).astype(np.float32))
        ref = DR.sample_masks(epi, probs, keep, mode, uni)
        drop, nd = ops.sample_masks(epi.cuda(), probs, keep.cuda(), mode, uniforms=uni.cuda())
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), ref.numpy(), err_msg=f"{fam} L={L}")
        assert nd.cpu().tolist() == [int(r.sum()) for r in ref]
    for _ in range(20):
        ids = rs.randint(0, 6, 64)
        assert ops.select_by_vote(torch.from_numpy(ids).cuda()) == DR.vote(ids.tolist())
    with pytest.raises(Exception):
        ops.sample_masks(torch.rand(10).cuda(), [0.5] * 65, None, mode, uniforms=torch.rand(65, 10).cuda())


def test_mask_rng_stream_continues_across_steps(ops):
    """Two steps of K=3 consume 6 consecutive rand_like(epi) draws of one stream (reference llava.py:650)."""
    L, probs = 100, [0.3, 0.5, 0.7]
    epi = torch.linspace(0, 1, L)
    keep = torch.zeros(L, dtype=torch.bool)
    rng = ops.TorchCpuCompatRNG(24)
    ref = TorchCpuMT19937(24)
    for step in range(2):
        uni = torch.from_numpy(np.stack([ref.rand_f32(L) for _ in probs]))
        want = DR.sample_masks(epi, probs, keep, DR.MODE_LLAVA_CUMULATIVE, uni)
        drop, _ = ops.sample_masks(epi.cuda(), probs, keep.cuda(), DR.MODE_LLAVA_CUMULATIVE, rng=rng)
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), want.numpy())


def test_vote_golden_and_random(ops, golden_dir):
    g = _load(golden_dir, "g4_vote.npz")
    for c in range(int(g["n_cases"])):
        ids = g[f"c{c}_ids"]
        w, t = ops.select_by_vote(torch.from_numpy(ids).cuda())
        assert w == int(g[f"c{c}_winner"]) and t == int(ids[w])
    rs = np.random.RandomState(0)
    for _ in range(50):
        ids = rs.randint(0, 4, rs.randint(1, 17))
        assert ops.select_by_vote(torch.from_numpy(ids).cuda()) == DR.vote(ids.tolist())


def test_bad_arguments_raise(ops):
    with pytest.raises(ValueError):
        ops.sample_masks(torch.zeros(9000).cuda(), [0.3], torch.zeros(9000, dtype=torch.bool).cuda(), 0,
                         uniforms=torch.zeros(1, 9000).cuda())
    with pytest.raises(ValueError):
        ops.calculate_vision_uncertainty(torch.zeros(4, 8))          # CPU tensor: no CPU fallback


@pytest.mark.parametrize("rng_mode", ["injected", "mt19937"])
def test_epis_kl_keep_set_and_masks_golden(ops, golden_dir, rng_mode):
    """dd_kl_keep (lowest_percent_kl_indices, instructblip.py:559-578) and the "epis_kl" masks (:464-485) against what the
    reference's own functions produced; plus a 576 x 32064 case against the oracle."""
    g, g7 = _load(golden_dir, "g3_masks.npz"), _load(golden_dir, "g7_variants.npz")
    for c in range(int(g7["n_cases"])):
        if f"c{c}_kl_lowest" not in g7.files:
            continue
        img = torch.from_numpy(g7[f"c{c}_kl_image_logits"]).cuda()
        step = torch.from_numpy(g[f"c{c}_step_logits"]).cuda()
        idx = ops.lowest_percent_kl_indices(img[None], step[None])
        assert sorted(idx.cpu().tolist()) == sorted(g7[f"c{c}_kl_lowest"].tolist()), f"case {c}"
        keep = torch.zeros(img.shape[0], dtype=torch.uint8, device="cuda")
        keep[idx] = 1
        epi = torch.from_numpy(g[f"c{c}_epi"]).cuda()
        probs = [float(p) for p in g[f"c{c}_probs"]]
        if rng_mode == "injected":
            drop, nd = ops.sample_masks(epi, probs, keep, ops.MASK_IBLIP_KL, uniforms=torch.from_numpy(g[f"c{c}_uniforms"]).cuda())
        else:
            drop, nd = ops.sample_masks(epi, probs, keep, ops.MASK_IBLIP_KL, rng=ops.TorchCpuCompatRNG(int(g[f"c{c}_seed"])))
        start, L = int(g[f"c{c}_start"]), epi.numel()
        np.testing.assert_array_equal(drop.cpu().numpy().astype(bool), g7[f"c{c}_iblip_kl_masks"][:, start:start + L] == 0,
                                      err_msg=f"case {c} {rng_mode}")
    gen = torch.Generator().manual_seed(4)
    img = torch.randn(576, 32064, generator=gen) * 3.0
    step = torch.randn(32064, generator=gen) * 3.0
    want = torch.nonzero(DR.kl_keep(img, step)).flatten().tolist()
    got = ops.lowest_percent_kl_indices(img.cuda(), step.cuda()).cpu().tolist()
    assert sorted(got) == sorted(want) and len(got) == 57
