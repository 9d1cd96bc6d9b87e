"""The 7B-shape kernel instantiations against the ORACLE (not against each other).

The nine-plane / rider / half-plane / fp8 slice kernels that carry the headline exist only for K = 4096 / 11008 / 14336
(csrc/dd_gemv.hip try_slices*: `spw == 16 || 43 || 56`); tests/test_gpu_rider.py, test_gpu_half_planes.py and
test_gpu_gemv_slices.py compare those forms bit for bit with each other and with a sequence decoded alone.  This file
anchors them to the reference: two-layer models with the 7B widths (d = 4096, d_ff = 11008 / 14336, 32 heads, MHA / GQA 4,
V = 2048) carry seeded `random_weights` through `load_state_dict`, and EVERY lane is compared with its own `RefDecoder`
(oracle/decode_ref.py — the restatement of the member loop models/llava.py:292-376, llavanext.py:490-600) step by step:

  token ids, keep sets, mask flags, masked counts, member argmax ids, winner   exact
  winner's logits, un-masked logits                                            <= 1e-3 of max |logit| (north star)

A discrete result may only differ where the oracle itself says the decision sits inside the logits tolerance (top-1 minus
top-2 of the argmax in question < 1e-3 of max |logit|, or a uniform closer to its threshold than the two sides' fp32
uncertainties move that threshold — at least the 1e-5 of test_mid_scale_llava_shapes_vs_oracle); such a lane leaves the comparison from that step on, the test reports it and fails
if more than one lane in eight does.  (The streams are seeded, so a given tree either has such a flip or not — no flake.)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights

K8 = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
K4 = [0.1, 0.3, 0.5, 0.7]             # the reference's shipped LLaVA-1.5 list (chair_test/chair_test.py:170)
TOL = 1e-3                            # logits: relative to the largest |logit| (BASELINE north_star)
STD = 0.012                           # d = 4096: q.k / sqrt(128) of a few units, logits of order one


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


@pytest.fixture(scope="module")
def llama_w():
    dims = (2048, 4096, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
    return dims, random_weights(RefCfg(*dims), 61, STD)


@pytest.fixture(scope="module")
def mistral_w():
    dims = (2048, 4096, 14336, 2, 32, 8, 128, 1e-5, 1000000.0)
    return dims, random_weights(RefCfg(*dims), 63, STD)


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


def _top_gap(v):
    v = np.asarray(v, np.float64)
    t = np.partition(v, -2)[-2:]
    return float((t[1] - t[0]) / np.abs(v).max())


def _oracle_lanes(family, rc, w, probs, embs, spans, n_new, seed0=50):
    import time
    t0 = time.time()
    refs = []
    for i, (emb, (s0, L)) in enumerate(zip(embs, spans)):
        r = RefDecoder(family, rc, w, probs, seed=seed0 + i)
        r.tokens_out = r.generate(emb, s0, L, n_new)
        refs.append(r)
    print(f"\n[oracle: {len(refs)} lanes x {n_new} tokens in {time.time() - t0:.0f} s]", end="")
    return refs


def _thresholds(epi, probs):
    """drop probabilities of models/llava.py:641-647 in float64: p = 0.1 + (mprob - 0.1) (e - lo) / (hi - lo)"""
    e = np.asarray(epi, np.float64)
    return 0.1 + (np.asarray(probs, np.float64)[:, None] - 0.1) * ((e - e.min()) / np.ptp(e))[None]


def _mask_margin(r, ref, probs):
    if r.uniforms is None:            # InstructBLIP: drop where e >= quantile(e, 1 - p) (instructblip.py:450-453): how close an uncertainty sits to a threshold
        e = ref.epi.numpy().astype(np.float64)
        thr = np.quantile(e, 1.0 - np.asarray(probs, np.float64))
        d = np.abs(e[None, :] - thr[:, None])
        d[d == 0.0] = np.inf          # (a threshold that IS an element — weight 0 of the interpolation — is the same element on both sides)
        return float(d.min() / np.ptp(e))
    return float(np.abs(r.uniforms - _thresholds(ref.epi.numpy(), probs)).min())


class LaneCheck:
    """One lane's step-by-step comparison with its oracle; `live` turns False at an excused near-tie."""

    def __init__(self, name, eng, ref, probs):
        self.name, self.eng, self.ref, self.probs = name, eng, ref, probs
        self.live, self.excuse, self.worst = True, None, 0.0
        self.topk_note = None           # the image span's top-k id sets differ at a near-tie: only the keep sets can feel it
        self.p_limit = 1e-5             # a uniform this close to its threshold may flip (set from the two sides' uncertainties in prefill())

    def _discrete(self, what, got, want, margin, limit, s):
        if np.array_equal(np.asarray(got), np.asarray(want)):
            return True
        info = f"{self.name} step {s}: {what} differs from the oracle; the oracle's margin at that decision is {margin:.3g}"
        assert margin < limit, info + f" (>= {limit:g}: not a near-tie — wrong result)"
        self.live, self.excuse = False, info
        return False

    def prefill(self):
        e, r = self.eng, self.ref
        err = _relerr(e.logits(), r.prefill_logits[-1].numpy())
        assert err <= TOL, f"{self.name}: prefill logits {err:.3g}"
        err = _relerr(e.image_logits(), r.image_logits.numpy())
        assert err <= TOL, f"{self.name}: image-span logits {err:.3g}"
        np.testing.assert_allclose(e.vision_uncert_dict()["epis_uncert_per_token"][0], r.epi.numpy(), rtol=5e-3, atol=1e-6,
                                   err_msg=f"{self.name}: epistemic uncertainty")
        img = np.sort(r.image_logits.numpy(), axis=1)[:, ::-1]
        k = r.topk_ids.shape[1]
        # the thresholds divide by (max - min) of the uncertainties — a few per cent of their size for these weights — so the last bits
        # of the two sides' fp32 uncertainties move a threshold by up to p_err: a uniform closer than that to its threshold is a tie
        p_err = float(np.abs(_thresholds(e.vision_uncert_dict()["epis_uncert_per_token"][0], self.probs) - _thresholds(r.epi.numpy(), self.probs)).max())
        assert p_err < 2e-4, f"{self.name}: thresholds from the two sides' uncertainties differ by {p_err:.3g}"
        self.p_limit = max(1e-5, 2 * p_err)
        got, want = np.sort(e.topk()[1], 1), np.sort(r.topk_ids.numpy(), 1)
        if not np.array_equal(got, want):
            rows = np.nonzero((got != want).any(axis=1))[0]                                  # the top-k boundary (llava.py:310)
            rgap = ((img[rows, k - 1] - img[rows, k]) / np.abs(img[rows]).max(axis=1)).max()
            assert rgap < TOL, f"{self.name}: top-k ids of image rows {rows.tolist()} differ from the oracle, boundary gap {rgap:.3g}"
            self.topk_note = f"top-k ids of image rows {rows.tolist()} differ at a boundary gap of {rgap:.3g}"

    def step(self, s):
        if not self.live:
            return
        e, r = self.eng, self.ref.records[s]
        st = e.last_step()
        err = _relerr(e.base_logits(), r.base_logits)
        assert err <= TOL, f"{self.name} step {s}: un-masked logits {err:.3g}"
        self.worst = max(self.worst, err)
        keep_margin = 0.0 if self.topk_note else _top_gap(r.base_logits)
        if not self._discrete("keep set (argmax of the un-masked logits in the rows' top-k ids)", st["keep"], r.keep, keep_margin, TOL, s):
            self.excuse += f" ({self.topk_note})" if self.topk_note else ""
            return
        if not self._discrete("mask flags", st["drop"], r.drop, _mask_margin(r, self.ref, self.probs), self.p_limit, s):
            return
        assert st["masked_numbers"].tolist() == r.masked_numbers, f"{self.name} step {s}: masked counts"
        mm = min(r.member_margin) if r.member_margin else 1.0
        if not self._discrete("member argmax ids", st["member_argmax"], r.member_argmax, mm, TOL, s):
            return
        assert st["winner"] == r.winner, f"{self.name} step {s}: winner"
        err = _relerr(e.logits(), r.logits)
        assert err <= TOL, f"{self.name} step {s}: winner's logits {err:.3g}"
        self.worst = max(self.worst, err)

    def tokens(self):
        if self.live:
            assert self.eng.tokens() == self.ref.tokens_out, f"{self.name}: token ids"


def _report(checks, what):
    gone = [c for c in checks if not c.live]
    print(f"\n{what}: {len(checks) - len(gone)} of {len(checks)} lanes compared to the end, worst logits error "
          f"{max(c.worst for c in checks):.2e} (tolerance {TOL:g})")
    for c in gone:
        print("  excused:", c.excuse)
    for c in checks:
        if c.topk_note and c.live:
            print(f"  note: {c.name}: {c.topk_note}; no keep set felt it")
    assert len(gone) * 8 <= len(checks), f"{what}: {len(gone)} lanes left the comparison at near-ties"


def _lanes(E, cfg, n, family, max_visual, **kw):
    import time
    t0 = time.time()
    engines = []
    for i in range(n):
        engines.append(E.DropoutEngine(cfg, family=family, max_seq=max_visual + 96, max_visual=max_visual, seed=50 + i, kv_format="fp16",
                                       share_weights_with=engines[0] if engines else None, **kw))
    print(f"\n[{n} engines created in {time.time() - t0:.0f} s]", end="")
    return engines


def _inputs(n, d, L, seed):
    gen = torch.Generator().manual_seed(seed)
    T0s = [L + 6 + (i % 5) for i in range(n)]
    embs = [torch.randn(T0, d, generator=gen) * 0.5 for T0 in T0s]
    spans = [(2 + (i % 3), L) for i in range(n)]
    return embs, spans


def _group_vs_oracle(E, engines, refs, embs, spans, probs, steps, what):
    import time
    t0 = time.time()
    for i, (e, emb, (s0, L)) in enumerate(zip(engines, embs, spans)):
        e.rng.manual_seed(50 + i)
        e.prefill(emb.cuda(), s0, L)
        e.set_eos([])
    checks = [LaneCheck(f"{what}, lane {i}", e, r, probs) for i, (e, r) in enumerate(zip(engines, refs))]
    for c in checks:
        c.prefill()
    grp = E.EngineGroup(engines)
    for s in range(steps):
        grp.decode_step(probs)
        for c in checks:
            c.step(s)
    for c in checks:
        c.tokens()
    print(f"\n[engine side: {len(engines)} lanes, prefill + {steps} group steps + comparisons in {time.time() - t0:.0f} s]", end="")
    _report(checks, what)


def _solo_vs_oracle(e, ref, emb, span, probs, steps, seed, mode, what):
    e.set_speculation(mode)
    e.rng.manual_seed(seed)
    e.prefill(emb.cuda(), *span)
    e.set_eos([])
    c = LaneCheck(what, e, ref, probs)
    c.prefill()
    for s in range(steps):
        e.decode_step(probs)
        c.step(s)
    c.tokens()
    e.set_speculation("default")
    assert c.live, c.excuse
    return c.worst


def test_rider_step_llama7b_shapes_every_lane_vs_oracle(E, llama_w):
    """(a) 16 lanes, K = 8, the rider form (72-row nine-plane kernels: K = 4096 single slices, K = 11008 `spw == 43`), then two of
    the lanes alone through the 8-row step and the speculative 16-row step.  The span crosses a 64-key attention tile."""
    dims, w = llama_w
    d = dims[1]
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    L, n, steps = 72, 16, 4
    embs, spans = _inputs(n, d, L, 9)
    refs = _oracle_lanes(FAMILY_LLAVA, rc, w, K8, embs, spans, steps + 1)
    engines = _lanes(E, cfg, n, FAMILY_LLAVA, L)
    engines[0].load_state_dict(w)
    _group_vs_oracle(E, engines, refs, embs, spans, K8, steps, "llama-7b shapes, 16 lanes K = 8 (rider form)")
    for li, mode in ((0, "never"), (9, "always")):
        worst = _solo_vs_oracle(engines[li], refs[li], embs[li], spans[li], K8, steps, 50 + li, mode,
                                f"llama-7b shapes, lane {li} alone (speculation {mode})")
        print(f"  lane {li} alone, speculation {mode}: worst logits error {worst:.2e}")
    for e in reversed(engines):
        e.close()


def test_half_plane_rider_k4_llama7b_shapes_every_lane_vs_oracle(E, llama_w):
    """(b) 28 lanes, K = 4: two sequences per operand plane, groups of fourteen in the rider form (seven half planes + two riding
    planes) — BASELINE config 2's step."""
    dims, w = llama_w
    d = dims[1]
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    L, n, steps = 40, 28, 3
    embs, spans = _inputs(n, d, L, 10)
    refs = _oracle_lanes(FAMILY_LLAVA, rc, w, K4, embs, spans, steps + 1)
    engines = _lanes(E, cfg, n, FAMILY_LLAVA, L)
    engines[0].load_state_dict(w)
    _group_vs_oracle(E, engines, refs, embs, spans, K4, steps, "llama-7b shapes, 28 lanes K = 4 (half-plane rider form)")
    for e in reversed(engines):
        e.close()


def test_rider_step_mistral7b_fp8_every_lane_vs_oracle(E, mistral_w):
    """(c) BASELINE config 5's weight format at its widths: d_ff = 14336, GQA 4, fp8 e4m3fn tiles + row scales; nine-plane fp8 slice
    kernels and the chunked K = 14336 kernel.  The oracle runs on the DEQUANTISED weights (scale * q in fp32: the arithmetic the
    kernels restate exactly), LLaVA-NeXT rule (masks reset every step).  Then one lane alone: 8-row fp8 kernels, both step forms."""
    from dropoutdecoding_amd.lm import dequantize_fp8, quantize_fp8
    dims, w = mistral_w
    d = dims[1]
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    wq = {k: (dequantize_fp8(*quantize_fp8(v)) if v.dim() == 2 and "embed_tokens" not in k else v) for k, v in w.items()}
    L, n, steps = 40, 16, 3
    embs, spans = _inputs(n, d, L, 11)
    refs = _oracle_lanes(FAMILY_NEXT, rc, wq, K8, embs, spans, steps + 1)
    del wq
    engines = _lanes(E, cfg, n, FAMILY_NEXT, L, weight_format="fp8")
    engines[0].load_state_dict(w)
    _group_vs_oracle(E, engines, refs, embs, spans, K8, steps, "mistral-7b shapes fp8, 16 lanes K = 8 (rider form)")
    for li, mode in ((12, "always"),):          # (the two-sweep 8-row fp8 step: test_fp8_weight_storage_next_family's kernels, K = 14336 here)
        worst = _solo_vs_oracle(engines[li], refs[li], embs[li], spans[li], K8, steps, 50 + li, mode,
                                f"mistral-7b shapes fp8, lane {li} alone (speculation {mode})")
        print(f"  lane {li} alone, speculation {mode}: worst logits error {worst:.2e}")
    for e in reversed(engines):
        e.close()


def test_solo_mistral7b_bf16_both_step_forms_vs_oracle(E, mistral_w):
    """(d) the 16-bit kernels at d_ff = 14336 (`spw == 56`), GQA 4: one sequence through the two-sweep 8-row step and the speculative
    16-row step, and 16 lanes in the rider form, each lane against its oracle."""
    dims, w = mistral_w
    d = dims[1]
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    L, n, steps = 72, 16, 3
    embs, spans = _inputs(n, d, L, 12)
    refs = _oracle_lanes(FAMILY_NEXT, rc, w, K8, embs, spans, steps + 1)
    engines = _lanes(E, cfg, n, FAMILY_NEXT, L)
    engines[0].load_state_dict(w)
    _group_vs_oracle(E, engines, refs, embs, spans, K8, steps, "mistral-7b shapes bf16, 16 lanes K = 8 (rider form)")
    for li, mode in ((0, "never"),):
        worst = _solo_vs_oracle(engines[li], refs[li], embs[li], spans[li], K8, steps, 50 + li, mode,
                                f"mistral-7b shapes bf16, lane {li} alone (speculation {mode})")
        print(f"  lane {li} alone, speculation {mode}: worst logits error {worst:.2e}")
    for e in reversed(engines):
        e.close()


def test_rider_step_instructblip_rule_llama7b_shapes_every_lane_vs_oracle(E, llama_w):
    """(e) BASELINE config 4's rule at its language model's widths (Vicuna-7B = LLaMA-7B shapes): quantile masks (no random draws), the vote on the
    argmax of the final hidden state, the last member's zeros leaking into the next un-masked row — 16 lanes in the rider form, each against its own
    oracle, then one lane alone.  32 visual tokens at the start of the sequence, as the Q-Former delivers them."""
    dims, w = llama_w
    d = dims[1]
    rc, cfg = RefCfg(*dims), E.LMConfig(*dims)
    L, n, steps = 32, 16, 3
    gen = torch.Generator().manual_seed(13)
    embs = [torch.randn(L + 8 + (i % 5), d, generator=gen) * 0.5 for i in range(n)]
    spans = [(0, L)] * n
    refs = _oracle_lanes(FAMILY_IBLIP, rc, w, K8, embs, spans, steps + 1)
    engines = _lanes(E, cfg, n, FAMILY_IBLIP, L)
    engines[0].load_state_dict(w)
    _group_vs_oracle(E, engines, refs, embs, spans, K8, steps, "llama-7b shapes, InstructBLIP rule, 16 lanes K = 8 (rider form)")
    worst = _solo_vs_oracle(engines[7], refs[7], embs[7], spans[7], K8, steps, 57, "never", "llama-7b shapes, InstructBLIP rule, lane 7 alone")
    print(f"  lane 7 alone: worst logits error {worst:.2e}")
    for e in reversed(engines):
        e.close()
