"""Engine-level regression of round 4's determinism finding (DESIGN.md "Determinism"): an fp32-cache group step whose member sweeps run on TWO
branches — the VALU attention tile pass of one sweep beside the MFMA GEMVs of the other on the same CUs — gives every sequence the bits of its
solo run.  Until round 4 it did not (round 3 kept fp32-cache engines on one branch, "cause not found"): the tile pass's packed FP32
multiply-adds went wrong beside MFMA workgroups (tests/test_gpu_sampler_repro.py holds the unit reproducer); the library is now built without
packed FP32 instructions.  LLaVA-1.5-7B shapes, 12 sequences = one 64-row sweep + one 32-row sweep per step (tools/lanes_mixed_ab.py's case)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fp32_cache_two_branch_group_step_equals_solo_runs():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    shapes = [(608, 5, 576), (640, 9, 576), (600, 1, 576), (615, 20, 576), (609, 5, 576), (700, 60, 576),
              (610, 3, 576), (633, 7, 576), (655, 11, 576), (602, 2, 576), (690, 33, 576), (611, 4, 576)]
    engs = []
    for i in range(len(shapes)):
        engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=768, max_visual=576, seed=5217, kv_format="fp32",
                                     share_weights_with=engs[0] if engs else None))
    engs[0].load_synthetic(1, 0.02)
    embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(50 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
    steps = 4
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.rng.manual_seed(5217)
        e.prefill(x, s0, L)
    grp = lm.EngineGroup(engs)
    rec = [[] for _ in engs]
    for s in range(steps):
        grp.decode_step(probs)                      # step 0 eager, then captured / replayed: two branches in each form
        for i, e in enumerate(engs):
            rec[i].append((e.last_step()["drop"].copy(), e.logits().copy(), e.base_logits().copy()))
    toks = [e.tokens() for e in engs]
    for i, (e, x, (T0, s0, L)) in enumerate(zip(engs, embs, shapes)):
        e.set_speculation("never")
        e.rng.manual_seed(5217)
        e.prefill(x, s0, L)
        for s in range(steps):
            e.decode_step(probs)
            np.testing.assert_array_equal(e.last_step()["drop"], rec[i][s][0], err_msg=f"lane {i} step {s}: masks")
            np.testing.assert_array_equal(e.base_logits(), rec[i][s][2], err_msg=f"lane {i} step {s}: un-masked logits")
            np.testing.assert_array_equal(e.logits(), rec[i][s][1], err_msg=f"lane {i} step {s}: member logits")
        assert e.tokens() == toks[i]
        e.set_speculation("default")
    for e in reversed(engs):
        e.close()
