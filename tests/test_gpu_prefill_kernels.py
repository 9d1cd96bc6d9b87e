"""Round-4 prefill kernels are re-distributions of the same arithmetic, so they leave the same bits (the prompt forward the reference calls at
models/llava.py:294-303; first-token ensemble :336-337):

* RMSNorm + hi/lo split with sixteen rows per workgroup (whole 1 KiB operand tiles per wave store) against one row per workgroup — the same
  256 partial sums per row in the same butterfly order, the same y = w * (x * rstd) (tools key 45);
* prefill attention (fp16 or fp32 cache; the vision towers' bidirectional form too) with the K / V tiles staged as MFMA operands — split hi + lo once per workgroup instead of once per
  wave — and one or two 16-query blocks per wave, against the kernel that stages fp32 tiles (tools key 46 = 2 / 1 / 0).

Compared after every form of prefill that reaches them: one prompt, prompts of several sequences as one matrix, an extension against the cache
(queries at a non-zero position), the first-token ensemble (masked keys); MHA and GQA; bf16, fp16 and fp8 weights (fp8: the batch of prompts shares one bf16 expansion per matrix).  Engines in libdropdec_tools.so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _state(eng):
    return (eng.image_logits().copy(), eng.logits().copy(), eng.kv_sums().copy(),
            {k: np.asarray(v).copy() for k, v in eng.vision_uncert_dict().items()})


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for k in a[3]:
        assert np.array_equal(a[3][k], b[3][k], equal_nan=True), k


@pytest.mark.parametrize("d,H,Hkv,T0,fmt,kv", [(256, 2, 2, 100, "bf16", "fp16"), (512, 4, 2, 321, "fp16", "fp16"), (512, 4, 1, 203, "bf16", "fp16"),
                                                   (4096, 32, 32, 129, "bf16", "fp16"), (512, 4, 2, 421, "fp8", "fp16"), (512, 4, 2, 300, "bf16", "fp32"),
                                                   (256, 2, 1, 150, "fp16", "fp32")])
def test_prefill_kernel_forms_leave_the_same_bits(d, H, Hkv, T0, fmt, kv):
    from dropoutdecoding_amd import _lib, build
    build.build()
    from dropoutdecoding_amd import lm
    T = _lib.load_tools()
    cfg = lm.LMConfig(2048, d, 2 * d + 256, 2, H, Hkv, 128, 1e-5, 10000.0)
    L, s0 = 64, 7
    engs = []
    for i in range(3):
        engs.append(lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=T0 + 160, max_visual=L, seed=9 + i, kv_format=kv, lib=T,
                                     weight_format=fmt, share_weights_with=engs[0] if engs else None))
    engs[0].load_synthetic(seed=4, std=0.05)
    g = torch.Generator().manual_seed(1)
    embs = [(torch.randn(T0 - 13 * i, d, generator=g) * 3.0).cuda() for i in range(3)]
    more = (torch.randn(70, d, generator=g) * 3.0).cuda()
    eng = engs[0]
    results = {}
    for norm16, attn16 in ((1, 2), (0, 0), (1, 0), (0, 2), (1, 1)):
        T.dd_tools_set_tuning(45, norm16)
        T.dd_tools_set_tuning(46, attn16)
        rec = []
        eng.rng.manual_seed(9)
        eng.prefill(embs[0], s0, L)                                    # one prompt
        rec.append(_state(eng))
        toks = eng.generate(4, mprobs=[0.2, 0.5, 0.8], eos=[])
        rec.append(_state(eng) + (toks,))
        eng.truncate(s0 + L + 3)                                       # queries at a non-zero position against the cache (70 rows: the GEMM path)
        eng.prefill_extend(more)
        rec.append(_state(eng))
        for e in engs:
            e.rng.manual_seed(9)
        lm.prefill_group(engs, embs, [(s0, L)] * 3)                    # three prompts as one matrix, ragged lengths
        rec += [_state(e) for e in engs]
        if (norm16, attn16) == (1, 2):                                 # ... and each of them as its own prefill leaves it (fp8: one expansion per batch)
            for e, x, grouped in zip(engs, embs, rec[-3:]):
                e.rng.manual_seed(9)
                e.prefill(x, s0, L)
                _same(_state(e), grouped)
        eng.rng.manual_seed(9)
        eng.prefill(embs[0], s0, L, first_step_ensemble=True, mprobs=[0.3, 0.6])      # masked keys in the prompt pass
        rec.append(_state(eng))
        results[(norm16, attn16)] = rec
    T.dd_tools_set_tuning(45, 1)
    T.dd_tools_set_tuning(46, 2)
    for e in reversed(engs):
        e.close()
    ref = results[(0, 0)]
    assert all(np.isfinite(r[0]).all() and np.isfinite(r[1]).all() for r in ref)
    for key, rec in results.items():
        for a, b in zip(rec, ref):
            _same(a, b)
            if len(a) > 4:
                assert a[4] == b[4], key


def test_vision_towers_same_bits_with_operand_staged_attention():
    """CLIP tower + projector (heads of 64), the EVA-style tower (heads of 88 at a pitch of 96): tools key 46 = 0 / 1, in a child process whose
    towers live in libdropdec_tools.so (tests/vit_attn_ab.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "vit_attn_ab.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "clip ok" in r.stdout and "eva ok" in r.stdout
