"""Unit level of the determinism work (DESIGN.md "Determinism"): the group step's mask sampler — 8 sequences, L = 576, K = 8, one
1,024-thread workgroup per sequence, each drawing from its own mt19937 stream — launched back to back on a stream of its own BESIDE
72-row slice-resident GEMVs looping on two other streams (what it met on a branch of the rider step), every launch's keep set, masks,
bit planes and counts compared with the oracle sampler over the host mt19937 (models/llava.py:443-482, 589-662).
Also: no kernel of the product library may use private scratch (the build gate), and the sampler's round-3 form — which did — is only in
libdropdec_tools.so.  tools/sampler_repro.py runs the same bodies for many more rounds (profiles/r04_sampler_repro.json)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_product_kernel_uses_private_scratch():
    """CPU half: the resource remarks of the last build (hipcc -Rpass-analysis=kernel-resource-usage) — every kernel of libdropdec.so
    reports ScratchSize 0; the tools-only variants are where the scratch is."""
    from dropoutdecoding_amd import build
    build.build()
    res = build.kernel_resources()
    assert res, "no resource remarks: was the library built by dropoutdecoding_amd/build.py?"
    product = [o for o in res if not o.endswith("_tools.o") and o != "dd_tools.o"]
    assert len(product) == len(build.SOURCES)
    n = 0
    for o in product:
        for k in res[o]:
            n += 1
            assert k["scratch"] == 0, (o, k)
    assert n > 300
    lanes = [k for k in res["dd_dropout.o"] if "k_sample_masks_lanes" in k["name"]]
    assert len(lanes) == 1 and lanes[0]["scratch"] == 0
    old = [k for k in res["dd_dropout_tools.o"] if "k_sample_masks_lanes_scratch" in k["name"]]
    assert len(old) == 1 and old[0]["scratch"] == 616


@pytest.mark.gpu
@pytest.mark.parametrize("scratch_form", [False, True])
def test_lanes_sampler_beside_72_row_gemvs_matches_the_oracle(scratch_form):
    import sampler_repro as R
    import torch
    torch.cuda.set_device(0)
    from dropoutdecoding_amd import _lib, build
    build.build()
    lib = _lib.load_tools()
    seqs = R._inputs()
    alone = R.sampler(scratch_form, False, 3, lib, seqs)
    assert alone["sequences_with_a_wrong_launch"] == 0, alone
    beside = R.sampler(scratch_form, True, 12, lib, seqs)
    assert beside["company_gemv_launches"] > 0
    assert beside["sequences_with_a_wrong_launch"] == 0, beside
    print(f"\n[sampler, scratch form {scratch_form}] {beside['sampler_launches']} launches of 8 workgroups beside {beside['company_gemv_launches']} "
          f"72-row GEMV launches: all masks equal the oracle's")


@pytest.mark.gpu
def test_private_scratch_survives_beside_other_queues():
    import sampler_repro as R
    import torch
    torch.cuda.set_device(0)
    from dropoutdecoding_amd import _lib, build
    build.build()
    lib = _lib.load_tools()
    out = R.probe(True, 6, lib, wgs=8)
    assert out["mismatching_words"] == 0, out
