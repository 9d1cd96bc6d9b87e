"""Unit level of the determinism work (DESIGN.md "Determinism").

THE FINDING (round 4): on gfx950, packed FP32 multiply-adds (v_pk_fma_f32) over operands just read from LDS give wrong sums while a workgroup of
an MFMA kernel is resident on the same CU — `pv_step_probe` below is the unit reproducer (the P.V step of the fp32-cache attention tile pass next to
the same sums as scalar v_fma_f32: 0 of 10^8 lanes differ alone, ~10^6 beside the 64-row slice GEMVs).  The product library is therefore built
without packed FP32 instructions (build.py NO_PACKED_FP32, gated), and the engine-level regression — an fp32-cache group step on two branches
equals every sequence decoded alone — is tests/test_gpu_fp32_branches.py.

Also here: the group step's mask sampler — 8 sequences, L = 576, K = 8, each drawing from its own mt19937 stream — launched back to back on a
stream of its own BESIDE 72-row slice-resident GEMVs looping on two other streams, every launch's keep set, masks, bit planes and counts compared
with the oracle sampler over the host mt19937 (models/llava.py:443-482, 589-662): the product's one-wave form and the two 1,024-thread forms
libdropdec_tools.so keeps.  Also: no kernel of the product library may use private scratch (the build gate), and the sampler's round-3 form — which
did — is only in libdropdec_tools.so.  tools/sampler_repro.py runs the same bodies for many more rounds.

SECOND FINDING (rounds 4-5): the 1,024-thread sampler is a victim of co-residency — on a stream of its own beside a group taking rider steps its
generator block in LDS ends up wrong about once in 50,000 launches (tools/sampler_repro.py sampler_streams; profiles/r04_determinism/,
profiles/r05_sampler_fault/).  Round 5 captured the words: they are the sampler's OWN values of other generations — stores of two of the four
regenerating waves missing for two regenerations in a row, values of the next generation visible before a regeneration's entry barrier — i.e. the
sixteen waves of the workgroup out of step across s_barrier; and a dynamic-LDS request that keeps a second sampler workgroup off the CU (84 KiB)
does not stop it, 120 KiB barely, 156 KiB does.  The product sampler is now ONE WAVE per sequence with no barrier at all
(csrc/dd_dropout.hip); `test_one_wave_sampler_is_clean_where_the_block_form_fails` runs both forms in the failing company."""
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
    assert len(lanes) == 1 and lanes[0]["scratch"] == 0          # (the one-wave sampler; the 1,024-thread forms are tools-only)
    old = [k for k in res["dd_dropout_tools.o"] if "k_sample_masks_lanes_scratch" in k["name"]]
    assert len(old) == 1 and old[0]["scratch"] == 616


@pytest.mark.gpu
@pytest.mark.parametrize("scratch_form", ["wave", "block", "scratch"])
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


def test_product_objects_have_no_packed_fp32_instructions():
    """CPU half of the finding: every object of libdropdec.so disassembles to zero v_pk_{fma,add,mul}_f32 (the build refuses otherwise); the
    probe's own object keeps them."""
    import os
    from dropoutdecoding_amd import build
    build.build()
    bdir = os.path.join(build.HERE, "build")
    for s_ in build.SOURCES:
        assert build.packed_fp32_ops(os.path.join(bdir, s_.replace(".hip", ".o"))) == 0, s_
    assert build.packed_fp32_ops(os.path.join(bdir, "dd_tools.o")) > 0


@pytest.mark.gpu
def test_pv_step_probe_packed_fp32_beside_mfma_workgroups():
    """The unit reproducer.  Alone, packed and scalar sums agree in every lane.  Beside 64-row slice-resident GEMVs (MFMA workgroups that leave room
    for the probe's 256-thread, 30 KiB-LDS workgroups on their CUs) they disagree in about one lane of a hundred on the MI355X boxes of round 4 —
    reported, not asserted: a board / firmware on which the count is 0 simply no longer needs the workaround."""
    import sampler_repro as R
    import torch
    torch.cuda.set_device(0)
    from dropoutdecoding_amd import _lib, build
    build.build()
    lib = _lib.load_tools()
    alone = R.pk_probe(False, 4, lib, pv=True)
    assert alone["lanes_with_wrong_packed_result"] == 0, alone
    beside = R.pk_probe(True, 6, lib, rows=64, pv=True)
    assert beside["company_gemv_launches"] > 0
    plain = R.pk_probe(True, 4, lib, rows=64, pv=False)         # register-fed packed chains: unaffected
    print(f"\n[packed FP32 beside MFMA workgroups] P.V step (LDS-fed v_pk_fma_f32): {beside['lanes_with_wrong_packed_result']} of {beside['lanes_checked']} "
          f"lane results differ from the scalar sums beside {beside['company_gemv_launches']} 64-row GEMV launches (alone: 0 of {alone['lanes_checked']}); "
          f"register-fed chains: {plain['lanes_with_wrong_packed_result']} of {plain['lanes_checked']}")


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


@pytest.mark.gpu
def test_one_wave_sampler_is_clean_where_the_block_form_fails():
    """The in-suite determinism guard.  The sampler on a stream of its own beside 32 sequences taking rider steps — the
    company in which round 4's 1,024-thread kernel draws wrong masks about once in 50,000 launches when it requests only the LDS it uses:
    ~25 s of that form, its wrong launches REPORTED (a board on which the count is 0 does not fail the suite), then ~40 s of the product's
    one-wave form in the same company, ASSERTED clean."""
    import sampler_repro as R
    import torch
    torch.cuda.set_device(0)
    from dropoutdecoding_amd import _lib, build
    build.build()
    lib = _lib.load_tools()
    wave = R.sampler_streams(1, 10 ** 6, lib, 32, form="wave", seconds=40.0)
    block = R.sampler_streams(1, 10 ** 6, lib, 32, lds_kib=0, form="block", seconds=25.0)
    lib.dd_tools_set_tuning(48, 1)
    print(f"\n[sampler beside rider steps] 1,024-thread form, 76 KiB request: {block['sequences_with_a_wrong_launch']} wrong in {block['sampler_launches']} "
          f"launches of 8 workgroups ({block['company_rider_steps']} rider steps beside); one-wave form: {wave['sequences_with_a_wrong_launch']} wrong in "
          f"{wave['sampler_launches']} launches ({wave['company_rider_steps']} rider steps beside)")
    import warnings
    warnings.warn(UserWarning(    # (so that a `pytest -q` log carries the counts: warnings are summarised at the end of the run)
        f"sampler beside rider steps: 1,024-thread form un-fenced {block['sequences_with_a_wrong_launch']} wrong in {block['sampler_launches']} launches; "
        f"one-wave form {wave['sequences_with_a_wrong_launch']} wrong in {wave['sampler_launches']} launches"))
    assert wave["company_rider_steps"] > 100 and wave["sampler_launches"] > 20000, wave
    assert wave["sequences_with_a_wrong_launch"] == 0, wave
