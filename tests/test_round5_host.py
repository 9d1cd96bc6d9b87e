"""Host-side logic added in round 5, on CPU: the bench line's `finishing_share` / in-step roofline source, the provenance analysis of the checking
sampler's dumps (tools/sampler_repro.py analyse_dump — the tool that named the sampler fault's words), and the packed-FP32 build gate
failing closed (build.py packed_fp32_ops)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _stats_csv(path, rows):
    path.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n' +
                    "".join(f'"{n}",{c},{c * avg},{avg},1.0,1,2,0.0\n' for n, c, avg in rows))


def test_finishing_share_counts_only_the_nine_plane_kernels_of_a_rider_sweep(tmp_path, monkeypatch):
    bench = importlib.import_module("bench")
    prof = tmp_path / "profiles"
    prof.mkdir()
    _stats_csv(prof / "r09_kernel_stats.csv", [
        ("void k_gemv_slices_seq<9, 4, 16, 3, 0, 2>(SliceArgs)", 100, 48000.0),       # gate/up (streaming)
        ("void k_gemv_slices<8, 9, 4, 16, 0, 1, 1>(SliceArgs)", 100, 20000.0),        # a whole-slice kernel (streaming)
        ("void k_attn_partial16_ride<1, 2>(AttnArgs)", 100, 12000.0),                 # the attention's tile pass (streaming)
        ("void k_gemv_finish4<4, 2, 9, 1>(FinishArgs)", 100, 6000.0),                 # finishing
        ("void k_gemv_finish4<4, 2, 9, 0>(FinishArgs)", 100, 6000.0),
        ("void k_attn_combine_ride<2>(AttnArgs)", 100, 8000.0),
        ("void k_gemv_finish4<4, 2, 4, 1>(FinishArgs)", 100, 1e6),                    # a 4-plane kernel: some other pass, not counted
        ("void k_row_partials(float const*)", 7, 1e6),                               # not a sweep kernel
    ])
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.finishing_share(os.path.join("profiles", "r09_kernel_stats.csv"))
    assert got["streaming_us_per_layer"] == 80.0 and got["finishing_us_per_layer"] == 20.0 and got["share"] == 0.2
    assert got["us_per_layer_by_kernel"]["k_attn_combine_ride<2>"] == 8.0
    assert bench.finishing_share(os.path.join("profiles", "absent.csv")) is None
    _stats_csv(prof / "r09_empty.csv", [("void k_row_partials(float const*)", 7, 1e6)])
    assert bench.finishing_share(os.path.join("profiles", "r09_empty.csv")) is None


def test_finishing_share_of_the_committed_round_5_trace():
    """The number DESIGN.md quotes (0.19 of a sweep's kernel time streams nothing algorithmic) is the one the committed trace gives — of the newest
    round (what bench.py reads) and of round 5's trace by name."""
    bench = importlib.import_module("bench")
    _, stats = bench._profile_files()
    assert os.path.basename(stats) in ("r05_kernel_stats.csv", "r06_kernel_stats.csv")
    assert 0.15 < bench.finishing_share(os.path.join("profiles", "r05_kernel_stats.csv"))["share"] < 0.25
    got = bench.finishing_share(stats)
    assert 0.15 < got["share"] < 0.25, got
    # the name as the library reports it (csrc/dd_gemv.hip NOTE_KERNEL) and as the trace prints it: since round 6 with the slices-per-workgroup
    # argument (the first r06 bench line lost its in-step figures to a name that lacked it)
    src = open(os.path.join(os.path.dirname(bench.__file__), "dropoutdecoding_amd", "csrc", "dd_gemv.hip")).read()
    assert 'NOTE_KERNEL("k_gemv_slices_seq<%d, %d, 16, %d, %d, %d, 2>"' in src
    prof = bench.committed_profile("k_gemv_slices_seq<9, 4, 16, 3, 0, 2, 2>")
    assert 40.0 < prof["stats_avg_us"] < 55.0 and prof["traffic"] and prof["traffic"] > 150e6      # gate/up: 180 MB algorithmic per launch


def test_sweep_traffic_of_the_committed_round_5_profile():
    """`roofline.group_step.memory_side_traffic`: every kernel of a rider sweep's layer from the committed PMC passes — the weights are 405 MB of
    the ~740 MB a layer of a 72-row sweep moves (caches 165, partial sums both ways ~150): DESIGN.md 3 "the step as the memory system sees it"."""
    bench = importlib.import_module("bench")
    pmc, stats = bench._profile_files()
    t = bench.sweep_traffic(stats, pmc, 32, 8, 64)
    assert 600e6 < t["read_bytes_per_layer_sweep"] < 720e6 and 60e6 < t["written_bytes_per_layer_sweep"] < 110e6, t
    assert t["bytes_per_token"] == round((t["read_bytes_per_layer_sweep"] + t["written_bytes_per_layer_sweep"]) * 32 * 8 / 64)
    assert 150 < t["kernel_us_per_layer_sweep"] < 260
    assert bench.sweep_traffic("profiles/absent.csv", pmc, 32, 8, 64) is None


def _dump(shadow, seen, later, src, lds_words=8192 * 3):
    d = np.zeros(16 + 5 * 640 + lds_words, np.uint32)
    d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9] = 1, 5, 3, 7, 1, 1234, 624, 99, lds_words
    d[10], d[11], d[12] = 0x8004A123, 0x15, 0x13
    for i, blk in enumerate((shadow, seen, later, src)):
        d[16 + i * 640:16 + i * 640 + 624] = blk
    return d


def test_analyse_dump_names_the_generation_the_wrong_words_come_from():
    """The round-5 finding in miniature: 64 words of the block hold the sampler's own values one regeneration old; the analysis reports the run,
    the owner stream and generation, and where the wrong values come from."""
    R = importlib.import_module("sampler_repro")
    seeds = [11, 12]
    G = R._generations(12, 12)
    shadow = G[7].copy()
    seen = shadow.copy()
    seen[64:128] = G[6][64:128]
    a = R.analyse_dump(_dump(shadow, seen, seen, G[0]), seeds)
    assert a["failed_checks"] == 1 and a["workgroup"] == 5 and a["member"] == 3 and a["check_site"] == "before a regeneration" and a["xcc_id"] == 5
    assert a["wrong_words"] == 64 and a["wrong_word_runs"] == [[64, 127]]
    assert a["still_wrong_microseconds_later"] == 64 and a["later_read_equals_first_read"]
    assert a["block_in_registers_is"] == {"stream": "test", "seed": 12, "generation": 7}
    p = a["provenance_of_wrong_values"]
    assert p["own_generation_-1_same_index"] == 64 and p["own_generation_+1_same_index"] == 0 and p["zero"] == 0
    assert p["values_found_in_any_generation_of"]["test:12"]["count"] == 64 and "test:11" not in p["values_found_in_any_generation_of"]
    # a transient: the second read, microseconds later, is right again
    b = R.analyse_dump(_dump(shadow, seen, shadow, G[0]), seeds)
    assert b["still_wrong_microseconds_later"] == 0 and not b["later_read_equals_first_read"]
    # a clean dump
    c = R.analyse_dump(_dump(shadow, shadow, shadow, G[0]), seeds)
    assert c["wrong_words"] == 0 and "provenance_of_wrong_values" not in c
    # words of another stream (the company's) are told apart from the sampler's own
    H = R._generations(77, 4)
    seen2 = shadow.copy()
    seen2[0:8] = H[2][0:8]
    e = R.analyse_dump(_dump(shadow, seen2, seen2, G[0]), seeds, company_seeds=[77])
    assert e["wrong_word_runs"] == [[0, 7]] and e["provenance_of_wrong_values"]["values_found_in_any_generation_of"]["company:77"]["count"] == 8
    assert e["provenance_of_wrong_values"]["own_generation_-1_same_index"] == 0


def test_generations_follow_the_oracle_generator():
    """The analysis' table of regenerated blocks is the oracle's stream: block g, tempered, is draws 624 g .. 624 g + 623 of TorchCpuMT19937."""
    R = importlib.import_module("sampler_repro")
    from oracle.mt19937 import TorchCpuMT19937
    G = R._generations(2024, 3)
    y = G[1].astype(np.uint64)
    y ^= y >> 11
    y ^= (y << 7) & 0x9D2C5680
    y ^= (y << 15) & 0xEFC60000
    y ^= y >> 18
    g = TorchCpuMT19937(2024)
    draws = g.raw(2 * 624).astype(np.uint64)
    assert np.array_equal(y & 0xFFFFFFFF, draws[624:])


def test_packed_fp32_gate_fails_closed(tmp_path):
    """build.py packed_fp32_ops: a host-only object (no .hip_fatbin) has nothing to check and counts 0; the probe's object has the instructions;
    anything the disassembler cannot read is an error, not a pass (the gate used to fail open)."""
    from dropoutdecoding_amd import build
    build.build()
    bdir = os.path.join(build.HERE, "build")
    assert build.packed_fp32_ops(os.path.join(bdir, "dd_tp.o")) == 0
    assert build.packed_fp32_ops(os.path.join(bdir, "dd_tools.o")) > 0
    junk = tmp_path / "junk.o"
    junk.write_bytes(b"\x7fELF" + bytes(range(200)))
    with pytest.raises(Exception):
        build.packed_fp32_ops(str(junk))
    with pytest.raises(Exception):
        build.packed_fp32_ops(str(tmp_path / "absent.o"))


@pytest.mark.parametrize("src,product_flags", [("stream_lab.hip", False), ("mall_lab.hip", False), ("seq_lab.hip", True), ("fuse_lab.hip", True),
                                               ("pkfma_war_repro.hip", False)])
def test_lab_tools_still_compile_for_gfx950(tmp_path, src, product_flags):
    """The stand-alone measurement tools of DESIGN.md 3e / 3g cross-compile (seq_lab / fuse_lab include csrc/dd_gemv_slices.h: a change of the product
    kernel's interface shows up here, not on the GPU box)."""
    import shutil
    import subprocess
    from dropoutdecoding_amd import build
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    cmd = [hipcc, f"--offload-arch={build.ARCH}", "-O3", "-std=c++17", *(build.NO_PACKED_FP32 if product_flags else []), "-c",
           os.path.join(ROOT, "tools", src), "-o", str(tmp_path / "x.o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
