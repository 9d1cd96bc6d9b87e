#!/usr/bin/env python3
"""bench.py — decoded tokens/s of the Dropout-Decoding hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...          # starts its own N ranks (torch.distributed.run, one per GPU) when not launched by one
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --config 4|5          # BASELINE configs 4 / 5 (InstructBLIP-Vicuna-7B, LLaVA-NeXT-Mistral-7B fp8) at engine level

One "step" = one batch of images (default 64 per GPU, lanes over one set of weights) through the drop-in class's `generate()` path: vision
front-end (CLIP-L/14-336; config 4: EVA ViT-g + Q-Former; config 5: CLIP over 5 anyres tiles) + LM prefill (576 visual + 32 prompt
positions for LLaVA-1.5) + uncertainty scorer + `--n-new` (128) decoded tokens per image, each by the K=8 ensemble step (un-masked
pass, masks, 8 masked members in one packed sweep, vote).  Weights are random-init tensors of the real shapes (no network /
checkpoints), data is synthetic; EOS is ignored so every run decodes the same number of tokens.  N > 1: every rank decodes its own images
(the path shards over independent images with no data-path collective) -> "scaling": "weak"; `--mode kshard` instead shards the K
members of ONE stream over the ranks with the RCCL exchange of dropoutdecoding_amd/dist.py.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synthetic_inputs(i: int, vocab: int, image_token: int, prompt_len: int = 32):
    """uint8[336,336,3] image and a 32-token prompt with exactly one <image> id, from default_rng(i)."""
    rng = np.random.default_rng(i)
    img = rng.integers(0, 256, size=(336, 336, 3), dtype=np.uint8)
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    px = ((img.astype(np.float32) / 255.0 - mean) / std).transpose(2, 0, 1)[None]
    ids = rng.integers(3, 31999, size=prompt_len).astype(np.int64)
    ids[0] = 1
    ids[5] = image_token
    return torch.from_numpy(ids)[None], torch.from_numpy(px)


def _host_info():
    """CPU model, physical cores, memory: what the CPU baseline ran on (SURVEY.md 8d asks for it next to the number)."""
    model, sockets = "unknown", set()
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            if ln.startswith("physical id"):
                sockets.add(ln.split(":", 1)[1].strip())
    except OSError:
        pass
    try:
        import psutil
        phys, logical, mem = psutil.cpu_count(logical=False) or os.cpu_count(), os.cpu_count(), psutil.virtual_memory()
        mem_gb, avail_gb = mem.total / 2**30, mem.available / 2**30
    except Exception:
        phys = logical = os.cpu_count()
        mem_gb = avail_gb = 0.0
    numa = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]) if os.path.isdir("/sys/devices/system/node") else 1
    return {"cpu_model": model, "sockets": max(1, len(sockets)), "physical_cores": phys, "logical_cpus": logical,
            "numa_nodes": numa, "mem_total_GiB": round(mem_gb, 1), "mem_available_GiB": round(avail_gb, 1)}


def cpu_baseline(K: int, budget_s: float = 20.0):
    """The oracle in reference-faithful mode (1 + K sequential batch-1 forwards, each on a copied KV cache: the cost
    structure of models/llava.py:292-359) timed on the host cores over the FULL 32-layer LLaVA-1.5-7B shapes at T = 608:
    whole decode steps (>= 2 after a warm-up step), every layer with its own weight memory, torch threads = physical cores.
    The reference's CPU path can run in bf16 or fp32; a 2-layer probe picks the faster dtype on this host."""
    from oracle.decode_ref import FAMILY_LLAVA, RefDecoder
    from oracle.lm_ref import KVCache, LMConfig
    host = _host_info()
    try:
        # glibc hands every > 128 KiB tensor a fresh mmap and unmaps it on free: the (1 + K) KV-cache copies per token then
        # cost page faults + TLB shoot-downs across all threads (seconds per token on a 128-core host).  Keep big blocks on
        # the heap instead — what a CPU deployment gets from tcmalloc / jemalloc.
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)      # M_MMAP_THRESHOLD
        libc.mallopt(-1, (1 << 31) - 1)  # M_TRIM_THRESHOLD
        host["malloc"] = "glibc, M_MMAP_THRESHOLD=1GiB (large tensors reuse heap memory)"
    except Exception:
        host["malloc"] = "glibc defaults"
    torch.set_num_threads(max(1, int(host["physical_cores"])))
    torch.manual_seed(0)
    T, L, d, dff = 608, 576, 4096, 11008
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]

    def make(nl, dt):
        cfg = LMConfig(32064, d, dff, nl, 32, 32, 128, 1e-5, 10000.0)
        w = {"model.embed_tokens.weight": (torch.randn(cfg.vocab_size, d) * 0.02).to(dt)}
        base = {"q": torch.randn(d, d) * 0.02, "g": torch.randn(dff, d) * 0.02, "dn": torch.randn(d, dff) * 0.02}
        for i in range(nl):                       # every layer owns its memory (nothing is cache-resident across layers)
            p = f"model.layers.{i}."
            s = 1.0 + 0.01 * i
            w[p + "input_layernorm.weight"] = torch.ones(d, dtype=dt)
            w[p + "post_attention_layernorm.weight"] = torch.ones(d, dtype=dt)
            for n in ("q", "k", "v", "o"):
                w[p + f"self_attn.{n}_proj.weight"] = (base["q"] * s).to(dt)
            w[p + "mlp.gate_proj.weight"] = (base["g"] * s).to(dt)
            w[p + "mlp.up_proj.weight"] = (base["g"] * (s + 0.5)).to(dt)
            w[p + "mlp.down_proj.weight"] = (base["dn"] * s).to(dt)
        w["model.norm.weight"] = torch.ones(d, dtype=dt)
        w["lm_head.weight"] = (torch.randn(cfg.vocab_size, d) * 0.02).to(dt)
        dec = RefDecoder(FAMILY_LLAVA, cfg, w, probs, seed=5217, dropout=K > 0)
        dec.cache = KVCache([(torch.randn(32, T, 128) * 0.5).to(dt) for _ in range(nl)], [(torch.randn(32, T, 128) * 0.5).to(dt) for _ in range(nl)])
        dec.span_start, dec.L = 5, L
        dec.epi = torch.rand(L)
        dec.topk_ids = torch.randint(0, 32000, (L, 5))
        return dec

    t_all = time.perf_counter()

    # the reference's CPU path runs in either dtype and torch's intra-op threading does not scale monotonically on big hosts
    # (many small ops per layer): probe decode steps of a 2-layer slice over dtype x thread count, run the full model with the best
    phys = max(1, int(host["physical_cores"]))
    cand = sorted({t for t in (8, 16, 32, 64, phys) if t <= phys})
    pr = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        dec = make(2, dt)
        for nt in cand:
            torch.set_num_threads(nt)
            dec.step(17)
            t1 = time.perf_counter()
            dec.step(17)
            pr[(name, nt)] = time.perf_counter() - t1
        del dec
    best, cores = min(pr, key=pr.get)
    torch.set_num_threads(cores)
    probe_txt = ", ".join(f"{k[0]}/{k[1]}t {v:.2f}s" for k, v in sorted(pr.items()))
    need_gib = {"bf16": 13.3, "fp32": 26.5}[best] + 0.7 * (1 + K) * {"bf16": 0.5, "fp32": 1.0}[best] + 4
    layers = 32
    if host["mem_available_GiB"] and host["mem_available_GiB"] < need_gib:
        if best == "fp32" and host["mem_available_GiB"] >= 13.3 + 0.35 * (1 + K) + 4:
            best = "bf16"
        else:
            layers = 8                              # not enough host memory for the whole model: say so in `sample`
    dec = make(layers, torch.bfloat16 if best == "bf16" else torch.float32)
    t_setup = time.perf_counter() - t_all
    dec.step(17)                                    # warm-up step (page faults, thread pool)
    steps = []
    while len(steps) < 2 or (sum(steps) < budget_s and len(steps) < 6):
        t1 = time.perf_counter()
        dec.step(17)
        steps.append(time.perf_counter() - t1)
    mean = sum(steps) / len(steps)
    if layers != 32:
        mean = mean * 32 / layers
    wbytes = 6.607e9 * (2 if best == "bf16" else 4)
    return {"value": round(1.0 / mean, 4), "unit": "tokens/s", "cores": cores, "kind": "port", "host": host,
            "weight_stream_GBs": round((1 + K) * wbytes / mean / 1e9, 1),
            "sample": f"oracle RefDecoder, reference-faithful decode steps ({1 + K} sequential batch-1 forwards, each on a copied KV cache) on the "
                      f"{'FULL 32-layer' if layers == 32 else str(layers) + '-layer slice (host memory too small for 32; scaled x32/' + str(layers) + ') of the'} "
                      f"LLaVA-1.5-7B shapes at T=608, K={K}, torch-CPU {best} with {cores} of {phys} physical cores' worth of threads (the fastest of a "
                      f"2-layer probe over dtype x threads, seconds per step: {probe_txt}); {len(steps)} timed steps after one warm-up: {[round(x, 2) for x in steps]} s "
                      f"(mean {sum(steps) / len(steps):.2f} s); decode only (prefill excluded); setup {t_setup:.0f} s + "
                      f"{time.perf_counter() - t_all - t_setup:.0f} s of timed CPU work"}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N ranks as fresh child processes (one per GPU, RCCL rendezvous on
    127.0.0.1) BEFORE this process makes any GPU call, relay their output (rank 0 prints the JSON line), exit with their status.
    The reference's own multi-GPU shape is N independent jobs over image shards (scripts/run_main_experiments.py:81-86)."""
    import socket
    import subprocess
    from dropoutdecoding_amd import build
    build.build()                                  # hipcc only (no GPU call): the ranks find both libraries built
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _profile_files(tag: str = ""):
    """The newest committed rocprofv3 summaries (profiles/rNN_<tag>pmc_summary.json, rNN_<tag>kernel_stats.csv; tag "c5_" = the config-5
    collection): PMC traffic / MFMA busy and kernel-trace durations.  Other files that merely end in the same words do not count."""
    import glob
    import re
    def newest(suffix):
        c = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)) if re.fullmatch(r"r\d\d_" + re.escape(tag + suffix), os.path.basename(f)))
        return os.path.relpath(c[-1], ROOT) if c else None
    return newest("pmc_summary.json") if not tag else (newest("pmc_summary.json") or _profile_files()[0]), \
        newest("kernel_stats.csv") if not tag else (newest("kernel_stats.csv") or _profile_files()[1])


def committed_profile(kernel: str, tag: str = ""):
    """HBM traffic / MFMA busy / kernel-trace average of `kernel` from the committed profile of the newest round — NOT measured in
    this run (bench.py cannot collect PMCs); the kernel is matched by the name the library reports for what it launched."""
    pmc, stats = _profile_files(tag)
    traffic = mfma_util = stats_avg_us = None
    try:
        for name, v in json.load(open(os.path.join(ROOT, pmc)))["kernels"].items():
            if name.startswith("void " + kernel):
                traffic = v["hbm_read_bytes_per_launch"] + v.get("hbm_write_bytes_per_launch", 0)
                mfma_util = v.get("mfma_util")
    except Exception:
        pass
    try:
        import csv
        for row in csv.DictReader(open(os.path.join(ROOT, stats))):
            if row["Name"].startswith("void " + kernel):
                stats_avg_us = round(float(row["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    # the in-step figures describe the tree the profile was collected on.  tools/collect_profiles.sh writes the hash of the
    # decode kernels' sources beside the summaries (profiles/rNN_sources.json); when it differs from this tree's, the line says so.
    matches, note = None, "no source stamp beside the committed profile (collected before round 6)"
    try:
        import re
        tag_r = re.match(r"profiles/(r\d\d)_", stats or "").group(1)
        stamp = json.load(open(os.path.join(ROOT, "profiles", f"{tag_r}_sources.json")))
        now = decode_sources_hash()
        matches = stamp.get("decode_sources_sha256") == now
        note = ("the committed profile was collected on these decode-kernel sources" if matches else
                f"the decode-kernel sources changed since the committed profile ({tag_r}) was collected: achieved / frac describe that tree; "
                "achieved_isolated / frac_isolated are measured in this run")
    except Exception:
        pass
    return {"traffic": traffic, "mfma_util": mfma_util, "stats_avg_us": stats_avg_us, "pmc_file": pmc, "stats_file": stats,
            "matches_tree": matches, "tree_note": note}


DECODE_SOURCES = ["dd_gemv.hip", "dd_gemv_slices.h", "dd_attn_decode.hip", "dd_lm_kernels.hip", "dd_lm_kernels.h", "dd_lm_device.h", "dd_common.h"]


def decode_sources_hash() -> str:
    """sha256 over the sources of the decode step's kernels (what the committed per-kernel profile figures depend on)."""
    import hashlib
    h = hashlib.sha256()
    for f in DECODE_SOURCES:
        h.update(open(os.path.join(ROOT, "dropoutdecoding_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def finishing_share(stats_file):
    """Share of a rider sweep's kernel time that streams nothing algorithmic — the finishing kernels of the slice GEMVs and the attention's combine —
    per layer, from the committed kernel trace (nine-plane kernels only: every layer of a 72-row sweep launches each of them once)."""
    import csv
    import re
    stream_pat = re.compile(r"void (k_gemv_slices_seq<9,|k_gemv_slices<\d+, 9,|k_gemv_slices_fp8c?<9,|k_attn_partial16_ride<)")
    finish_pat = re.compile(r"void (k_gemv_finish4<\d+, \d+, 9, \d+>|k_attn_combine_ride<)")
    st = fi = 0.0
    per_layer = {}
    try:
        rows = list(csv.DictReader(open(os.path.join(ROOT, stats_file))))
    except Exception:
        return None
    calls = max((int(r["Calls"]) for r in rows if stream_pat.match(r["Name"])), default=0)
    for r in rows:
        name, tot = r["Name"], float(r["TotalDurationNs"])
        if stream_pat.match(name):
            st += tot
            per_layer[name.split("(")[0][5:]] = round(tot / max(calls, 1) / 1e3, 2)
        elif finish_pat.match(name):
            fi += tot
            per_layer[name.split("(")[0][5:]] = round(tot / max(calls, 1) / 1e3, 2)
    if st + fi == 0:
        return None
    return {"share": round(fi / (st + fi), 4), "finishing_us_per_layer": round(fi / max(calls, 1) / 1e3, 2),
            "streaming_us_per_layer": round(st / max(calls, 1) / 1e3, 2), "us_per_layer_by_kernel": per_layer, "file": stats_file,
            "note": "kernel durations inside the running step (other branches share the chip); finishing = k_gemv_finish4 of the four matrices + the "
                    "attention's combine, streaming = the slice GEMVs + the attention's tile pass"}


def sweep_traffic(stats_file, pmc_file, n_layers, sweeps, sequences):
    """What ONE layer of one rider sweep moves between L2 and the memory side, all of its kernels together (weights, caches, partial sums both
    ways, operand planes) — from the committed PMC passes (bytes per launch) and kernel trace (launches per layer) — and what that makes per
    decoded token of a group step.  The per-kernel `roofline.frac` prices ONE kernel by its algorithmic bytes; this is the step as the memory
    system sees it."""
    import csv
    import re
    pat = re.compile(r"void (k_gemv_slices_seq<9,|k_gemv_slices<\d+, 9,|k_gemv_slices_fp8c?<9,|k_attn_partial16_ride<|k_gemv_finish4<\d+, \d+, 9, \d+>|k_attn_combine_ride<)")
    try:
        rows = {r["Name"]: r for r in csv.DictReader(open(os.path.join(ROOT, stats_file)))}
        pmc = json.load(open(os.path.join(ROOT, pmc_file)))["kernels"]
    except Exception:
        return None
    calls = max((int(r["Calls"]) for n, r in rows.items() if re.match(r"void k_gemv_slices_seq<9,|void k_gemv_slices_fp8<9,", n)), default=0)
    rd = wr = us = 0.0
    missing = []
    for n, r in rows.items():
        if not pat.match(n):
            continue
        per_layer = int(r["Calls"]) / max(calls, 1)
        us += float(r["TotalDurationNs"]) / max(calls, 1) / 1e3
        v = pmc.get(n)
        if v is None or v.get("hbm_read_bytes_per_launch") is None:
            missing.append(n.split("(")[0][5:])
            continue
        rd += v["hbm_read_bytes_per_launch"] * per_layer
        wr += v.get("hbm_write_bytes_per_launch", 0) * per_layer
    if not calls or rd == 0:
        return None
    per_tok = (rd + wr) * n_layers * sweeps / sequences
    return {"read_bytes_per_layer_sweep": round(rd), "written_bytes_per_layer_sweep": round(wr), "kernel_us_per_layer_sweep": round(us, 1),
            "bytes_per_token": round(per_tok), "kernels_without_counters": missing, "files": [pmc_file, stats_file],
            "note": "L2 <-> memory-side bytes of every kernel of a rider sweep's layer (PMC FETCH_SIZE / WRITE_SIZE per launch x launches per layer); "
                    "a 64-lane step is layers x sweeps of these — divide by the step time of tools/rider_ab.py for the rate inside the decode steps "
                    "(DESIGN.md 3g: 190.6 GB in 35.5 ms = 5.4 TB/s); GBs_at_value below has the vision front-end and the prefill inside the time"}


def measured_read_ceiling(tools):
    """What a plain streaming READ reaches on THIS box, measured in this run (dd_hbm_read_bench, HIP events on the launch stream): 1 GiB windows
    of a 6 GiB buffer (far beyond the 256 MiB memory-side cache), median of 9.  `peak` stays the 8 TB/s of the spec; this is the ceiling a
    kernel that only reads can be held against (the guide's measured copy: 6.29 TB/s)."""
    import ctypes as C
    try:
        buf = torch.empty(6 << 30, dtype=torch.uint8, device="cuda")
        buf.zero_()
        st = torch.cuda.current_stream().cuda_stream
        vals, off, nbytes = [], 0, 1 << 30
        for _ in range(9):
            off = (off + nbytes + (512 << 20)) % ((6 << 30) - nbytes - 1)
            off -= off % 4096
            g = C.c_float()
            if tools.dd_hbm_read_bench(buf.data_ptr() + off, nbytes, 1, 4096, C.byref(g), st) != 0:
                return None
            vals.append(g.value)
        del buf
        torch.cuda.empty_cache()
        vals.sort()
        return round(vals[len(vals) // 2], 1)
    except Exception:
        return None


def roofline_leg(lm_cfg, family, weight_format, kv_format, T0, L, dom_rows, rows8, wide, profile_tag=""):
    """The dominant kernel (gate/up decode GEMV of the member pass) timed alone with HIP events on its launch stream while cycling over
    the layers' weights, on an engine of its own created through libdropdec_tools.so (the timing hooks are not in the product library)."""
    from dropoutdecoding_amd import _lib, lm
    tools = _lib.load_tools()
    eng = lm.DropoutEngine(lm_cfg, family=family, max_seq=T0 + 80, max_visual=L, kv_format=kv_format, weight_format=weight_format, lib=tools)
    eng.load_synthetic(0, 0.02)
    eng.prefill(torch.randn(T0, lm_cfg.hidden_size, device="cuda") * 0.5, 0 if family == lm.FAMILY_IBLIP else 5, L)
    ms_pair, by = eng.time_gemv(2, dom_rows, 96)                       # the whole GEMV (both kernels when wide)
    ms = eng.time_gemv(2 + 8, dom_rows, 96)[0] if wide else ms_pair    # the streaming kernel alone
    kernel = eng.last_gemv_kernel()
    kinds = {}
    for which, name in ((0, "qkv"), (1, "o_proj"), (3, "down_proj")):
        m2, b2 = eng.time_gemv(which, dom_rows, 96)
        kinds[name] = round(b2 / (m2 * 1e-3) / 1e9, 1)
    kinds["gate_up_incl_finish"] = round(by / (ms_pair * 1e-3) / 1e9, 1)
    rows_cmp = None
    if wide:
        m8, b8 = eng.time_gemv(2, rows8, 96)
        kinds["gate_up_8_rows"] = round(b8 / (m8 * 1e-3) / 1e9, 1)
        rows_cmp = {}
        for r_ in (16, 32, 64) + ((72,) if dom_rows == 72 else ()):        # the same matrix at the other pass widths (streaming kernel alone)
            mr = eng.time_gemv(2 + 8, r_, 96)[0]
            rows_cmp[str(r_)] = {"us": round(mr * 1e3, 2), "frac": round(by / (mr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "us_per_8_rows": round(mr * 1e3 * 8 / r_, 2)}
    sweep_ms = eng.time_sweep(rows8, 5)
    sweep_bytes = eng.algorithmic_bytes(0)
    eng.close()
    measured_read = measured_read_ceiling(tools)
    prof = committed_profile(kernel, profile_tag)
    achieved = by / (ms * 1e-3) / 1e9
    rows_what = ("the members of a group of sequences + the un-masked rows of another group riding in further operand planes" if dom_rows == 72
                 else f"the members of {dom_rows // 8} sequences")
    what = (f"{kernel} (gate/up decode GEMV of a {dom_rows}-row pass = {rows_what}: streams the matrix once, K in slices "
            "resident in LDS; its finishing kernel k_gemv_finish4 adds the slices' partial sums and applies SiLU*up)") if wide \
        else f"{kernel} (gate/up decode GEMV, {rows8} rows)"
    # `achieved` / `frac`: the kernel INSIDE the running step — its average duration in the committed kernel trace of the newest round
    # (profiles/, rocprofv3 --kernel-trace --stats of this same command), the figure that follows from the profile; the same kernel launched
    # alone in a loop, timed live with HIP events, is `achieved_isolated` / `frac_isolated` (no other branch sharing the chip: always higher).
    in_step = by / (prof["stats_avg_us"] * 1e-6) / 1e9 if prof["stats_avg_us"] else None
    return {"bound": "hbm", "kernel": what, "kernel_name": kernel, "achieved": round(in_step if in_step else achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round((in_step if in_step else achieved) / HBM_PEAK_GBS, 4),
            "frac_source": ("in-step average duration from " + str(prof["stats_file"]) + " (inside the step the kernel shares its CUs with the other branches' "
                            "attention and finishing workgroups — since round 5 every one of them fits beside it —, so its own duration is longer than alone "
                            "while the step is shorter: group_step.frac_at_value is the step-level figure)" if in_step
                            else "isolated launches (no committed kernel trace names this kernel)"),
            # the same two rates against what a plain read reaches on THIS box in THIS run (not the 8 TB/s of the spec): the kernel's distance from
            # the memory system's own ceiling
            "measured_read_GBs": measured_read, "measured_read_source": "dd_hbm_read_bench, 1 GiB windows of a 6 GiB buffer, median of 9, measured in this run",
            "frac_of_measured": (round((in_step if in_step else achieved) / measured_read, 4) if measured_read else None),
            "frac_of_measured_isolated": (round(achieved / measured_read, 4) if measured_read else None),
            "profile_matches_tree": prof["matches_tree"], "profile_tree_note": prof["tree_note"],
            "achieved_isolated": round(achieved, 1), "frac_isolated": round(achieved / HBM_PEAK_GBS, 4),
            "finishing_share": finishing_share(prof["stats_file"]) if prof["stats_file"] and dom_rows == 72 else None,
            "traffic": prof["traffic"],
            "traffic_source": {"source": "committed profile", "measured_in_this_run": False, "file": prof["pmc_file"],
                               "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch, FETCH_SIZE x2 (gfx950); "
                                      "matched by the kernel name the library reports (dd_tools_last_gemv_kernel)"},
            "mfma_util": prof["mfma_util"], "rows_per_launch": dom_rows,
            # frac counts the ALGORITHMIC bytes (the weights); the kernel also writes its K-slice partial sums: HBM bytes moved / duration
            "traffic_frac": (round(prof["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if prof["traffic"] else None),
            "gate_up_streaming_kernel_by_rows": rows_cmp, "bytes_per_launch": by, "ms_per_launch": round(ms, 5),
            "kernel_stats_avg_us": prof["stats_avg_us"], "kernel_stats_file": prof["stats_file"],
            # the same kernel inside the running step (committed kernel trace): there it shares the chip with the other branch's kernels of
            # the two-branch group step, so its duration is above the solo launch timed here — the fraction by that duration:
            "frac_in_step": (round(by / (prof["stats_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if prof["stats_avg_us"] else None),
            "gemv_incl_finish": {"ms": round(ms_pair, 5), "GBs": round(by / (ms_pair * 1e-3) / 1e9, 1), "frac": round(by / (ms_pair * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "other_gemv_GBs": kinds,
            "packed_sweep_8_rows": {"ms": round(sweep_ms, 4), "algorithmic_bytes": sweep_bytes, "GBs": round(sweep_bytes / (sweep_ms * 1e-3) / 1e9, 1)}}


def sweep_bytes(lm_cfg, T: float, weight_bytes: float, kv_bytes: float) -> float:
    """HBM bytes of ONE sweep over the LM at context T: the matrices + the K/V cache (SURVEY.md 8d's per-token figure is two of these)."""
    c = lm_cfg
    q, kv = c.num_heads * c.head_dim, c.num_kv_heads * c.head_dim
    params = c.num_layers * ((q + 2 * kv) * c.hidden_size + c.hidden_size * q + 3 * c.hidden_size * c.intermediate_size) + c.vocab_size * c.hidden_size
    return params * weight_bytes + T * c.num_layers * 2 * kv * kv_bytes


def end_to_end(single, lm_cfg, T_mean, weight_bytes, kv_bytes):
    """One-image-at-a-time tokens/s against the HBM roofline, two ways: by SURVEY 8(d)'s algorithmic bytes per token (two sweeps — what
    the reference's step streams) and by the bytes this build actually streamed (one sweep where the speculative step held)."""
    if not single:
        return None
    one = sweep_bytes(lm_cfg, T_mean, weight_bytes, kv_bytes)
    sw = single.get("sweeps_per_step") or 2.0
    tps = single["value"]
    return {"tokens_per_s": tps, "bytes_per_token_algorithmic": round(2 * one), "algorithmic_equivalent_GBs": round(2 * one * tps / 1e9, 1),
            "frac_algorithmic": round(2 * one * tps / 1e9 / HBM_PEAK_GBS, 4),
            "sweeps_per_token": round(sw, 3), "bytes_per_token_streamed": round(sw * one), "streamed_GBs": round(sw * one * tps / 1e9, 1),
            "frac": round(sw * one * tps / 1e9 / HBM_PEAK_GBS, 4),
            "note": "frac = bytes actually streamed per token (sweeps_per_token x (W_lm + T x kv_tok) at the mean context; a failed speculation counts "
                    "its 16-row sweep and its 8-row re-run) x tokens/s / 8 TB/s; frac_algorithmic credits SURVEY 8(d)'s 2 x (W_lm + T x kv_tok) per token "
                    "whether or not both sweeps ran; one image at a time, prefill and vision front-end included in the time"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-new", type=int, default=128)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configs: 3 (default, the metric's: LLaVA-1.5-7B K=8), 2 (LLaVA-1.5-7B K=4 [0.1,0.3,0.5,0.7]), 1 (--original), "
                         "4 (InstructBLIP-Vicuna-7B K=8: EVA ViT-g + Q-Former front-end, 32 visual tokens), 5 (LLaVA-NeXT-Mistral-7B K=8: CLIP over 5 anyres "
                         "tiles, 2928 visual tokens, fp8 weights); each through its drop-in class")
    ap.add_argument("--mode", choices=["replicas", "kshard"], default="replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--original", action="store_true", help="stock greedy decode (K=1, no dropout), BASELINE configs[0]")
    ap.add_argument("--images-per-gpu", type=int, default=None,
                    help="images decoded concurrently per GPU (lanes over one set of weights, 1..64); 1 = the reference's "
                         "one-image-at-a-time loop; default 64 (config 2: 56)")
    ap.add_argument("--no-batch-tower", action="store_true", help="one vision-tower call per image instead of one per 16 images (A/B)")
    ap.add_argument("--prefill-chunk", type=int, default=None, help="prompts per LM prefill pass (dd_lm_prefill_group); 1 = one prefill per image")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="dd_set_tuning(key, value) before the run (product switches)")
    ap.add_argument("--single-images", type=int, default=5, help="images of the one-image-at-a-time legs (after one warm-up image)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the isolated-kernel leg (libdropdec_tools.so)")
    ap.add_argument("--no-determinism-check", action="store_true", help="skip the warm-up's repeat of batch 0 from the same seeds")
    ap.add_argument("--no-build", action="store_true", help="never compile: raise when libdropdec.so is stale (for runs under a profiler)")
    args = ap.parse_args()
    if args.config == 1:
        args.original = True
    if args.config == 2:
        args.k = 4

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return spawn_ranks(args)                    # nothing in this process has touched the GPU yet

    # Build (or check) the libraries BEFORE this process touches the GPU: hipcc is a child process tree, and on this pool a process that has
    # initialised the GPU — under rocprofv3 that is every process, from its first instruction — must not start one.  Every rank
    # calls it: build.py serialises the builders with a file lock, the later ones find the tree fresh.  --no-build / DD_NO_BUILD=1: raise instead.
    from dropoutdecoding_amd import build
    if args.no_build:
        os.environ["DD_NO_BUILD"] = "1"
    build.build()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    # DD_BENCH_SHARE_DEVICE=1: every rank on cuda:0 over "gloo" — the only way a ONE-GPU box can run the world > 1 code paths (rank-private
    # image shards, the barriers, all_reduce(MAX) of the time, rank 0's JSON relay, K-shard's exchange); RCCL refuses two ranks on one
    # device.  Not a measurement of scaling: the line says so (`shared_device`).
    share_dev = os.environ.get("DD_BENCH_SHARE_DEVICE", "0") not in ("", "0")
    backend = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev_index = 0 if share_dev else local
        torch.cuda.set_device(dev_index)
        if share_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        backend = dist.get_backend()
    else:
        torch.cuda.set_device(0)

    if use_dist:
        torch.distributed.barrier()
    from dropoutdecoding_amd import _lib as _ddlib
    from dropoutdecoding_amd import config as ddcfg
    from dropoutdecoding_amd import lm

    for kv in args.tune:
        k_, v_ = kv.split("=")
        _ddlib.check(_ddlib.load().dd_set_tuning(int(k_), int(v_)), "dd_set_tuning")
    probs = (ddcfg.VOTING_NUMBERS_K4 if args.k == 4 else ddcfg.VOTING_NUMBERS_K8[:args.k]) if args.k <= 8 else [0.1 + 0.05 * i for i in range(args.k)]
    ddcfg.settings["voting_numbers"] = list(probs)
    K_eff = 0 if args.original else len(probs)
    if args.images_per_gpu is None:
        args.images_per_gpu = 56 if args.config == 2 else 64     # K = 4: whole groups of fourteen
    if args.prefill_chunk is None:
        args.prefill_chunk = 4 if args.config == 5 else 16     # config 5: 4 x 3072 rows per pass over the (fp8-expanded) weights
    B = 1 if args.mode == "kshard" else max(1, min(64, args.images_per_gpu))

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------------------------------
    # the workload: `run_steps(first, n)` = n batches of B images / sequences; `one_stream(i)` = one image / sequence alone
    # ------------------------------------------------------------------------------------------------------------------
    kshard = None
    from dropoutdecoding_amd.vlm import GroupPipeline
    wname, weight_bytes = "bf16", 2.0
    if args.config == 4:
        from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration
        model = CustomInstructBlipForConditionalGeneration.from_synthetic(max_new_tokens=args.n_new + 8)
        family, L, model_name, front = lm.FAMILY_IBLIP, 32, "InstructBLIP-Vicuna-7B", "EVA ViT-g/14 + Q-Former front-end + "
    elif args.config == 5:
        from dropoutdecoding_amd.llavanext import CustomLlavaNextForConditionalGeneration
        model = CustomLlavaNextForConditionalGeneration.from_synthetic(max_new_tokens=args.n_new + 8)
        family, L, model_name, front = lm.FAMILY_NEXT, 2928, "LLaVA-NeXT-Mistral-7B", "CLIP-L/14-336 over 5 anyres tiles + "
        wname, weight_bytes = "fp8", 1.0
    else:
        from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration
        model = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=args.n_new + 8)
        family, L, model_name, front = lm.FAMILY_LLAVA, 576, "LLaVA-1.5-7B", "CLIP-L/14-336 front-end + "
    model.original = args.original
    if args.no_batch_tower:
        from dropoutdecoding_amd.vlm import DropoutVLM
        type(model)._visual_embeds_batch = DropoutVLM._visual_embeds_batch
    eng = model.engine
    lm_cfg, prompt_len = eng.cfg, 32
    T0 = L + prompt_len - (0 if args.config == 4 else 1)       # LLaVA families: the <image> placeholder is replaced by the L tokens
    if args.mode == "kshard" and use_dist:
        from dropoutdecoding_amd.dist import KShardDecoder
        kshard = model.kshard = KShardDecoder(eng, rank, world, time_exchange=True)
    pipe = None
    if B > 1:
        # batches back to back: while one set of lanes decodes, the next batch's CLIP + prefill run on a second stream
        pipe = GroupPipeline(model, lanes=B)
        pipe.prefill_chunk = max(1, args.prefill_chunk)
        if args.config == 5:
            pipe.tower_chunk = 6          # two tower calls of three 5-tile images each (16 tiles per call), ahead of 4-prompt prefill passes

    def batch_inputs(i, n=B):
        out = []
        for b in range(n):
            if args.config == 4:        # InstructBLIP: 224 x 224 image, instruction ids for the Q-Former, prompt ids for the LM (no placeholder)
                rng = np.random.default_rng(i * B + b)
                px = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32))
                ids = torch.from_numpy(rng.integers(3, 31999, size=prompt_len).astype(np.int64))[None]
                qids = torch.from_numpy(rng.integers(1000, 30000, size=12).astype(np.int64))[None]
                out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda(), qformer_input_ids=qids.cuda(), qformer_attention_mask=torch.ones_like(qids).cuda()))
                continue
            ids, px = synthetic_inputs(i * B + b, eng.cfg.vocab_size, model.image_token_index)
            if args.config == 5:        # LLaVA-NeXT anyres: a 672 x 672 image = the base view + 2 x 2 tiles of 336 x 336
                rng = np.random.default_rng(7_000_000 + i * B + b)
                px = torch.from_numpy(rng.standard_normal((1, 5, 3, 336, 336), dtype=np.float32))
                out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda(), image_sizes=torch.tensor([[672, 672]])))
                continue
            out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda()))
        return out

    def one_stream(i):
        kw = batch_inputs(i, 1)[0]
        o = model.generate(**kw, max_new_tokens=args.n_new, eos_token_id=[])
        assert o.shape[1] == (1 if args.config == 4 else kw["input_ids"].shape[1]) + args.n_new      # InstructBLIP returns BOS + new ids

    def run_steps(first, n):
        if pipe is None:
            for i in range(n):
                one_stream(first + i)
            return
        done = 0
        for outs in pipe.run((batch_inputs(first + i) for i in range(n)), max_new_tokens=args.n_new, eos_token_id=[]):
            assert len(outs) == B and all(o.shape[1] == (1 if args.config == 4 else prompt_len) + args.n_new for o in outs)
            done += 1
        assert done == n

    img0 = rank * 10_000 if args.mode == "replicas" else 0

    def determinism_check():
        """Batch `img0` decoded twice from the same generator seeds, token ids compared (inside the warm-up: costs one batch).  The repetition
        stress of the overlapped pipeline is tools/stress_pipeline.py (profiles/r05_stress.jsonl); this is the per-run smoke of it."""
        lanes = pipe.sets[0] + pipe.sets[1]

        def once():
            torch.cuda.synchronize()
            for m in lanes:
                m.engine.rng.manual_seed(5217)
            torch.cuda.synchronize()
            outs = list(pipe.run(iter([batch_inputs(img0)]), max_new_tokens=args.n_new, eos_token_id=[]))
            return [o[0].tolist() for o in outs[0]]
        a, b = once(), once()
        differing = [i for i in range(len(a)) if a[i] != b[i]]
        return {"batches_compared": 1, "images": len(a), "tokens_per_image": args.n_new, "images_with_different_tokens": len(differing),
                "same_tokens": not differing, "first_differing_images": differing[:8]}

    det = None
    if pipe is not None and rank == 0 and args.warmup > 0 and not args.no_determinism_check:
        det = determinism_check()
    run_steps(img0, args.warmup)
    if kshard is not None:
        kshard.exchange_ms(reset=True)
    barrier()
    t0 = time.perf_counter()
    run_steps(img0 + args.warmup, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    exchange = kshard.exchange_ms(reset=True) if kshard is not None else None

    # ---- one image at a time (the reference's own loop: chair_test.py:274-346): adaptive speculation, always, never -------------
    def single_leg(mode, first, images):
        eng.set_speculation(mode)
        one_stream(first)                           # warm-up image (graphs of this mode captured)
        eng.spec_stats(reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(images):
            one_stream(first + 1 + i)
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t1) / images
        st = eng.spec_stats(reset=True)
        eng.set_speculation("default")
        return {"value": round(args.n_new / t1, 2), "unit": "tokens/s", "ms_per_image": round(t1 * 1e3, 1), "images": images,
                "speculation": mode, "speculation_hit_rate": None if st["hit_rate"] is None else round(st["hit_rate"], 3),
                "steps": {"held": st["held"], "rerun": st["rerun"], "plain_two_sweep": st["plain"], "policy_switched_off": st["switched_off"]},
                "sweeps_per_step": None if st["sweeps_per_step"] is None else round(st["sweeps_per_step"], 3)}

    single = single_two = single_keep = None
    if B > 1 and rank == 0 and args.single_images > 0 and args.mode == "replicas":
        single = single_leg("adaptive", img0 + 900, args.single_images)
        single["note"] = ("one image at a time: " + front + "prefill + scorer + ensemble steps; a step is ONE sweep over the weights when the masks "
                          "sampled for an empty keep set stand (speculative step, exact), two otherwise; the library's adaptive policy stops speculating "
                          "while fewer than about one in three checks hold.  RANDOM weights almost never predict a token that is in a visual token's "
                          "top-k list, so the keep set is nearly always empty and the hit rate here is far above a trained checkpoint's: see "
                          "single_stream_two_sweep (what every step costs when the speculation never holds) and single_stream_nonempty_keep_sets")
        if not args.original:
            single_two = single_leg("never", img0 + 920, max(2, args.single_images // 2))
            single_two["note"] = "the same with speculation off: un-masked sweep, then the packed member sweep, every step (the reference's order)"
    if use_dist:
        torch.distributed.barrier()
        t = torch.tensor([dt], device="cpu" if backend == "gloo" else "cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    streams = world if args.mode == "replicas" else 1
    tokens = streams * args.steps * args.n_new * B
    value = tokens / dt

    rows8 = min(max(K_eff, 1), 8)
    wide = B > 1 and 1 <= K_eff <= 8
    dom_rows = (64 if B >= 8 else (32 if B >= 4 else 16)) if wide else rows8
    # rider form of the group step (csrc/dd_engine.hip): from 16 sequences on (whole groups of eight, 16-bit weights) the un-masked rows ride in
    # a ninth operand plane of the member sweeps
    # K <= 4 (BASELINE config 2, the reference's own settings): the members of a sequence fill half an operand plane, a 64-row member sweep
    # carries sixteen sequences (classic form: one fused un-masked sweep + B / 16 member sweeps)
    half_planes = wide and B >= 16 and K_eff <= 4 and wname != "fp8"
    rider_hp = half_planes and B >= 28 and B % 14 == 0        # whole groups of fourteen: seven half planes + two riding planes per sweep
    # (fp8 tiles: nine-plane kernels exist for Mistral-7B's shapes — config 5)
    rider = wide and B >= 16 and B % 8 == 0 and (wname != "fp8" or args.config == 5) and not half_planes
    if rider or rider_hp:
        dom_rows = 72
    roof = None
    if rank == 0 and not args.no_roofline:
        try:
            roof = roofline_leg(lm_cfg, family, wname, "fp16", T0, L, dom_rows, rows8, wide, "c5_" if args.config == 5 else "")
        except Exception as e:                                         # the throughput number must still be reported
            roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "error": f"{type(e).__name__}: {e}"}
    if rank == 0 and single and args.config in (1, 2, 3) and not args.original:
        # the keep sets a trained checkpoint has: 64 lm_head rows scaled up make the un-masked argmax one of those ids at every step and
        # fill the visual tokens' top-k lists with them, so tens of visual tokens are kept per step and the check fails nearly always.
        # (Changes the weights: last leg of the run.)
        try:
            g = torch.Generator(device="cuda").manual_seed(3)
            W = torch.randn(lm_cfg.vocab_size, lm_cfg.hidden_size, device="cuda", generator=g) * 0.02
            W[1000:1064] *= 8.0
            eng._load(lm.T_LM_HEAD, 0, W)
            del W
            single_keep = single_leg("adaptive", img0 + 940, max(2, args.single_images // 2))
            single_keep["note"] = ("lm_head with 64 rows scaled x8: every step keeps tens of visual tokens, the speculation's check fails, the adaptive "
                                   "policy falls back to two-sweep steps and re-probes every 32 steps")
        except Exception as e:
            single_keep = {"value": None, "error": f"{type(e).__name__}: {e}"}
    if roof is not None and roof.get("achieved") and wide:
        # what a group step of B sequences streams: rider form = B / 8 sweeps over the matrices (classic: one fused un-masked sweep + the member
        # sweeps); every sequence's cache is read twice (un-masked row, members).  `value` includes the vision front-end and the prefill, so
        # the rate below is a lower bound on what the decode steps sustain.
        T_mean = T0 + args.n_new / 2
        w_only = sweep_bytes(lm_cfg, 0, weight_bytes, 2.0)
        kv_seq = sweep_bytes(lm_cfg, T_mean, weight_bytes, 2.0) - w_only
        sweeps = B // 8 if rider else (B // 14 if rider_hp else 1 + -(-B // (16 if half_planes else dom_rows // 8)))
        per_tok = (sweeps * w_only + 2 * B * kv_seq) / B
        per_gpu = value / max(1, streams)
        roof["group_step"] = {"sequences": B, "form": "rider" if rider else ("rider, half planes" if rider_hp else ("classic, half planes" if half_planes else "classic")), "sweeps_per_step": sweeps, "rows_per_sweep": dom_rows,
                              "bytes_per_token_streamed": round(per_tok), "bytes_per_token_algorithmic": round(2 * (w_only + kv_seq)),
                              "streamed_GBs_at_value": round(per_tok * per_gpu / 1e9, 1), "frac_at_value": round(per_tok * per_gpu / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "roofline.frac is per LAUNCH of the dominant kernel by its weight bytes: a 72-row launch (rider form) carries 9/8 of a "
                                      "64-row launch's rows and takes ~1.13x its time (gate_up_streaming_kernel_by_rows.us_per_8_rows), but a step needs one "
                                      "sweep fewer; frac_at_value = bytes the steps stream per token x tokens/s per GPU / 8 TB/s, with the vision front-end "
                                      "and the prefill inside the time"}
        if rider and roof.get("kernel_stats_file"):
            tr = sweep_traffic(roof["kernel_stats_file"], roof["traffic_source"]["file"], lm_cfg.num_layers, sweeps, B)
            if tr:
                # PMC bytes of a layer of one sweep / its algorithmic bytes (the layer's matrices once + the sixteen caches a rider sweep reads:
                # eight groups' members and eight riding rows) — 1.0 would mean nothing but weights and caches moves
                alg_layer = (w_only - lm_cfg.vocab_size * lm_cfg.hidden_size * weight_bytes) / lm_cfg.num_layers + 16 * kv_seq / lm_cfg.num_layers
                tr["algorithmic_bytes_per_layer_sweep"] = round(alg_layer)
                roof["step_traffic_ratio"] = tr["step_traffic_ratio"] = round((tr["read_bytes_per_layer_sweep"] + tr["written_bytes_per_layer_sweep"]) / alg_layer, 3)
                tr["GBs_at_value"] = round(tr["bytes_per_token"] * per_gpu / 1e9, 1)
                tr["frac_at_value"] = round(tr["bytes_per_token"] * per_gpu / 1e9 / HBM_PEAK_GBS, 4)
            roof["group_step"]["memory_side_traffic"] = tr
    if roof is not None and single:
        roof["end_to_end"] = end_to_end(single, lm_cfg, T0 + args.n_new / 2, weight_bytes, 2.0)
        if single_two:
            roof["end_to_end_two_sweep"] = end_to_end(single_two, lm_cfg, T0 + args.n_new / 2, weight_bytes, 2.0)

    if rank == 0:
        kv_note = "fp16 KV cache = the reference's cache width"
        wl = (f"{model_name} Dropout Decoding, {B} synthetic " + ({4: "224x224", 5: "672x672 (anyres: 5 tiles of 336x336)"}.get(args.config, "336x336") + " image(s)") + f" per step and GPU -> each {L} visual tokens + "
              f"{prompt_len}-token prompt (prefill {T0}), {args.n_new} decoded tokens each (EOS ignored), K={K_eff} voting_numbers={list(probs) if K_eff else []}, "
              f"random-init weights of the real shapes ({wname} weights, fp32 activations, {kv_note})"
              + (f"; the {B} are independent sequences (own KV cache and rng stream, results identical to decoding each alone) whose "
                 + ("member passes run 8 sequences (64 rows) per sweep over the weights, each sweep carrying the un-masked rows of 8 other sequences "
                    f"in a ninth operand plane (72 rows; no sweep of their own: {B // 8} sweeps per step)" if rider else
                    (f"member passes run 14 sequences per sweep over the weights (K <= 4 members fill half an operand plane: seven planes of two "
                     f"sequences), each sweep carrying the un-masked rows of 14 other sequences in two more planes (72 rows; {B // 14} sweeps per step)"
                     if rider_hp else None) or
                    (f"un-masked passes share one sweep over the weights and whose member passes run 16 sequences (64 rows: K <= 4 members fill "
                     f"half an operand plane, two sequences share one) per sweep: 1 + {-(-B // 16)} sweeps per step" if half_planes else
                     f"un-masked passes share one sweep over the weights and whose member passes run {dom_rows // 8} sequences ({dom_rows} rows) per sweep"))
                 + "; the next batch's vision front-end + prefill overlap the current batch's decode on a second stream" if B > 1 else ""))
        metric = {1: f"decoded tokens/sec {model_name} --original", 2: f"decoded tokens/sec {model_name} K=4 ensemble", 3: f"decoded tokens/sec {model_name} K=8 ensemble",
                  4: f"decoded tokens/sec {model_name} K=8 ensemble", 5: f"decoded tokens/sec {model_name} K=8 ensemble, fp8 weights"}[args.config]
        if args.original:
            metric = f"decoded tokens/sec {model_name} --original"
        elif K_eff not in (4, 8) or (args.config == 3 and K_eff != 8):
            metric = f"decoded tokens/sec {model_name} K={K_eff} ensemble"
        line = {
            "metric": metric, "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak" if args.mode == "replicas" else "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": wl, "baseline_config": args.config,
                       "batch_note": (f"`value` is the aggregate over {B} independent sequences decoded concurrently per GPU (throughput mode, the "
                                      "reference's multi-process sharding on one GPU); the reference's own shape, one image at a time, is `single_stream`"
                                      if B > 1 else "one image at a time"),
                       "mode": args.mode, "images_per_step_per_gpu": B, "n_new": args.n_new, "K": K_eff,
                       "prefill_included": True, "vision_front_end_included": True, "device_bytes": eng.device_bytes},
            "single_stream": single, "single_stream_two_sweep": single_two, "single_stream_nonempty_keep_sets": single_keep,
            "roofline": roof, "determinism_check": det,
        }
        if use_dist:
            # what the process group itself reports (the driver computes scaling from `value` per N; nothing is estimated here)
            line["process_group"] = {"backend": backend, "ranks": torch.distributed.get_world_size(), "rccl_ranks": torch.distributed.get_world_size() if backend == "nccl" else 0,
                                     "shared_device": share_dev,
                                     "note": ("all ranks on ONE GPU over gloo: exercises the world > 1 code paths, not a scaling measurement" if share_dev
                                              else "one rank per GPU; backend nccl = RCCL over xGMI")}
        if exchange is not None:
            line["kshard_exchange"] = exchange
        if world == 1 and not args.no_cpu_baseline and args.config in (1, 2, 3):   # the CPU leg restates config 3's shapes
            try:
                line["cpu_baseline"] = cpu_baseline(max(K_eff, 1) if not args.original else 0)
            except Exception as e:                                     # the GPU number must still be reported
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(line), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
