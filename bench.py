#!/usr/bin/env python3
"""bench.py — decoded tokens/s of the Dropout-Decoding hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one image through `CustomLlavaForConditionalGeneration.generate()`: CLIP-L/14-336 front-end +
prefill of 608 positions (576 visual + 32 prompt tokens) + uncertainty scorer + `--n-new` (128) decoded tokens,
each by the K=8 ensemble step (un-masked pass, masks, 8 masked members in one packed sweep, vote).  Weights are
random-init tensors of the real LLaVA-1.5-7B shapes (no network / checkpoints), data is synthetic; EOS is ignored
so every run decodes the same number of tokens.  N > 1: every rank decodes its own images (the path shards over
independent images with no data-path collective) -> "scaling": "weak"; `--mode kshard` instead shards the K
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


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-new", type=int, default=128)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--mode", choices=["replicas", "kshard"], default="replicas")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--original", action="store_true", help="stock greedy decode (K=1, no dropout), BASELINE configs[0]")
    ap.add_argument("--images-per-gpu", type=int, default=32,
                    help="images decoded concurrently per GPU (lanes over one set of weights, 1..64); 1 = the reference's "
                         "one-image-at-a-time loop")
    ap.add_argument("--no-batch-tower", action="store_true", help="one vision-tower call per image instead of one per 16 images (A/B)")
    ap.add_argument("--prefill-chunk", type=int, default=16, help="prompts per LM prefill pass (dd_lm_prefill_group); 1 = one prefill per image")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="dd_set_tuning(key, value) before the run (experiments)")
    ap.add_argument("--single-images", type=int, default=5, help="images of the one-image-at-a-time leg (after one warm-up image)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)

    from dropoutdecoding_amd import build
    if local == 0:
        build.build()                      # one builder per node; the others wait (no concurrent writes of the .so)
    if use_dist:
        torch.distributed.barrier()
    from dropoutdecoding_amd import config as ddcfg
    from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration

    for kv in args.tune:
        from dropoutdecoding_amd import _lib as _ddlib
        k_, v_ = kv.split("=")
        _ddlib.check(_ddlib.load().dd_set_tuning(int(k_), int(v_)), "dd_set_tuning")
    probs = ddcfg.VOTING_NUMBERS_K8[:args.k] if args.k <= 8 else [0.1 + 0.05 * i for i in range(args.k)]
    ddcfg.settings["voting_numbers"] = probs
    model = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=args.n_new + 8)
    model.original = args.original
    if args.no_batch_tower:
        from dropoutdecoding_amd.vlm import DropoutVLM
        type(model)._visual_embeds_batch = DropoutVLM._visual_embeds_batch
    eng = model.engine
    if args.mode == "kshard" and use_dist:
        from dropoutdecoding_amd.dist import KShardDecoder
        model.kshard = KShardDecoder(eng, rank, world)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    B = 1 if args.mode == "kshard" else max(1, min(64, args.images_per_gpu))
    lanes = [model] + [model.spawn_lane() for _ in range(B - 1)]
    from dropoutdecoding_amd.vlm import generate_group

    def one(i, n_lanes=B):
        """one step = one batch of n_lanes images through generate(): CLIP + prefill each, then all decoded together"""
        batch = []
        for b in range(n_lanes):
            ids, px = synthetic_inputs(i * B + b, eng.cfg.vocab_size, model.image_token_index)
            batch.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda()))
        if n_lanes == 1:
            outs = [model.generate(**batch[0], max_new_tokens=args.n_new, eos_token_id=[])]
        else:
            outs = generate_group(lanes[:n_lanes], batch, max_new_tokens=args.n_new, eos_token_id=[])
        for o, kw in zip(outs, batch):
            assert o.shape[1] == kw["input_ids"].shape[1] + args.n_new
        return outs

    img0 = rank * 10_000 if args.mode == "replicas" else 0
    pipe = None
    if B > 1:
        # batches back to back: while one set of lanes decodes, the next batch's CLIP + prefill run on a second stream
        from dropoutdecoding_amd.vlm import GroupPipeline
        pipe = GroupPipeline(model, lanes=B)
        pipe.prefill_chunk = max(1, args.prefill_chunk)

    def batch_inputs(i):
        out = []
        for b in range(B):
            ids, px = synthetic_inputs(i * B + b, eng.cfg.vocab_size, model.image_token_index)
            out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda()))
        return out

    def run_steps(first, n):
        if pipe is None:
            for i in range(n):
                one(first + i)
            return
        done = 0
        for outs in pipe.run((batch_inputs(first + i) for i in range(n)), max_new_tokens=args.n_new, eos_token_id=[]):
            assert len(outs) == B and all(o.shape[1] == 32 + args.n_new for o in outs)
            done += 1
        assert done == n

    run_steps(img0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(img0 + args.warmup, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    single = None
    if B > 1 and rank == 0 and args.single_images > 0:
        # the same path one image at a time (the reference's loop: chair_test.py:274-346), several images after a warm-up
        one(img0 + 900, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.single_images):
            one(img0 + 901 + i, 1)
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t1) / args.single_images
        single = {"value": round(args.n_new / t1, 2), "unit": "tokens/s", "ms_per_image": round(t1 * 1e3, 1),
                  "images": args.single_images,
                  "note": "one image at a time: prefill + 128 ensemble steps; a step is ONE sweep over the weights when the masks sampled for "
                          "an empty keep set stand (speculative step, exact), two otherwise; the host reads the check's verdict and launches the re-run "
                          "only when it is needed (dd_lm_decode_step_sync)"}
    if use_dist:
        torch.distributed.barrier()
    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    streams = world if args.mode == "replicas" else 1
    tokens = streams * args.steps * args.n_new * B
    value = tokens / dt

    # dominant kernel: the gate/up decode GEMV (44 % of the streamed bytes), HIP events on the launch stream, cycling over the
    # 32 layers' weights.  With several images per GPU the member passes are 64-row passes (the members of eight sequences; 32 /
    # 16 rows with fewer images): the slice-resident pair k_gemv_slices (streams the weights) + k_gemv_finish (adds the slices,
    # epilogue).  The trailing template argument of k_gemv_slices names the matrix (2 = gate/up + SiLU).
    K_eff = 0 if args.original else len(probs)
    rows8 = min(max(K_eff, 1), 8)
    wide = B > 1 and 1 <= K_eff <= 8
    dom_rows = (64 if B >= 8 else (32 if B >= 4 else 16)) if wide else rows8
    ms_pair, by = eng.time_gemv(2, dom_rows, 96)                       # the whole GEMV (both kernels when wide)
    ms = eng.time_gemv(2 + 8, dom_rows, 96)[0] if wide else ms_pair    # the streaming kernel alone
    dom_kernel = ("k_gemv_slices_seq<8, 8, 16, 3, 0, 2>" if dom_rows == 64 else f"k_gemv_slices<1, {dom_rows // 8}, 8, 16, 16, 2, 0, 2>") if wide \
        else "k_gemv<2, 2, 8, 1, 1, 0, 0>"
    dom_name = (f"{dom_kernel} (gate/up decode GEMV of a {dom_rows}-row pass = the members of {dom_rows // 8} sequences: streams the 180 MB of "
                "weights once, K in 4 slice pairs (64 rows: one slice of a pair resident in LDS at a time); its finishing kernel k_gemv_finish adds the "
                "pairs' partial sums and applies SiLU*up)") if wide \
        else f"{dom_kernel} (gate/up decode GEMV, 8 rows)"
    achieved = by / (ms * 1e-3) / 1e9
    sweep_ms = eng.time_sweep(rows8, 5)
    sweep_bytes = eng.algorithmic_bytes(0)
    kinds = {}
    for which, name in ((0, "qkv"), (1, "o_proj"), (3, "down_proj")):
        m2, b2 = eng.time_gemv(which, dom_rows, 96)
        kinds[name] = round(b2 / (m2 * 1e-3) / 1e9, 1)
    kinds["gate_up_incl_finish"] = round(by / (ms_pair * 1e-3) / 1e9, 1)
    rows_cmp = None
    if wide:
        m8, b8 = eng.time_gemv(2, rows8, 96)
        kinds["gate_up_8_rows"] = round(b8 / (m8 * 1e-3) / 1e9, 1)
        # the same matrix at the other pass widths (streaming kernel alone): a wider pass streams the same weight bytes for more
        # rows, so its algorithmic GB/s per launch is lower while the time per row falls
        rows_cmp = {}
        for r_ in (16, 32, 64):
            mr = eng.time_gemv(2 + 8, r_, 96)[0]
            rows_cmp[str(r_)] = {"us": round(mr * 1e3, 2), "frac": round(by / (mr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "us_per_8_rows": round(mr * 1e3 * 8 / r_, 2)}

    # HBM traffic / MFMA busy of the dominant kernel: NOT measured in this run (bench.py cannot collect PMCs) — read from the
    # committed profile of this round (separate rocprofv3 --pmc passes, FETCH_SIZE x2 for gfx950), matched by kernel name;
    # the kernel-trace average duration of the same kernel from the committed stats sits beside the live HIP-event number.
    traffic = mfma_util = stats_avg_us = None
    prof = {"pmc": "profiles/r02_pmc_summary.json", "stats": "profiles/r02_kernel_stats.csv"}
    try:
        pm = json.load(open(os.path.join(ROOT, prof["pmc"])))
        for name, v in pm["kernels"].items():
            if name.startswith("void " + dom_kernel):
                traffic = v["hbm_read_bytes_per_launch"] + v.get("hbm_write_bytes_per_launch", 0)
                mfma_util = v.get("mfma_util")
    except Exception:
        pass
    try:
        import csv
        for row in csv.DictReader(open(os.path.join(ROOT, prof["stats"]))):
            if row["Name"].startswith("void " + dom_kernel):
                stats_avg_us = round(float(row["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    # end to end against SURVEY 8(d)'s per-token bytes (2 sweeps x (weights + KV at the mean context)): the one-image-at-a-time
    # rate is the like-for-like figure (a batch amortises the weight read, so the aggregate is not comparable)
    bytes_tok = 27.13e9
    e2e = None
    if single:
        e2e = {"tokens_per_s": single["value"], "bytes_per_token": bytes_tok, "achieved_GBs": round(bytes_tok * single["value"] / 1e9, 1),
               "frac": round(bytes_tok * single["value"] / 1e9 / HBM_PEAK_GBS, 4),
               "note": "SURVEY 8(d): 2*W_lm + 2*T*kv_tok at T=672 (bf16 KV) per decoded token; one image at a time, prefill included"}

    if rank == 0:
        line = {
            "metric": "decoded tokens/sec LLaVA-1.5-7B K=8 ensemble" if not args.original else "decoded tokens/sec LLaVA-1.5-7B --original",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak" if args.mode == "replicas" else "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"LLaVA-1.5-7B Dropout Decoding, {B} synthetic 336x336 image(s) per step and GPU -> each 576 visual tokens + "
                                   f"32-token prompt (prefill 608), {args.n_new} decoded tokens per image (EOS ignored), K={K_eff} voting_numbers={probs if K_eff else []}, "
                                   "random-init weights of the real shapes (bf16 weights, fp32 activations, fp16 KV cache = the reference's cache width)"
                                   + (f"; the {B} images are {B} independent sequences (own KV cache and rng stream, results identical to "
                                      "decoding each alone) whose un-masked passes share one sweep over the weights and whose member passes run eight sequences "
                                      "(64 rows) per sweep; the next batch's CLIP + prefill overlap the current batch's decode on a second stream"
                                      if B > 1 else ""),
                       "batch_note": (f"`value` is the aggregate over {B} independent images decoded concurrently per GPU (throughput mode, the "
                                      "reference's multi-process sharding on one GPU); the reference's own shape, one image at a time, is `single_stream`"
                                      if B > 1 else "one image at a time"),
                       "mode": args.mode, "images_per_step_per_gpu": B, "n_new": args.n_new, "K": K_eff,
                       "one_image_at_a_time": single,
                       "prefill_included": True, "device_bytes": eng.device_bytes},
            "single_stream": single,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": {"source": "committed profile", "measured_in_this_run": False, "file": prof["pmc"],
                                            "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch, FETCH_SIZE x2 (gfx950)"},
                         "mfma_util": mfma_util,     # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), same PMC summary
                         "rows_per_launch": dom_rows,
                         # frac counts the ALGORITHMIC bytes (the weights); the kernel also writes its K-slice partial sums (the
                         # 4 slice pairs x 64 rows x 22016 columns a finishing kernel adds up): HBM bytes actually moved / duration
                         "traffic_frac": (round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
                         "gate_up_streaming_kernel_by_rows": rows_cmp,
                         "bytes_per_launch": by, "ms_per_launch": round(ms, 5),
                         "kernel_stats_avg_us": stats_avg_us, "kernel_stats_file": prof["stats"],
                         "gemv_incl_finish": {"ms": round(ms_pair, 5), "GBs": round(by / (ms_pair * 1e-3) / 1e9, 1),
                                              "frac": round(by / (ms_pair * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "end_to_end": e2e,
                         "other_gemv_GBs": kinds,
                         "packed_sweep_8_rows": {"ms": round(sweep_ms, 4), "algorithmic_bytes": sweep_bytes,
                                          "GBs": round(sweep_bytes / (sweep_ms * 1e-3) / 1e9, 1)}},
        }
        if world == 1 and not args.no_cpu_baseline:
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
