"""Determinism stress: R repetitions of (prefill B lanes, S group steps) from the same seeds.  Every repetition is compared
  * with the first repetition: tokens, KV checksums and (DD_STRESS_TRACE=1) the per-step trace records — for a lane that differs the FIRST
    step and the FIRST quantity that differs are named (un-masked logits / argmax -> keep set -> rng index -> masks -> member argmax -> winner),
  * with one lane decoded ALONE (DD_STRESS_SOLO=1, default on): lane rep % B through the single-sequence step on an engine of its own.

    python tools/stress_lanes.py B R S ["key=value,..." ...]       # dd_tools_set_tuning keys, e.g. "33=0" branch-local sampling, "34=1" round-3 sampler

Environment: DD_STRESS_K (8 | 4 | 3 ...: members), DD_STRESS_FAMILY (llava | iblip), DD_STRESS_T0 (prompt length), DD_STRESS_POISON=n (n LDS-poison
launches of 512 workgroups beside every step), DD_STRESS_STOP=n (stop after n differing repetitions), DD_STRESS_KV (fp16 | fp32),
DD_STRESS_LOG=path (append one JSON line with the summary).  Exit code 1 when a repetition or a solo check differs."""
import json, os, sys, time
# the shipped libdropdec.so unless a tools-only feature is asked for (tuning keys, the step trace, LDS poison): the summary records which
NEED_TOOLS = len(sys.argv) > 4 or any(os.environ.get(k, "0") not in ("", "0") for k in ("DD_STRESS_TRACE", "DD_STRESS_POISON"))
if NEED_TOOLS:
    os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
S = int(sys.argv[3]) if len(sys.argv) > 3 else 60
L = _lib.load()
for stg in sys.argv[4:]:
    for p in stg.split(","):
        k, v = p.split("=")
        L.dd_tools_set_tuning(int(k), int(v))
K = int(os.environ.get("DD_STRESS_K", "8"))
PROBS = list(VOTING_NUMBERS_K8) if K == 8 else [0.1, 0.3, 0.5, 0.7, 0.2, 0.4, 0.6, 0.8][:K]   # K = 4: chair_test.py:170
FAMILY = os.environ.get("DD_STRESS_FAMILY", "llava")
KV = os.environ.get("DD_STRESS_KV", "fp16")
TRACE = os.environ.get("DD_STRESS_TRACE", "0") not in ("", "0")
SOLO = os.environ.get("DD_STRESS_SOLO", "1") not in ("", "0")
POISON = int(os.environ.get("DD_STRESS_POISON", "0"))
STOP = int(os.environ.get("DD_STRESS_STOP", "0"))
if FAMILY == "iblip":
    fam, CFG, Lvis, T0 = lm.FAMILY_IBLIP, lm.VICUNA_7B, 32, int(os.environ.get("DD_STRESS_T0", "64"))
else:
    fam, CFG, Lvis, T0 = lm.FAMILY_LLAVA, lm.LLAVA15_7B, 576, int(os.environ.get("DD_STRESS_T0", "608"))
span0 = 0 if FAMILY == "iblip" else 5
span_len = min(Lvis, T0 - 16)
engs = []
for i in range(B + (1 if SOLO else 0)):
    engs.append(lm.DropoutEngine(CFG, family=fam, max_seq=T0 + S + 80, max_visual=Lvis, kv_format=KV,
                                 share_weights_with=engs[0] if engs else None))
solo = engs.pop() if SOLO else None
engs[0].load_synthetic(0, 0.02)
embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(B)]
traces = None
if TRACE:
    traces = [torch.zeros(S + 2, 32, dtype=torch.int32, device="cuda") for _ in range(B)]
    for e, t in zip(engs, traces):
        L.dd_tools_trace_attach(e._h, t.data_ptr(), S + 2, e.rng.handle)
side = torch.cuda.Stream() if POISON else None
FIELDS = [(23, "un-masked logits"), (1, "un-masked argmax"), (2, "keep-set size"), (22, "rng read index")] + [(4 + k, f"n_drop[{k}]") for k in range(8)] + \
         [(3, "drop bits")] + [(12 + k, f"member argmax[{k}]") for k in range(8)] + [(20, "winner"), (21, "token"), (0, "token count")]


def first_divergence(a, b):
    """(step, name) of the first differing trace field of one lane, fields in the order the step computes them."""
    for s in range(min(len(a), len(b))):
        if (a[s] != b[s]).any():
            for idx, name in FIELDS:
                if a[s][idx] != b[s][idx]:
                    return s, name
    return None


first = None
bad = bad_solo = 0
events = []
t_start = time.time()
reps_done = 0
for rep in range(R):
    torch.cuda.synchronize()
    for e in engs:
        e.rng.manual_seed(24)
    if traces is not None:
        for t in traces:
            t.zero_()
    torch.cuda.synchronize()                  # (the rng is seeded on torch's current stream, the engines run on their own)
    for e, x in zip(engs, embs):
        e.prefill(x, span0, span_len)
    g = lm.EngineGroup(engs)
    for _ in range(S):
        if POISON:
            L.dd_tools_lds_poison(POISON, 512, 64 * 1024, side.cuda_stream)
        g.decode_step(PROBS)
    toks = [e.tokens() for e in engs]
    sums = [e.kv_sums().copy() for e in engs]
    tr = [t.cpu().numpy().copy() for t in traces] if traces is not None else None
    reps_done += 1
    if SOLO:
        i = rep % B
        solo.rng.manual_seed(24)
        torch.cuda.synchronize()
        solo.prefill(embs[i], span0, span_len)
        for _ in range(S):
            solo.decode_step(PROBS)
        st = solo.tokens()
        if st != toks[i]:
            bad_solo += 1
            s0 = next(s for s in range(min(len(st), len(toks[i]))) if st[s] != toks[i][s]) if st[:len(toks[i])] != toks[i][:len(st)] else min(len(st), len(toks[i]))
            msg = f"rep {rep}: lane {i} differs from its SOLO run at token {s0}"
            print(msg, flush=True)
            events.append(msg)
    if first is None:
        first, first_sums, first_tr = toks, sums, tr
        continue
    kv_bad = [i for i in range(B) if toks[i] == first[i] and not (sums[i] == first_sums[i]).all()]
    if kv_bad:
        msg = f"rep {rep}: same tokens but different KV checksums in lanes {kv_bad[:12]}"
        print(msg, flush=True)
        events.append(msg)
    diff = [(i, next(s for s in range(len(t)) if s >= len(first[i]) or t[s] != first[i][s])) for i, t in enumerate(toks) if t != first[i]]
    tr_diff = []
    if tr is not None:
        for i in range(B):
            d = first_divergence(first_tr[i], tr[i])
            if d is not None:
                tr_diff.append((i, d[0], d[1]))
    if diff or tr_diff or kv_bad:
        bad += 1
        msg = f"rep {rep}: {len(diff)} lanes differ in tokens; first (lane, token index): {sorted(diff, key=lambda d: d[1])[:6]}"
        if tr is not None:
            msg += f"; trace: first (lane, step, quantity): {sorted(tr_diff, key=lambda d: d[1])[:6]}"
        print(msg, flush=True)
        events.append(msg)
        if STOP and bad >= STOP:
            break
    if rep % 20 == 19:
        print(f"  ... rep {rep + 1}: {bad} differing so far, {time.time() - t_start:.0f} s", flush=True)
summary = {"tool": "stress_lanes", "library": "libdropdec_tools.so" if os.environ.get("DD_USE_TOOLS_LIB", "0") not in ("", "0") else "libdropdec.so",
           "lanes": B, "repetitions": reps_done, "steps_per_repetition": S, "steps_total": reps_done * S, "K": K, "family": FAMILY, "kv": KV,
           "settings": sys.argv[4:], "trace": TRACE, "solo_checks": reps_done if SOLO else 0, "poison_launches_per_step": POISON,
           "differ_from_first": bad, "differ_from_solo": bad_solo, "events": events[:20], "seconds": round(time.time() - t_start, 1)}
print(f"{bad} of {reps_done - 1} repetitions differ from the first, {bad_solo} of {reps_done if SOLO else 0} solo checks differ "
      f"({B} lanes, {S} steps, K={K}, {FAMILY}, settings {sys.argv[4:]})")
print(json.dumps(summary))
if os.environ.get("DD_STRESS_LOG"):
    with open(os.environ["DD_STRESS_LOG"], "a") as f:
        f.write(json.dumps(summary) + "\n")
sys.exit(1 if bad or bad_solo else 0)      # a guard, not a log: a differing repetition fails the wrapper script
