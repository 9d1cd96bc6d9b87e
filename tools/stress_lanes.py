"""Determinism stress: R repetitions of (prefill B lanes, S group steps) from the same seeds; every repetition's tokens against the first's.
Reports the lanes and the first step at which a repetition differs.   python tools/stress_lanes.py 64 12 60 "26=1" """
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
T0 = int(os.environ.get("DD_STRESS_T0", "608"))
engs = []
for i in range(B):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16",
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(B)]
first = None
bad = 0
for rep in range(R):
    torch.cuda.synchronize()
    if os.environ.get("DD_STRESS_RECAPTURE"):
        L.dd_tools_set_tuning(26, 1)              # a tuning call starts a new graph-key epoch: the steps of this repetition are captured anew
    if os.environ.get("DD_STRESS_NOSYNC"):
        for e, x in zip(engs, embs):
            e.rng.manual_seed(24)
            e.prefill(x, 5, min(576, T0 - 16))
    if not os.environ.get("DD_STRESS_NOSYNC"):
        for e, x in zip(engs, embs):
            e.rng.manual_seed(24)
        torch.cuda.synchronize()                  # (the rng is seeded on torch's current stream, the engines run on their own)
        for e, x in zip(engs, embs):
            e.prefill(x, 5, min(576, T0 - 16))
    g = lm.EngineGroup(engs)
    for _ in range(S):
        g.decode_step(VOTING_NUMBERS_K8)
    toks = [e.tokens() for e in engs]
    sums = [e.kv_sums().copy() for e in engs]
    if first is None:
        first, first_sums = toks, sums
        continue
    kv_bad = [i for i in range(B) if toks[i] == first[i] and not (sums[i] == first_sums[i]).all()]
    if kv_bad:
        print(f"rep {rep}: same tokens but different KV checksums in lanes {kv_bad[:12]}", flush=True)
        bad_kv = True
    diff = [(i, next(s for s in range(len(t)) if s >= len(first[i]) or t[s] != first[i][s])) for i, t in enumerate(toks) if t != first[i]]
    if diff:
        bad += 1
        print(f"rep {rep}: {len(diff)} lanes differ; first differing (lane, token index): {sorted(diff, key=lambda d: d[1])[:6]}", flush=True)
print(f"{bad} of {R - 1} repetitions differ from the first ({B} lanes, {S} steps, settings {sys.argv[4:]})")
