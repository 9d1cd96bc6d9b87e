"""A/B of the group step forms: ms per step for tuning settings given as key=value lists (26 rider form, 30 half planes for K <= 4, ...).
   python tools/rider_ab.py 32 "26=0" "26=1" "26=1,22=0" """
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

PROBS = {8: VOTING_NUMBERS_K8, 4: [0.1, 0.3, 0.5, 0.7], 3: [0.3, 0.5, 0.7]}[int(os.environ.get("DD_AB_K", "8"))]   # DD_AB_K=4: the reference's K = 4

torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
settings = sys.argv[2:] or ["26=0", "26=1"]
steps = 24
engs = []
C5 = os.environ.get("DD_AB_MODEL", "") == "mistral-fp8"       # BASELINE config 5's shapes: GQA 4, 2,928 visual tokens, fp8 matrices
T0, LV = (2960, 2928) if C5 else (608, 576)
for i in range(B):
    engs.append(lm.DropoutEngine(lm.MISTRAL_7B if C5 else lm.LLAVA15_7B, family=lm.FAMILY_NEXT if C5 else lm.FAMILY_LLAVA, max_seq=T0 + 176, max_visual=LV,
                                 kv_format="fp16", weight_format="fp8" if C5 else "bf16", share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(B)]
L = _lib.load()
first = None
for rep in range(2):
    for stg in settings:
        kv = [tuple(int(x) for x in p.split("=")) for p in stg.split(",")]
        for k, v in kv:
            L.dd_tools_set_tuning(k, v)
        for e, x in zip(engs, embs):
            e.rng.manual_seed(24)
            e.prefill(x, 5, LV)
        g = lm.EngineGroup(engs)
        for _ in range(4):
            g.decode_step(PROBS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.decode_step(PROBS)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        toks = [e.tokens() for e in engs]
        if first is None:
            first = toks
        print(f"{stg}: {ms:.2f} ms per group step of {B} lanes; tokens {'same' if toks == first else 'DIFFER'}", flush=True)
        for k, v in kv:          # back to the defaults
            L.dd_tools_set_tuning(k, {26: 1, 27: 1, 28: 4, 29: 4, 30: 1, 31: 1, 22: 1, 23: 2, 21: 0}.get(k, 0))
