"""GPU tuning sweep of the decode GEMV variants (run through gpurun). Prints GB/s per GEMV kind and the packed sweep time."""
import itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib, lm

torch.cuda.set_device(0)
eng = lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576)
eng.load_synthetic(0, 0.02)
emb = torch.randn(672, 4096, device="cuda")
eng.prefill(emb, 5, 576)
L = _lib.load()
rows = []
for u, nt, il, dg in [(8, 0, 1, 0), (8, 0, 1, 64), (8, 0, 1, 32), (8, 0, 1, 0), (8, 0, 1, 64), (8, 0, 1, 32)]:
    L.dd_set_tuning(0, u); L.dd_set_tuning(4, nt); L.dd_set_tuning(3, dg)
    r = {"U": u, "RING": nt, "DIAG": dg}
    for which, name in ((0, "qkv"), (1, "o"), (2, "gateup"), (3, "down")):
        best = 0
        for _ in range(3):
            ms, by = eng.time_gemv(which, 8, 96)
            best = max(best, by / ms / 1e6)
        r[name] = round(best)
    r["sweep8_ms"] = round(min(eng.time_sweep(8, 5) for _ in range(3)), 3)
    rows.append(r)
    print(json.dumps(r), flush=True)
