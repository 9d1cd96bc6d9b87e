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
for nb, dg in [(8, 0), (16, 0), (32, 0)]:
    L.dd_set_tuning(3, dg)
    r = {"rows": nb, "U": dg if dg != 8 else 4}
    for which, name in ((0, "qkv"), (1, "o"), (2, "gateup"), (3, "down")):
        best = 1e9
        for _ in range(3):
            ms, by = eng.time_gemv(which, nb, 96)
            best = min(best, ms)
        r[name + "_us"] = round(best * 1e3, 2)
    print(json.dumps(r), flush=True)
L.dd_set_tuning(3, 0)
