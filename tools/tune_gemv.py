"""GPU timing of the decode GEMV kinds (run through gpurun): 8 rows (k_gemv), 16 / 32 rows with the K-split-over-waves
kernels (k_gemv_groups) and with the slice-resident kernels (dd_gemv_slices.h + k_gemv_finish).  us per launch, TB/s of
weight bytes."""
import json, os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib, lm

torch.cuda.set_device(0)
cfg = {"llava": lm.LLAVA15_7B, "mistral": lm.MISTRAL_7B}[sys.argv[1] if len(sys.argv) > 1 else "llava"]
eng = lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576)
eng.load_synthetic(0, 0.02)
emb = torch.randn(672, 4096, device="cuda")
eng.prefill(emb, 5, 576)
L = _lib.load()
for nb, slices in [(8, 0), (16, 0), (16, 1), (32, 0), (32, 1)]:
    L.dd_tools_set_tuning(13, slices)
    r = {"rows": nb, "slices": slices}
    tot = 0.0
    for which, name in ((0, "qkv"), (1, "o"), (2, "gateup"), (3, "down")):
        best, by = 1e9, 0
        for _ in range(3):
            ms, by = eng.time_gemv(which, nb, 96)
            best = min(best, ms)
        r[name + "_us"] = round(best * 1e3, 2)
        r[name + "_TBs"] = round(by / (best * 1e-3) / 1e12, 2)
        tot += best * 1e3
    r["layer_us"] = round(tot, 1)
    print(json.dumps(r), flush=True)
L.dd_tools_set_tuning(13, 1)
