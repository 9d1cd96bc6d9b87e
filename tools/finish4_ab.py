"""Which epilogue of k_gemv_finish4 differs from k_gemv_finish?  4 lanes (32-row passes), 2 layers of 7B shapes."""
import os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
torch.cuda.set_device(0)
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = lm.LMConfig(2048, 4096, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
engs = []
for i in range(NL):
    engs.append(lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=120, max_visual=24, seed=50 + i, share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(seed=3, std=0.02)
gen = torch.Generator().manual_seed(9)
embs = [(torch.randn(30 + i % 5, 4096, generator=gen) * 0.5).cuda() for i in range(NL)]
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
def run(mask):
    _lib.load().dd_tools_set_tuning(24, mask)
    for i, (e, x) in enumerate(zip(engs, embs)):
        e.rng.manual_seed(50 + i)
        e.prefill(x, 2 + i % 3, 24)
    g = lm.EngineGroup(engs)
    out = []
    for _ in range(2):
        g.decode_step(probs)
        out.append([(e.logits().copy(), e.base_logits().copy()) for e in engs])
    return out
ref = run(0)
for name, mask in (("STORE", 1), ("RESID", 2), ("SILU", 4), ("QKV", 8), ("all", 15)):
    got = run(mask)
    worst = max(float(np.abs(a[k] - b[k]).max()) for s in range(2) for a, b in zip(got[s], ref[s]) for k in (0, 1))
    print(f"finish4 for {name}: max |difference| vs k_gemv_finish {worst:.3g}", flush=True)
