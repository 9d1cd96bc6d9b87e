"""fp8 slice-resident GEMVs (k_gemv_slices_fp8) vs the wave-split fp8 kernels (tuning 13 = 0): bitwise?  2 layers of the given shapes.
    python tools/fp8_slices_ab.py [dff=14336] [lanes=2]"""
import os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
torch.cuda.set_device(0)
dff = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Hkv = 8 if dff == 14336 else 32
cfg = lm.LMConfig(2048, 4096, dff, 2, 32, Hkv, 128, 1e-5, 10000.0)
engs = []
for i in range(NL):
    engs.append(lm.DropoutEngine(cfg, family=lm.FAMILY_NEXT, max_seq=160, max_visual=48, seed=50 + i, weight_format="fp8", kv_format="fp16",
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(seed=3, std=0.02)
gen = torch.Generator().manual_seed(9)
embs = [(torch.randn(60 + i % 5, 4096, generator=gen) * 0.5).cuda() for i in range(NL)]
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
def run(slices):
    _lib.load().dd_tools_set_tuning(13, slices)
    for i, (e, x) in enumerate(zip(engs, embs)):
        e.rng.manual_seed(50 + i)
        e.prefill(x, 2 + i % 3, 48)
    g = lm.EngineGroup(engs)
    out = []
    for _ in range(3):
        g.decode_step(probs)
        out.append([(e.logits().copy(), e.base_logits().copy()) for e in engs])
    return out, [e.kv_sums().copy() for e in engs]
ref, rs = run(0)
got, gs = run(1)
worst = max(float(np.abs(a[k] - b[k]).max()) for s in range(3) for a, b in zip(got[s], ref[s]) for k in (0, 1))
print(f"fp8, d_ff {dff}, {NL} lanes: slices vs wave-split kernels: max |d logits| {worst:.3g}; kv sums equal: {all(np.array_equal(a, b) for a, b in zip(gs, rs))}")
# one lane alone: speculative 16-row step vs two-sweep 8-row step
e = engs[0]
res = {}
for mode in ("never", "always"):
    e.set_speculation(mode)
    e.rng.manual_seed(50)
    e.prefill(embs[0], 2, 48)
    e.generate(12, mprobs=probs)
    res[mode] = (e.tokens(), e.logits().copy(), e.kv_sums().copy())
print(f"   solo: speculative vs two-sweep: tokens equal {res['never'][0] == res['always'][0]}, max |d logits| {np.abs(res['never'][1] - res['always'][1]).max():.3g}, "
      f"kv sums equal {np.array_equal(res['never'][2], res['always'][2])}, hit rate {e.spec_stats()['hit_rate']}")
