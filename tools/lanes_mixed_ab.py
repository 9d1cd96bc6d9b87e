"""Debug / A-B: 11 lanes (one 8-sequence member sweep, one 2-sequence sweep, one single) as a group with 1 or 2 branches vs each
lane alone: which lanes differ?   python tools/lanes_mixed_ab.py [kv=fp32]"""
import os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm

torch.cuda.set_device(0)
KV = sys.argv[1] if len(sys.argv) > 1 else "fp32"
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 11
for kv in sys.argv[3:]:
    k_, v_ = kv.split("=")
    _lib.load().dd_tools_set_tuning(int(k_), int(v_))
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
shapes = [(608, 5, 576), (640, 9, 576), (600, 1, 576), (615, 20, 576), (609, 5, 576), (700, 60, 576),
          (610, 3, 576), (633, 7, 576), (655, 11, 576), (602, 2, 576), (690, 33, 576), (611, 4, 576), (644, 8, 576), (603, 2, 576),
          (620, 6, 576), (699, 30, 576)][:NL]
engs = []
for i in range(len(shapes)):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=768, max_visual=576, seed=5217, kv_format=KV,
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(1, 0.02)
embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(50 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
n_steps = int(os.environ.get("DD_AB_STEPS", "3"))
POISON = int(os.environ.get("DD_AB_POISON", "0"))     # LDS-poison launches beside every step (dd_tools_lds_poison)
side = torch.cuda.Stream() if POISON else None
res = {}
for br in (1, 2):
    _lib.load().dd_tools_set_tuning(23, br)
    for e, x, (T0, s0, L) in zip(engs, embs, shapes):
        e.rng.manual_seed(5217)
        e.prefill(x, s0, L)
    grp = lm.EngineGroup(engs)
    rec = [[] for _ in engs]
    for s in range(n_steps):
        if POISON:
            _lib.load().dd_tools_lds_poison(POISON, 512, 64 * 1024, side.cuda_stream)
        grp.decode_step(probs)
        for i, e in enumerate(engs):
            rec[i].append((e.last_step()["drop"].copy(), e.logits().copy(), e.base_logits().copy()))
    res[br] = rec
solo = [[] for _ in engs]
for i, (e, x, (T0, s0, L)) in enumerate(zip(engs, embs, shapes)):
    e.rng.manual_seed(5217)
    e.prefill(x, s0, L)
    for s in range(n_steps):
        e.decode_step(probs)
        solo[i].append((e.last_step()["drop"].copy(), e.logits().copy(), e.base_logits().copy()))
for br in (1, 2):
    for i in range(len(engs)):
        for s in range(n_steps):
            a, b = res[br][i][s], solo[i][s]
            bad = [name for name, x, y in (("drop", a[0], b[0]), ("logits", a[1], b[1]), ("base_logits", a[2], b[2])) if not np.array_equal(x, y)]
            if bad:
                print(f"branches {br}: lane {i} step {s} differs from its solo run in {bad}; max |d logits| {np.abs(a[1] - b[1]).max():.3g}")
print(f"done: kv {KV}, {NL} lanes")
