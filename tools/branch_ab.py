"""A/B: the member sweeps of a group step on one stream (tuning 23 = 1) or alternating over two (23 = 2): same tokens / logits?
ms per group step?  LLaVA-1.5-7B shapes, K = 8, fp16 KV.   python tools/branch_ab.py [lanes=32] [steps=24]"""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
engs = []
for i in range(B):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16",
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
embs = [torch.randn(608, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(B)]
out = {}
for br in (1, 2, 3, 4, 1, 2, 4):
    _lib.load().dd_tools_set_tuning(23, br)
    for e, x in zip(engs, embs):
        e.rng.manual_seed(24)
        e.prefill(x, 5, 576)
    g = lm.EngineGroup(engs)
    for _ in range(4):
        g.decode_step(VOTING_NUMBERS_K8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.decode_step(VOTING_NUMBERS_K8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res = ([e.tokens() for e in engs], [e.logits().copy() for e in (engs[0], engs[B // 2], engs[-1])], [e.kv_sums().copy() for e in (engs[0], engs[-1])])
    print(f"branches {br}: {ms:.2f} ms per group step of {B} lanes = {ms / B:.3f} ms per image-token", flush=True)
    if br in out:
        continue
    out[br] = res
a = out[1]
for br in (2, 3, 4):
    b = out[br]
    assert a[0] == b[0], f"tokens differ with {br} branches"
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        np.testing.assert_array_equal(x, y)
print("2 / 3 / 4 branches: tokens, logits and KV checksums identical to one branch")
