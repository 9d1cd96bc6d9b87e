"""Host enqueue time vs GPU time of a group step (8 lanes, LLaVA-1.5-7B shapes, K=8)."""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
KV = sys.argv[2] if len(sys.argv) > 2 else "fp32"
from dropoutdecoding_amd import _lib
if len(sys.argv) > 3:
    _lib.load().dd_tools_set_tuning(9, int(sys.argv[3]))        # sequences per member sweep (1, 2, 4, 8)
for kv in sys.argv[4:]:                                   # further tuning keys: key=value
    k_, v_ = kv.split("=")
    _lib.load().dd_tools_set_tuning(int(k_), int(v_))
engs = []
for i in range(B):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format=KV,
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
for i, e in enumerate(engs):
    e.prefill(torch.randn(608, 4096, generator=torch.Generator().manual_seed(i)).cuda(), 5, 576)
g = lm.EngineGroup(engs)
for _ in range(3):
    g.decode_step(VOTING_NUMBERS_K8)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
per = []
for _ in range(n):
    ta = time.perf_counter()
    g.decode_step(VOTING_NUMBERS_K8)
    per.append(round((time.perf_counter() - ta) * 1e3, 2))
t1 = time.perf_counter()
print("host ms per call:", per)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"lanes {B} kv {KV} seqs/sweep {sys.argv[3] if len(sys.argv) > 3 else 'default'}: host enqueue {(t1 - t0) / n * 1e3:.2f} ms/group-step, total {(t2 - t0) / n * 1e3:.2f} ms/group-step "
      f"= {(t2 - t0) / n / B * 1e3:.2f} ms per image-token")
