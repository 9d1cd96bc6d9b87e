"""Experiment: 32 lanes as ONE group (one stream) vs as TWO groups of 16 on two streams driven by two host threads — do the
gaps of one group's sweep (finishing kernels, attention, launch boundaries) fill with the other group's weight stream?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
B = 32
engs = []
for i in range(B):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16",
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
for i, e in enumerate(engs):
    e.prefill(torch.randn(608, 4096, generator=torch.Generator().manual_seed(i)).cuda(), 5, 576)
torch.cuda.synchronize()
n = 20


def run(groups):
    def worker(g):
        for _ in range(n):
            g.decode_step(VOTING_NUMBERS_K8)
    for g in groups:
        for _ in range(3):
            g.decode_step(VOTING_NUMBERS_K8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(g,)) for g in groups]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = run([lm.EngineGroup(engs)])
print(f"one group of 32 lanes: {one:.2f} ms per step of all 32 lanes", flush=True)
for split in (16, 8):
    groups = []
    for g0 in range(0, B, split):
        st = torch.cuda.Stream()
        for e in engs[g0:g0 + split]:
            e.torch_stream = st
        groups.append(lm.EngineGroup(engs[g0:g0 + split]))
    t = run(groups)
    print(f"{B // split} groups of {split} lanes on {B // split} streams / host threads: {t:.2f} ms per step of all 32 lanes", flush=True)
