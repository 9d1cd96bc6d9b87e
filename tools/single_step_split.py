"""One sequence at LLaVA-1.5-7B shapes: ms per step with the speculative step on / off, how often the speculation holds, and
the step's kernels by total time (from hipEvents around N steps; run under rocprofv3 --kernel-trace --stats for the split)."""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm, _lib
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
kv = sys.argv[1] if len(sys.argv) > 1 else "fp16"
e = lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format=kv)
e.load_synthetic(0, 0.02)
x = torch.randn(608, 4096, generator=torch.Generator().manual_seed(1)).cuda()
for spec in (1, 0):
    _lib.load().dd_tools_set_tuning(14, spec)
    e.rng.manual_seed(5217)
    e.prefill(x, 5, 576)
    for _ in range(4):
        e.decode_step(VOTING_NUMBERS_K8)
    torch.cuda.synchronize()
    n, ok = 120, 0
    t0 = time.perf_counter()
    for _ in range(n):
        e.decode_step(VOTING_NUMBERS_K8)
        if spec:
            ok += int(e.spec_ok())          # syncs: per-step latency, not the pipelined rate
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"speculative={spec}: {dt:.3f} ms per step (synchronous), speculation held in {ok}/{n} steps" if spec else f"speculative={spec}: {dt:.3f} ms per step")
    e.rng.manual_seed(5217)
    e.prefill(x, 5, 576)
    t0 = time.perf_counter()
    toks = e.generate(128, mprobs=VOTING_NUMBERS_K8)
    torch.cuda.synchronize()
    print(f"   generate(128) pipelined: {(time.perf_counter() - t0) / 127 * 1e3:.3f} ms per step")
