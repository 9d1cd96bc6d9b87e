"""How fast does the bench host stream a batch-1 linear?  (sizing the CPU baseline: bench.py cpu_baseline)"""
import time, torch, os
import torch.nn.functional as F
print(torch.__config__.parallel_info())
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    for dt in (torch.float32, torch.bfloat16):
        Ws = [torch.randn(11008, 4096).to(dt) for _ in range(6)]
        x = torch.randn(1, 4096).to(dt)
        for name, f in (("F.linear", lambda x, W: F.linear(x, W)), ("mv", lambda x, W: torch.mv(W, x[0])),
                        ("mm_t", lambda x, W: torch.mm(W, x.t()))):
            f(x, Ws[0])
            t0 = time.perf_counter()
            for i in range(12):
                f(x, Ws[i % 6])
            t = (time.perf_counter() - t0) / 12
            print(nt, dt, name, f"{t*1e3:.2f} ms  {Ws[0].numel()*Ws[0].element_size()/t/1e9:.1f} GB/s", flush=True)
