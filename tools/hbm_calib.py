import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib
torch.cuda.set_device(0)
L = _lib.load()
buf = torch.empty(4 << 30, dtype=torch.uint8, device="cuda"); buf.fill_(1)
for nb in (1024, 2048, 4096, 8192, 16384):
    g = C.c_float()
    _lib.check(L.dd_hbm_read_bench(buf.data_ptr(), buf.numel(), 5, nb, C.byref(g), torch.cuda.current_stream().cuda_stream))
    print("blocks", nb, "read GB/s", round(g.value, 1), flush=True)
# torch copy for reference (read+write)
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
b.copy_(a); s.record()
for _ in range(10): b.copy_(a)
e.record(); torch.cuda.synchronize()
print("torch copy (read+write) GB/s", round(2 * a.numel() * 10 / (s.elapsed_time(e) * 1e-3) / 1e9, 1))
