"""Streaming-read ceiling of this board as a function of kernel size (bytes per launch) and block count."""
import ctypes as C, os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import _lib
torch.cuda.set_device(0)
L = _lib.load()
buf = torch.empty(6 << 30, dtype=torch.uint8, device="cuda"); buf.fill_(1)
st = torch.cuda.current_stream().cuda_stream
def one(off, nbytes, nb):
    g = C.c_float()
    _lib.check(L.dd_hbm_read_bench(buf.data_ptr() + off, nbytes, 1, nb, C.byref(g), st))
    return g.value
for mb in (33, 90, 180, 262, 1024, 4096):
    nbytes = mb << 20
    for nb in (1024, 2048, 4096, 8192):
        vals = []
        off = 0
        for it in range(12):
            off = (off + nbytes + (512 << 20)) % ((6 << 30) - nbytes - 1)
            off -= off % 4096
            vals.append(one(off, nbytes, nb))
        vals.sort()
        print(f"{mb:5d} MB per launch, {nb:5d} blocks: median {vals[len(vals)//2]:7.1f} GB/s  best {vals[-1]:7.1f}", flush=True)
