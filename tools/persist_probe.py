"""Plain streaming reads with the decode sweep's launch structure (one launch per weight matrix, LLaVA-1.5-7B sizes):
what does the memory system deliver when the kernel does nothing but read?  The gap to the GEMV timings
(tools/tune_gemv.py) is what the GEMV kernel itself costs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dropoutdecoding_amd import _lib, build

build.build()
L = _lib.load()
d, dff = 4096, 11008
raw = {"qkv": 3 * d * d * 2, "o": d * d * 2, "gate_up": 2 * dff * d * 2, "down": d * dff * 2}
n_layers = 32
st = torch.cuda.Stream()
buf = torch.empty(sum(raw.values()) * n_layers, dtype=torch.uint8, device="cuda")
buf.zero_()
torch.cuda.synchronize()
for name, b in raw.items():
    for grid in (512, 1024):
        unit = grid * 64 * 16 * 8
        ph = (b // unit) * unit
        arr = (C.c_size_t * 1)(ph)
        for U in (4, 8, 16):
            for wpc in (0, 3, 2, 1):
                if U == 4:
                    continue
                ms = C.c_float()
                rc = L.dd_persist_read_bench(buf.data_ptr(), sum(raw.values()), arr, 1, n_layers, 0 | (wpc << 8), grid, U, 5, C.byref(ms), st.cuda_stream)
                assert rc == 0, L.dd_last_error()
                us = ms.value * 1e3 / n_layers
                print(f"{name:8s} {ph / 1e6:7.1f} MB grid {grid:5d} U {U:2d} wg/cu<={wpc}: {us:6.2f} us/launch {ph / us / 1e3:7.1f} GB/s", flush=True)
