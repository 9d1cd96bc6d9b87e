"""Decode GEMVs of the engine in isolation (dd_lm_time_gemv: HIP events, weights cycled over the 32 layers):
per matrix and row count, the whole GEMV (streaming kernel + finishing kernel) and the streaming kernel alone."""
import os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm

torch.cuda.set_device(0)
from dropoutdecoding_amd import _lib
row_sets = (8, 16, 32, 64, 72)
for kv in sys.argv[1:]:                      # key=value: dd_tools_set_tuning; rows=64: only that pass width
    k_, v_ = kv.split("=")
    if k_ == "rows":
        row_sets = (int(v_),)
    else:
        _lib.load().dd_tools_set_tuning(int(k_), int(v_))
C5 = os.environ.get("DD_AB_MODEL", "") == "mistral-fp8"       # BASELINE config 5's matrices: Mistral-7B shapes, fp8 tiles
e = lm.DropoutEngine(lm.MISTRAL_7B if C5 else lm.LLAVA15_7B, family=lm.FAMILY_NEXT if C5 else lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16",
                     weight_format="fp8" if C5 else "bf16")
e.load_synthetic(0, 0.02)
e.prefill(torch.randn(608, 4096, generator=torch.Generator().manual_seed(0)).cuda(), 5, 576)
names = {0: "qkv", 1: "o_proj", 2: "gate/up", 3: "down"}
for rows in row_sets:
    out = []
    for which in range(4):
        ms, by = e.time_gemv(which, rows, 96)
        ms_s = e.time_gemv(which + 8, rows, 96)[0] if rows >= 16 else ms
        out.append(f"{names[which]} {ms * 1e3:6.1f} us ({by / ms / 1e9 * 1e3 / 1e3:5.2f} TB/s; streaming kernel {ms_s * 1e3:5.1f} us = {by / ms_s / 1e9:5.2f} TB/s)")
    print(f"{rows:2d} rows: " + " | ".join(out), flush=True)
