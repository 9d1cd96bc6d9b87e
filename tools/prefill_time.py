"""Prefill time of one prompt (LLaVA-1.5-7B shapes) under the prefill GEMM's tuning keys: 15 = XCD-aware block order,
16 = rows from which the 128 x 512 LDS-staged block is used.  Also checks that every variant gives the same bits
(image logits, prefill logits row, KV checksums).     python tools/prefill_time.py [T0] [order,big_rows]"""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import lm, _lib

torch.cuda.set_device(0)
T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 608
L = min(576, T0 - 8) if T0 < 2000 else T0 - 32
e = lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=T0 + 192, max_visual=L, kv_format="fp16")
e.load_synthetic(0, 0.02)
x = torch.randn(T0, 4096, generator=torch.Generator().manual_seed(0)).cuda()
lib = _lib.load()
ref = None
VARIANTS = ((0, 0), (1, 0), (1, 1024), (0, 1024), (1, 512), (1, 0))
if len(sys.argv) > 2:                            # "order,big_rows": one variant only (for rocprofv3 --pmc passes)
    VARIANTS = (tuple(int(v) for v in sys.argv[2].split(",")),)
for order, big in VARIANTS:
    lib.dd_tools_set_tuning(15, order)
    lib.dd_tools_set_tuning(16, big)
    for _ in range(2):
        e.prefill(x, 5, L)
    torch.cuda.synchronize()
    t0 = time.perf_counter()                     # the engine runs on its own stream: wall clock around a full sync
    for _ in range(5):
        e.prefill(x, 5, L)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    sig = (e.image_logits().tobytes(), e.logits().tobytes(), e.kv_sums().tobytes())
    if ref is None:
        ref = sig
    same = all(a == b for a, b in zip(sig, ref))
    print(f"T0 {T0} xcd_order {order} big_rows {big}: prefill {ms:.2f} ms   same bits as the first variant: {same}", flush=True)

# several sequences at once (dd_lm_prefill_group) against one prefill per sequence
if T0 <= 1024:
    lib.dd_tools_set_tuning(15, 1)
    lib.dd_tools_set_tuning(16, 1024)
    for n in (8, 16):
        lanes = [e] + [lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=T0 + 192, max_visual=L, kv_format="fp16",
                                        share_weights_with=e) for _ in range(n - 1)]
        xs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(n)]
        for mode in ("one by one", "group"):
            def go():
                if mode == "group":
                    lm.prefill_group(lanes, xs, [(5, L)] * n)
                else:
                    for q, x_ in zip(lanes, xs):
                        q.prefill(x_, 5, L)
            go()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                go()
            torch.cuda.synchronize()
            print(f"{n} sequences of {T0} rows, {mode}: {(time.perf_counter() - t0) / 3 / n * 1e3:.2f} ms per sequence", flush=True)
        for q in lanes[1:]:
            q.close()
