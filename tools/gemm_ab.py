"""Prefill GEMM forms A/B (dd_set_tuning key 20: 0 = the register-staged 128 x 512 block of round 3, 1 = the LDS-DMA 160 x 512 block of
round 6): prefill time of 8-layer engines at LLaVA-1.5-7B shapes (16 prompts of 608 rows as ONE matrix of 16 x 640 rows — bench.py's prefill
pass) and Mistral-7B shapes (one prompt of 2,960 rows, config 5), alternating A B A B, with a bit-equality check of everything the prefill
leaves.     python tools/gemm_ab.py [llava|mistral]"""
import dataclasses, os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm, _lib

torch.cuda.set_device(0)
lib = _lib.load()
MISTRAL = lm.LMConfig(32064, 4096, 14336, 8, 32, 8, 128, 1e-5, 1000000.0)
LLAVA = dataclasses.replace(lm.LLAVA15_7B, num_layers=8)
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
FORMS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 0, 1, 0, 1]     # 8 / 10: the LDS-DMA block with 128 / 160 rows forced
for name, cfg, T0, n, wf in (("llava", LLAVA, 608, 16, "bf16"), ("mistral", MISTRAL, 2960, 1, "bf16"), ("mistral-fp8", MISTRAL, 2960, 1, "fp8"),
                            ("mistral4-fp8", MISTRAL, 2960, 4, "fp8")):
    if ONLY and not name.startswith(ONLY):
        continue
    L = T0 - 32
    lanes = []
    for i in range(n):
        lanes.append(lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=T0 + 64, max_visual=L, kv_format="fp16", weight_format=wf,
                                      share_weights_with=lanes[0] if lanes else None))
    lanes[0].load_synthetic(0, 0.02)
    xs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(n)]
    ref = None
    for form in FORMS:
        lib.dd_tools_set_tuning(20, form)

        def go():
            if n > 1:
                lm.prefill_group(lanes, xs, [(5, L)] * n)
            else:
                lanes[0].prefill(xs[0], 5, L)
        go()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        sig = tuple((e.image_logits().tobytes(), e.logits().tobytes(), e.kv_sums().tobytes()) for e in lanes)
        if ref is None:
            ref = sig
        print(f"{name}: {n} x {T0} rows, 8 layers, GEMM block form {form} ({ {0: 'register-staged 128 x 512', 1: 'LDS-DMA, rows by the launch', 8: 'LDS-DMA 128 x 512', 10: 'LDS-DMA 160 x 512'}[form]}): {ms:.2f} ms "
              f"({ms / 8 * 1e3:.0f} us per layer)   same bits as the first run: {sig == ref}", flush=True)
    for e in reversed(lanes):
        e.close()
lib.dd_tools_set_tuning(20, 1)
