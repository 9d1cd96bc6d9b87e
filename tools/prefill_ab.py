"""Prefill time under the round-4 prefill kernel switches (tools keys 45 = 16-row RMSNorm split, 46 = operand-staged fp16-cache attention with
1 / 2 query blocks per wave, 0 = the round-2 kernel), 8-layer engines of LLaVA-1.5-7B (MHA, 608 rows, 16 prompts as one matrix) and
Mistral-7B (GQA 4, 2960 rows) shapes; also checks that every variant leaves the same bits.     python tools/prefill_ab.py [llava|mistral]"""
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
for name, cfg, T0, n in (("llava", LLAVA, 608, 16), ("mistral", MISTRAL, 2960, 1)):
    if ONLY and name != ONLY:
        continue
    L = T0 - 32
    lanes = []
    for i in range(n):
        lanes.append(lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=T0 + 64, max_visual=L, kv_format="fp16",
                                      share_weights_with=lanes[0] if lanes else None))
    lanes[0].load_synthetic(0, 0.02)
    xs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(n)]
    ref = None
    for norm16, attn16 in ((0, 0), (1, 0), (1, 1), (1, 2), (0, 0), (1, 1), (1, 2)):
        lib.dd_tools_set_tuning(45, norm16)
        lib.dd_tools_set_tuning(46, attn16)

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
        print(f"{name}: {n} x {T0} rows, 8 layers, norm16 {norm16} attn16 {attn16}: {ms:.2f} ms ({ms / 8 / n * 1e3:.0f} us per layer and prompt)   "
              f"same bits as the first variant: {sig == ref}", flush=True)
    for e in reversed(lanes):
        e.close()
lib.dd_tools_set_tuning(45, 1)
lib.dd_tools_set_tuning(46, 1)
