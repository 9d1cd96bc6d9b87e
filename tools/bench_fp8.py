"""fp8-weight engine timing (LLaVA-NeXT-Mistral-7B shapes = BASELINE config 5, and LLaVA-1.5-7B shapes)."""
import json, os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm
torch.cuda.set_device(0)
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
for name, cfg, fam, L, T0 in (("llava-1.5-7b", lm.LLAVA15_7B, lm.FAMILY_LLAVA, 576, 608),
                             ("llava-next-mistral-7b", lm.MISTRAL_7B, lm.FAMILY_NEXT, 2928, 2960)):
    for fmt in ("bf16", "fp8"):
        eng = lm.DropoutEngine(cfg, family=fam, max_seq=T0 + 140, max_visual=L, weight_format=fmt)
        eng.load_synthetic(0, 0.02)
        emb = torch.randn(T0, 4096, device="cuda")
        eng.prefill(emb, 5, L); torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.prefill(emb, 5, L); torch.cuda.synchronize(); t2 = time.perf_counter()
        eng.generate(64, mprobs=probs); torch.cuda.synchronize(); t3 = time.perf_counter()
        r = {"model": name, "weights": fmt, "prefill_ms": round((t2 - t1) * 1e3, 1), "decode_ms_per_step": round((t3 - t2) / 63 * 1e3, 3),
             "sweep8_ms": round(eng.time_sweep(8, 3), 3), "device_GB": round(eng.device_bytes / 1e9, 2)}
        for which, nm in ((0, "qkv"), (2, "gateup"), (3, "down")):
            ms, by = eng.time_gemv(which, 8, 64)
            r[nm + "_GBs"] = round(by / ms / 1e6)
        print(json.dumps(r), flush=True)
        B = 8                                  # the same as 8 lanes over these weights
        lanes = [eng] + [lm.DropoutEngine(cfg, family=fam, max_seq=T0 + 140, max_visual=L, weight_format=fmt, share_weights_with=eng)
                         for _ in range(B - 1)]
        for e in lanes:
            e.prefill(torch.randn(T0, 4096, device="cuda"), 5, L)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        lm.EngineGroup(lanes).generate(64, mprobs=probs); torch.cuda.synchronize(); t5 = time.perf_counter()
        print(json.dumps({"model": name, "weights": fmt, "lanes": B, "group_step_ms": round((t5 - t4) / 63 * 1e3, 3),
                          "ms_per_image_token": round((t5 - t4) / 63 / B * 1e3, 3)}), flush=True)
        for e in reversed(lanes):
            e.close()
        del eng, lanes; torch.cuda.empty_cache()
