"""LLaVA-NeXT-Mistral-7B shapes (GQA 32/8, 5x576+48 = 2928 visual tokens), synthetic weights: engine-level timing."""
import json, os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dropoutdecoding_amd import lm
torch.cuda.set_device(0)
K = 8
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
for name, cfg, fam, L, T0 in (("llava-next-mistral-7b", lm.MISTRAL_7B, lm.FAMILY_NEXT, 2928, 2960),
                             ("instructblip-vicuna-7b", lm.VICUNA_7B, lm.FAMILY_IBLIP, 32, 64)):
    KV = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    eng = lm.DropoutEngine(cfg, family=fam, max_seq=T0 + 140, max_visual=L, kv_format=KV)
    eng.load_synthetic(0, 0.02)
    emb = torch.randn(T0, 4096, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.prefill(emb, 0 if fam == lm.FAMILY_IBLIP else 5, L); torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.prefill(emb, 0 if fam == lm.FAMILY_IBLIP else 5, L); torch.cuda.synchronize(); t2 = time.perf_counter()
    toks = eng.generate(128, mprobs=probs); torch.cuda.synchronize(); t3 = time.perf_counter()
    st = eng.last_step()
    print(json.dumps({"model": name, "prefill_ms": round((t2 - t1) * 1e3, 1), "decode_ms_per_step": round((t3 - t2) / 127 * 1e3, 3),
                      "tokens_per_s_incl_prefill": round(128 / (t3 - t1), 1), "masked_numbers": st["masked_numbers"].tolist(),
                      "sweep8_ms": round(eng.time_sweep(8, 3), 3), "device_GB": round(eng.device_bytes / 1e9, 2)}), flush=True)
    # the same shapes as 8 lanes over these weights (fused base pass + grouped member sweeps)
    B = 8
    lanes = [eng] + [lm.DropoutEngine(cfg, family=fam, max_seq=T0 + 140, max_visual=L, share_weights_with=eng, kv_format=KV) for _ in range(B - 1)]
    for i, e in enumerate(lanes):
        e.prefill(torch.randn(T0, 4096, device="cuda"), 0 if fam == lm.FAMILY_IBLIP else 5, L)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    lm.EngineGroup(lanes).generate(128, mprobs=probs); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(json.dumps({"model": name, "lanes": B, "group_step_ms": round((t5 - t4) / 127 * 1e3, 3),
                      "ms_per_image_token": round((t5 - t4) / 127 / B * 1e3, 3),
                      "decode_tokens_per_s": round(B * 127 / (t5 - t4), 1)}), flush=True)
    for e in reversed(lanes):
        e.close()
    del eng, lanes; torch.cuda.empty_cache()
