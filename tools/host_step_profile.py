"""Where does a reference-faithful CPU decode forward spend its time on the bench host? (one layer-forward, T=608)"""
import sys, os, time, torch
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from oracle.lm_ref import LMConfig, KVCache, lm_hidden, lm_logits
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.set_num_threads(nt)
for dt in (torch.bfloat16, torch.float32):
    d, dff, nl, T = 4096, 11008, 2, 608
    cfg = LMConfig(32064, d, dff, nl, 32, 32, 128, 1e-5, 10000.0)
    w = {"model.embed_tokens.weight": torch.randn(8, d).to(dt), "model.norm.weight": torch.ones(d, dtype=dt), "lm_head.weight": torch.randn(32064, d).to(dt)}
    for i in range(nl):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(d, dtype=dt); w[p + "post_attention_layernorm.weight"] = torch.ones(d, dtype=dt)
        for n in "qkvo": w[p + f"self_attn.{n}_proj.weight"] = (torch.randn(d, d) * 0.02).to(dt)
        w[p + "mlp.gate_proj.weight"] = (torch.randn(dff, d) * 0.02).to(dt); w[p + "mlp.up_proj.weight"] = (torch.randn(dff, d) * 0.02).to(dt)
        w[p + "mlp.down_proj.weight"] = (torch.randn(d, dff) * 0.02).to(dt)
    cache = KVCache([(torch.randn(32, T, 128) * 0.5).to(dt) for _ in range(nl)], [(torch.randn(32, T, 128) * 0.5).to(dt) for _ in range(nl)])
    x = torch.randn(1, d).to(dt)
    def fwd():
        c = cache.clone()
        return lm_hidden(cfg, w, x, torch.tensor([T]), c, torch.ones(T + 1, dtype=torch.long))
    fwd()
    t0 = time.perf_counter(); fwd(); print(dt, nt, "threads: forward of 2 layers", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        fwd()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12))
