"""Wall-clock phases of one generate() on the GPU (vision front-end / merge / prefill / decode)."""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_inputs
from dropoutdecoding_amd import config as ddcfg
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration
torch.cuda.set_device(0)
ddcfg.settings["voting_numbers"] = ddcfg.VOTING_NUMBERS_K8
m = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=136)
eng = m.engine
def sync(): torch.cuda.synchronize()
for it in range(3):
    ids, px = synthetic_inputs(it, 32064, m.image_token_index)
    sync(); t0 = time.perf_counter()
    ids, px = ids.cuda(), px.cuda()
    vis = m._visual_embeds(pixel_values=px); sync(); t1 = time.perf_counter()
    emb, start = m._merge(ids, vis); sync(); t2 = time.perf_counter()
    eng.prefill(emb, start, 576); sync(); t3 = time.perf_counter()
    toks = eng.generate(128); sync(); t4 = time.perf_counter()
    m._publish_prefill_diagnostics(); sync(); t5 = time.perf_counter()
    print(f"iter {it}: vision {1e3*(t1-t0):.1f} ms, merge {1e3*(t2-t1):.1f}, prefill {1e3*(t3-t2):.1f}, decode(127 steps) {1e3*(t4-t3):.1f} "
          f"({(t4-t3)/127*1e3:.3f} ms/step), diag {1e3*(t5-t4):.1f}", flush=True)
# host-side enqueue cost of one step (no sync)
t0 = time.perf_counter()
for _ in range(20): eng.decode_step()
t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.3f} ms/step; drained in {1e3*(t2-t1):.1f} ms")
