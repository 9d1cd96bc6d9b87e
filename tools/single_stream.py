"""One image at a time (the reference's loop): decoded tokens/s over several images, and how often the speculative
single-sweep step held.  python tools/single_stream.py [n_images] [spec 0/1]"""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_inputs
from dropoutdecoding_amd import config as ddcfg, _lib
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration

torch.cuda.set_device(0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ddcfg.settings["voting_numbers"] = ddcfg.VOTING_NUMBERS_K8
model = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=136)
_lib.load().dd_tools_set_tuning(14, spec)
eng = model.engine
n_new = 128
def one(i):
    ids, px = synthetic_inputs(i, eng.cfg.vocab_size, model.image_token_index)
    return model.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), max_new_tokens=n_new, eos_token_id=[])
one(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_img):
    one(1 + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"spec={spec}: {n_img} images, {n_img * n_new / dt:.1f} tok/s, {dt / n_img * 1e3:.1f} ms per image")
# success rate of the speculation on one more image, step by step
ids, px = synthetic_inputs(99, eng.cfg.vocab_size, model.image_token_index)
model._prepare(ids.cuda(), n_new, None, 1, [], False, dict(pixel_values=px.cuda()))
oks = []
for _ in range(64):
    eng.decode_step()
    oks.append(eng.spec_ok())
print("speculation held in", sum(oks), "of", len(oks), "steps")
