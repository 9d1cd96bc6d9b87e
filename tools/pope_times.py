"""POPE-shaped load (one generated token per question, 6 questions per image), LLaVA-1.5-7B shapes, synthetic weights:
questions per second with and without settings['reuse_image_prefix']."""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import config as ddcfg
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration

torch.cuda.set_device(0)
ddcfg.settings["voting_numbers"] = ddcfg.VOTING_NUMBERS_K8
model = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=16)
rng = np.random.default_rng(0)
n_img, per = 6, 6
imgs = [torch.randn(1, 3, 336, 336, generator=torch.Generator().manual_seed(i)).cuda() for i in range(n_img)]
head = [1] + rng.integers(3, 31999, size=4).tolist() + [model.image_token_index]
qs = [[torch.tensor([head + rng.integers(3, 31999, size=int(rng.integers(8, 14))).tolist() + [29901]]).cuda() for _ in range(per)]
      for _ in range(n_img)]
for reuse in (False, True, False, True):
    ddcfg.settings["reuse_image_prefix"] = reuse
    model._prefix = None
    outs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n_img):
        for q in qs[i]:
            outs.append(int(model.generate(input_ids=q, pixel_values=imgs[i], max_new_tokens=1, eos_token_id=[])[0, -1]))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"reuse_image_prefix={reuse}: {n_img * per / dt:.1f} questions/s ({dt / (n_img * per) * 1e3:.1f} ms per question), answers {outs[:8]}")

# where a reused question's time goes
eng = model.engine
ddcfg.settings["reuse_image_prefix"] = True
model.generate(input_ids=qs[0][0], pixel_values=imgs[0], max_new_tokens=1, eos_token_id=[])
tail = torch.randn(12, 4096, device="cuda")
for name, fn in (("truncate+extend (12 rows)", lambda: (eng.truncate(581), eng.prefill_extend(tail))),
                 ("whole generate() of a reused question", lambda: model.generate(input_ids=qs[0][1], pixel_values=imgs[0], max_new_tokens=1, eos_token_id=[]))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
