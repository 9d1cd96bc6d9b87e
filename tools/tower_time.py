"""CLIP-ViT-L/14-336 + LLaVA projector on own kernels: one image per call vs 16 images per call (ms per image)."""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")      # timing hooks / experiment knobs: libdropdec_tools.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration

m = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=8)
tower = m.tower_hip if hasattr(m, "tower_hip") else m.tower
px = torch.randn(16, 3, 336, 336, generator=torch.Generator().manual_seed(0)).cuda()
for mode in ("one image per call", "16 images per call"):
    def go():
        if mode.startswith("16"):
            return tower(px)
        return torch.cat([tower(px[i:i + 1]) for i in range(16)])
    a = go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t0) / 3 / 16 * 1e3:.2f} ms per image", flush=True)
