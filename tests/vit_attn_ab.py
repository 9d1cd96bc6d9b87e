"""Child process of tests/test_gpu_prefill_kernels.py::test_vision_towers_*: the vision front-ends (CLIP tower + projector, EVA-style tower with
heads of 88 at a pitch of 96, Q-Former with its cross-attention against another key count) under tools key 46 = 0 (fp32-staged attention
tiles) and 1 (operand-staged tiles) must give the same bits.  Run with DD_USE_TOOLS_LIB=1 so that the towers live in libdropdec_tools.so."""
import os
import sys

os.environ["DD_USE_TOOLS_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_grad_enabled(False)
from transformers import (CLIPVisionConfig, CLIPVisionModel, InstructBlipQFormerConfig, InstructBlipVisionConfig, InstructBlipVisionModel,
                          LlamaConfig, LlavaConfig)
from transformers.models.llava.modeling_llava import LlavaMultiModalProjector
from dropoutdecoding_amd import _lib
from dropoutdecoding_amd.vision import ClipTowerHIP

lib = _lib.load()
torch.manual_seed(0)


def both(make, run):
    outs = []
    for key in (0, 1, 0):
        lib.dd_tools_set_tuning(46, key)
        obj = make()
        outs.append(run(obj).clone())
        obj.close()
    lib.dd_tools_set_tuning(46, 1)
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    return outs[0]


def bf16_(m):
    for p in m.parameters():
        p.copy_(p.to(torch.bfloat16).float())
    return m


vc = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, image_size=112, patch_size=14,
                      projection_dim=32)
tc = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2)
cfg = LlavaConfig(vision_config=vc, text_config=tc, vision_feature_layer=-2, vision_feature_select_strategy="default")
vt, proj = bf16_(CLIPVisionModel(vc).eval()), bf16_(LlavaMultiModalProjector(cfg).eval())
px = torch.randn(5, 3, 112, 112, generator=torch.Generator().manual_seed(1)).cuda()
both(lambda: ClipTowerHIP.from_hf(vt, proj, feature_layer=-2), lambda t: t(px))          # 5 images as one matrix (one attention launch)
both(lambda: ClipTowerHIP.from_hf(vt, proj, feature_layer=-2), lambda t: t(px[:1]))      # one image
print("clip ok", flush=True)
ec = InstructBlipVisionConfig(hidden_size=704, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, image_size=56, patch_size=14)
ev = bf16_(InstructBlipVisionModel(ec).eval().cuda())
px2 = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(2)).cuda()
both(lambda: ClipTowerHIP.from_hf_instructblip(ev), lambda t: t(px2))
print("eva ok", flush=True)
