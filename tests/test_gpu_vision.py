"""The HIP vision tower + projector against HF's CLIPVisionModel / LlavaMultiModalProjector in fp32 (the same
bf16-valued weights on both sides).  Tolerance: 1e-3 of the largest |feature| (same bar as the LM logits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _bf16_(m):
    for p in m.parameters():
        p.copy_(p.to(torch.bfloat16).float())
    return m


def _ref(vt, proj, px, layer=-2):
    hs = vt(px, output_hidden_states=True).hidden_states[layer][:, 1:]
    return proj(hs) if proj is not None else hs


def close(a, b, rel=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= rel * np.abs(b).max()


@pytest.fixture(scope="module")
def built():
    from dropoutdecoding_amd import build
    build.build()
    return True


def test_tiny_clip_tower_and_projector(built):
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlavaConfig, LlamaConfig
    from transformers.models.llava.modeling_llava import LlavaMultiModalProjector
    from dropoutdecoding_amd.vision import ClipTowerHIP
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=2, image_size=56,
                          patch_size=14, projection_dim=32)
    tc = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, vision_feature_layer=-2, vision_feature_select_strategy="default")
    vt = _bf16_(CLIPVisionModel(vc).eval())
    proj = _bf16_(LlavaMultiModalProjector(cfg).eval())
    for p in proj.parameters():
        p.mul_(4.0)
    _bf16_(proj)
    px = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    want = _ref(vt, proj, px)
    tower = ClipTowerHIP.from_hf(vt, proj, feature_layer=-2)
    got = tower(px.cuda()).cpu()
    assert got.shape == want.shape == (3, 16, 256)
    assert close(got.numpy(), want.numpy()), np.abs(got.numpy() - want.numpy()).max() / np.abs(want.numpy()).max()
    raw = ClipTowerHIP.from_hf(vt, None, feature_layer=-2)
    assert close(raw(px.cuda()).cpu().numpy(), _ref(vt, None, px).numpy())
    with pytest.raises(ValueError):
        tower(torch.zeros(1, 3, 28, 28).cuda())
    tower.close()
    raw.close()


def test_full_size_clip_l_336_tower(built):
    """CLIP-ViT-L/14-336 shapes (24 layers, d=1024, 577 tokens) + the 1024->4096->4096 projector, random init."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from dropoutdecoding_amd.vision import ClipTowerHIP
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                          patch_size=14, projection_dim=768)
    vt = _bf16_(CLIPVisionModel(vc).eval())

    class Proj(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear_1, self.act, self.linear_2 = torch.nn.Linear(1024, 4096), torch.nn.GELU(), torch.nn.Linear(4096, 4096)

        def forward(self, x):
            return self.linear_2(self.act(self.linear_1(x)))
    proj = _bf16_(Proj().eval())
    px = torch.randn(1, 3, 336, 336, generator=torch.Generator().manual_seed(2))
    want = _ref(vt, proj, px)
    tower = ClipTowerHIP.from_hf(vt, proj, feature_layer=-2)
    got = tower(px.cuda())
    torch.cuda.synchronize()
    assert got.shape == (1, 576, 4096)
    assert close(got.cpu().numpy(), want.numpy()), np.abs(got.cpu().numpy() - want.numpy()).max() / np.abs(want.numpy()).max()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pxg = px.cuda()
    s.record()
    for _ in range(5):
        tower(pxg)
    e.record()
    torch.cuda.synchronize()
    print(f"\nCLIP-L/14-336 + projector on own kernels: {s.elapsed_time(e) / 5:.2f} ms per image")
    tower.close()
