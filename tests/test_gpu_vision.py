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


def test_eva_vit_g_style_tower_tiny_and_full_size(built):
    """InstructBLIP's vision tower (EVA ViT-g/14: heads of 88 padded to a pitch of 96 for the matrix-core attention, patch bias,
    no pre-LayerNorm, GELU, post-LayerNorm over all P + 1 tokens) on own kernels vs HF's InstructBlipVisionModel in fp32:
    what reference models/instructblip.py:607-612 obtains from `self.vision_model(...)`."""
    from transformers import InstructBlipVisionConfig, InstructBlipVisionModel
    from dropoutdecoding_amd.vision import ClipTowerHIP
    for name, kw, n_img in (("tiny", dict(hidden_size=704, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=8,
                                          image_size=56, patch_size=14), 2),
                            ("ViT-g/14-224 (1408 wide, 39 layers, 16 heads of 88)",
                             dict(hidden_size=1408, intermediate_size=6144, num_hidden_layers=39, num_attention_heads=16,
                                  image_size=224, patch_size=14), 1)):
        torch.manual_seed(0)
        vc = InstructBlipVisionConfig(**kw)
        hf = InstructBlipVisionModel(vc).eval().cuda()
        with torch.no_grad():
            for p in hf.parameters():
                p.copy_((p * (2.0 if p.dim() > 1 else 1.0)).to(torch.bfloat16).float())     # bf16-valued weights on both sides
        tower = ClipTowerHIP.from_hf_instructblip(hf)
        px = torch.randn(n_img, 3, vc.image_size, vc.image_size, generator=torch.Generator().manual_seed(1)).cuda()
        with torch.no_grad():
            want = hf(px).last_hidden_state.float()
        got = tower(px)
        assert got.shape == want.shape == (n_img, (vc.image_size // 14) ** 2 + 1, vc.hidden_size)
        err = float((got - want).abs().max() / want.abs().max())
        print(f"\n[EVA tower {name}] max error vs HF fp32: {err:.2e}")
        assert err < 2e-3, (name, err)
        tower.close()
        del hf
        torch.cuda.empty_cache()


def test_qformer_and_language_projection_tiny_and_full_size(built):
    """InstructBLIP's Q-Former + language_projection on own kernels vs HF's InstructBlipQFormerModel + nn.Linear in fp32 —
    what reference models/instructblip.py:613-633 computes — at a tiny shape and at the released shape (768 wide, 12 layers of
    12 heads, cross-attention to 257 x 1408 vision tokens every 2nd layer, 32 queries, projection to 4096).  Also: padded
    instruction tokens dropped by the caller leave the query rows unchanged, and an empty instruction works."""
    from transformers import InstructBlipQFormerConfig
    from transformers.models.instructblip.modeling_instructblip import InstructBlipQFormerModel
    from dropoutdecoding_amd.vision import QFormerHIP
    # weight scale: HF's init (std 0.02) times 3 for the 3-layer case; times 2 at 12 layers — at times 3 the random 12-layer
    # post-LN stack is chaotic (HF fp32 vs HF fp64: 6e-4; 1e-5 relative noise on the vision tokens moves the output by 6e-3),
    # which measures the network, not the kernels.  At times 2: HF fp32 vs fp64 1.4e-6, own kernels vs fp64 1.3e-5.
    for name, kw, Q, n_enc, proj, n_text, wscale in (
            ("tiny", dict(vocab_size=100, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                          encoder_hidden_size=192, cross_attention_frequency=2, max_position_embeddings=64), 8, 17, 256, 5, 3.0),
            ("released shape", dict(vocab_size=30523, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                    intermediate_size=3072, encoder_hidden_size=1408, cross_attention_frequency=2), 32, 257, 4096, 11, 2.0)):
        torch.manual_seed(0)
        qc = InstructBlipQFormerConfig(**kw)
        hf = InstructBlipQFormerModel(qc).eval()                 # reference on the CPU: plain fp32 arithmetic
        lp = torch.nn.Linear(qc.hidden_size, proj)
        qtok = torch.randn(1, Q, qc.hidden_size, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            for p in list(hf.parameters()) + list(lp.parameters()):
                p.copy_((p * (wscale if p.dim() > 1 else 1.0)).to(torch.bfloat16).float())  # bf16-valued weights on both sides
            qtok = qtok.to(torch.bfloat16).float()
        qf = QFormerHIP.from_hf(hf, qtok, lp, max_text_tokens=32, max_encoder_tokens=n_enc)
        g = torch.Generator().manual_seed(1)
        enc = torch.randn(1, n_enc, qc.encoder_hidden_size, generator=g)
        ids = torch.randint(0, qc.vocab_size, (1, n_text), generator=g)

        def hf_run(ids_, mask_):
            with torch.no_grad():
                am = torch.cat([torch.ones(1, Q, dtype=torch.long), mask_], dim=1)
                o = hf(input_ids=ids_, attention_mask=am, query_embeds=qtok, encoder_hidden_states=enc,
                       encoder_attention_mask=torch.ones(1, n_enc, dtype=torch.long), return_dict=True)
                return o.last_hidden_state[0].float(), lp(o.last_hidden_state[:, :Q])[0].float()

        want_hid, want = hf_run(ids, torch.ones_like(ids))
        encg = enc[0].cuda()
        got, got_hid = (t_.cpu() for t_ in qf(ids[0], encg, return_hidden=True))
        assert got.shape == want.shape == (Q, proj) and got_hid.shape == want_hid.shape
        e_hid = float((got_hid - want_hid).abs().max() / want_hid.abs().max())
        err = float((got - want).abs().max() / want.abs().max())
        print(f"\n[Q-Former {name}] max error vs HF fp32: hidden {e_hid:.2e} (query rows {float((got_hid[:Q] - want_hid[:Q]).abs().max()):.2e}, "
              f"instruction rows {float((got_hid[Q:] - want_hid[Q:]).abs().max()):.2e} abs), projected {err:.2e}")
        assert e_hid < 1e-3 and err < 1e-3, (name, e_hid, err)
        # two padded tokens at the end (masked in HF) == the same call without them, on the query rows
        pad_ids = torch.cat([ids, torch.zeros(1, 2, dtype=ids.dtype)], dim=1)
        pad_mask = torch.cat([torch.ones_like(ids), torch.zeros(1, 2, dtype=ids.dtype)], dim=1)
        _, want_pad = hf_run(pad_ids, pad_mask)
        assert float((got - want_pad).abs().max() / want_pad.abs().max()) < 1e-3
        # no instruction at all: HF runs the query rows alone
        got0 = qf(None, encg)
        assert got0.shape == (Q, proj) and bool(torch.isfinite(got0).all())
        # an id outside the vocabulary must not pass silently
        bad = qf(torch.tensor([qc.vocab_size + 5]), encg)
        assert not bool(torch.isfinite(bad).all())
        with pytest.raises(Exception):
            qf(torch.zeros(33, dtype=torch.long), encg)            # over the text capacity
        qf.close()
        del hf
        torch.cuda.empty_cache()


def test_batched_tower_equals_single_image_calls_bitwise(built):
    """Several images per call run through the tower as one matrix (each image padded to whole 128-row blocks, one attention
    launch over the images): the result must be, bit for bit, what one call per image gives — CLIP tower + projector (LLaVA),
    the raw-feature form, and the EVA form (class token kept, post-LayerNorm, heads of 88)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel, InstructBlipVisionConfig, InstructBlipVisionModel, LlavaConfig, LlamaConfig
    from transformers.models.llava.modeling_llava import LlavaMultiModalProjector
    from dropoutdecoding_amd.vision import ClipTowerHIP
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, image_size=112,
                          patch_size=14, projection_dim=32)          # 65 tokens -> 128 rows per image
    tc = LlamaConfig(vocab_size=64, hidden_size=256, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, vision_feature_layer=-2, vision_feature_select_strategy="default")
    vt = _bf16_(CLIPVisionModel(vc).eval())
    proj = _bf16_(LlavaMultiModalProjector(cfg).eval())
    px = torch.randn(19, 3, 112, 112, generator=torch.Generator().manual_seed(1)).cuda()    # 16 + 3: two chunks
    for tower in (ClipTowerHIP.from_hf(vt, proj, feature_layer=-2), ClipTowerHIP.from_hf(vt, None, feature_layer=-2)):
        one = torch.cat([tower(px[i:i + 1]) for i in range(px.shape[0])])
        many = tower(px)
        assert many.shape == one.shape
        assert torch.equal(many, one)
        tower.close()
    assert close(many[:3].cpu().numpy(), _ref(vt, None, px[:3].cpu()).numpy())
    ec = InstructBlipVisionConfig(hidden_size=704, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=8, image_size=56,
                                  patch_size=14)
    ev = InstructBlipVisionModel(ec).eval().cuda()
    for p in ev.parameters():
        p.copy_(p.to(torch.bfloat16).float())
    tower = ClipTowerHIP.from_hf_instructblip(ev)
    px = torch.randn(4, 3, 56, 56, generator=torch.Generator().manual_seed(2)).cuda()
    one = torch.cat([tower(px[i:i + 1]) for i in range(4)])
    many = tower(px)
    assert torch.equal(many, one)
    want = ev(px).last_hidden_state.float()
    assert float((many - want).abs().max() / want.abs().max()) < 1e-3
    tower.close()
