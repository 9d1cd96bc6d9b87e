"""Drop-in for the reference's models/llava.py: CustomLlavaForConditionalGeneration (LLaVA-1.5).

Same class name, `from_pretrained(path, torch_dtype=..., device_map=...)`, `.to()`, and
`.generate(**processor_outputs, max_new_tokens, num_beams, pad_token_id)` returning `[1, S + n_new]`
(reference chair_test/chair_test.py:192-194, 341-346; models/llava.py:155-388).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import config as _config
from .lm import FAMILY_LLAVA, LMConfig
from .vlm import DropoutVLM, build_engine, lm_state_dict_from_hf

seed = 24                          # reference models/llava.py:16
_config._module_imported(seed)


class _Projector(torch.nn.Module):
    """LlavaMultiModalProjector layout (linear_1 -> GELU -> linear_2) for the synthetic-weights model."""

    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.linear_1 = torch.nn.Linear(d_in, d_out)
        self.act = torch.nn.GELU()
        self.linear_2 = torch.nn.Linear(d_out, d_out)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


class CustomLlavaForConditionalGeneration(DropoutVLM):
    family = FAMILY_LLAVA
    supports_prefix_reuse = True        # visual tokens depend on the image only

    def __init__(self, engine, embed_tokens, vision_tower, projector, image_token_index, vision_feature_layer=-2,
                 vision_feature_select_strategy="default", eos_token_id=None, config=None, native_vision: bool = True):
        super().__init__(engine, embed_tokens, image_token_index, eos_token_id, config)
        self.vision_tower, self.multi_modal_projector = vision_tower, projector
        self.vision_feature_layer = vision_feature_layer
        self.vision_feature_select_strategy = vision_feature_select_strategy
        # CLIP tower + projector on the dd_vit_* kernels (vision.py); the HF modules stay available as
        # `vision_tower` / `multi_modal_projector` for callers that poke at them, but are not on the path.
        self.tower_hip = None
        if native_vision and vision_feature_select_strategy == "default" and isinstance(vision_feature_layer, int):
            from .vision import ClipTowerHIP
            self.tower_hip = ClipTowerHIP.from_hf(vision_tower, projector, feature_layer=vision_feature_layer)

    # reference models/llava.py:229-250
    def _visual_embeds(self, pixel_values: Optional[torch.Tensor] = None, **_) -> torch.Tensor:
        if pixel_values is None:
            raise ValueError("pixel_values is required (one image per prompt)")
        if self.tower_hip is not None:
            return self.tower_hip(pixel_values.to(self.device))[0]
        pv = pixel_values.to(self.device, dtype=next(self.vision_tower.parameters()).dtype)
        out = self.vision_tower(pv, output_hidden_states=True)
        feat = out.hidden_states[self.vision_feature_layer]
        if self.vision_feature_select_strategy == "default":
            feat = feat[:, 1:]
        elif self.vision_feature_select_strategy != "full":
            raise ValueError(f"Unexpected select feature strategy: {self.vision_feature_select_strategy}")  # llava.py:241-244
        return self.multi_modal_projector(feat)[0]

    def _visual_embeds_batch(self, inputs_list):
        """One tower call for several images (dd_vit_forward runs up to 16 images as one matrix: 2.0 instead of 5.3 ms per image,
        the same bits as one call per image)."""
        pvs = [inp.get("pixel_values") for inp in inputs_list]
        if (self.tower_hip is None or any(p is None or p.dim() != 4 or p.shape[0] != 1 for p in pvs)
                or len({tuple(p.shape) for p in pvs}) != 1):
            return [self._visual_embeds(**inp) for inp in inputs_list]
        out = self.tower_hip(torch.cat([p.to(self.device) for p in pvs], dim=0))
        return [out[i] for i in range(out.shape[0])]

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_hf_model(cls, hf, max_new_tokens: int = 1024, device="cuda", original: bool = False, tp=None, tp_group=None):
        """Wrap an already-loaded transformers LlavaForConditionalGeneration (4.44 or 5.x attribute layout)."""
        cfg = hf.config
        inner = getattr(hf, "model", hf)
        vt = getattr(hf, "vision_tower", None) or inner.vision_tower
        mp = getattr(hf, "multi_modal_projector", None) or inner.multi_modal_projector
        sd = lm_state_dict_from_hf(hf)
        lm_cfg = LMConfig.from_hf(cfg.text_config)
        vc = cfg.vision_config
        L = (vc.image_size // vc.patch_size) ** 2 + (0 if cfg.vision_feature_select_strategy == "default" else 1)
        eng = build_engine(lm_cfg, cls.family, checkpoint_dtype=sd["lm_head.weight"].dtype, max_visual=L, max_new_tokens=max_new_tokens, seed=_config.effective_seed, tp=tp, tp_group=tp_group)
        eng.load_state_dict(sd)
        dev = eng.device
        embed = sd["model.embed_tokens.weight"].to(dev, torch.float16 if eng.weight_format == "fp16" else torch.bfloat16)   # the engine's 16-bit type
        gen = getattr(hf, "generation_config", None)
        eos = getattr(gen, "eos_token_id", None) if gen is not None else None
        if eos is None:
            eos = getattr(cfg.text_config, "eos_token_id", None)
        m = cls(eng, embed, vt.to(dev).eval(), mp.to(dev).eval(), getattr(cfg, "image_token_index", None) or cfg.image_token_id,
                cfg.vision_feature_layer, cfg.vision_feature_select_strategy, eos, cfg)
        m.original = original
        return m

    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.float16, device_map="auto", max_new_tokens: int = 1024, **kw):
        from transformers import LlavaForConditionalGeneration
        hf = LlavaForConditionalGeneration.from_pretrained(model_path, torch_dtype=torch_dtype, low_cpu_mem_usage=True)
        dev = device_map if isinstance(device_map, (str, torch.device)) and str(device_map) not in ("auto", "balanced") else "cuda"
        return cls.from_hf_model(hf, max_new_tokens=max_new_tokens, device=dev, tp=kw.get("tp"), tp_group=kw.get("tp_group"))

    @classmethod
    def from_synthetic(cls, lm_cfg: Optional[LMConfig] = None, seed: int = 0, max_new_tokens: int = 256,
                       image_token_index: int = 32000, vision_dtype=torch.bfloat16):
        """Random-init weights of the real LLaVA-1.5-7B shapes (bench.py: no network, no checkpoints)."""
        from transformers import CLIPVisionConfig, CLIPVisionModel
        from .lm import LLAVA15_7B
        lm_cfg = lm_cfg or LLAVA15_7B
        vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                              image_size=336, patch_size=14, projection_dim=768)
        eng = build_engine(lm_cfg, cls.family, max_visual=576, max_new_tokens=max_new_tokens, prompt_tokens=64,
                           seed=_config.effective_seed)
        eng.load_synthetic(seed, 0.02)
        dev = eng.device
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.device(dev):
            vt = CLIPVisionModel(vc).to(vision_dtype).eval()
            proj = _Projector(1024, lm_cfg.hidden_size).to(vision_dtype).eval()
        embed = (torch.randn(lm_cfg.vocab_size, 64, generator=g).repeat(1, lm_cfg.hidden_size // 64)).to(dev, torch.bfloat16)
        return cls(eng, embed, vt, proj, image_token_index, -2, "default", None, None)
