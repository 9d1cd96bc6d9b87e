"""Drop-in for the reference's models/llavanext.py: CustomLlavaNextForConditionalGeneration (LLaVA-NeXT, anyres).

Differences from LLaVA-1.5 carried over from the reference: the mask is reset before every
member (models/llavanext.py:546), top-10 ids for the keep set (:652), `settings['use_random'][0]` selects
"epis_no_overlap" (:547-550), `logits_mask_prob` diagnostic (:584).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import config as _config
from .config import settings
from .lm import FAMILY_NEXT, LMConfig
from .vlm import DropoutVLM, build_engine, lm_state_dict_from_hf

seed = 506                         # reference models/llavanext.py:18
_config._module_imported(seed)


class CustomLlavaNextForConditionalGeneration(DropoutVLM):
    family = FAMILY_NEXT
    supports_prefix_reuse = True        # visual tokens depend on the image only

    def __init__(self, engine, embed_tokens, hf_vision, image_token_index, eos_token_id=None, config=None,
                 native_vision: bool = True):
        super().__init__(engine, embed_tokens, image_token_index, eos_token_id, config)
        self._hf = hf_vision            # HF (Llava)NextModel stripped of its language model: tiles -> CLIP -> projector -> unpad/newline packing
        # per-tile CLIP tower + projector on the dd_vit_* kernels; the anyres unpad / image_newline packing is index
        # arithmetic on the resulting [tiles, 576, d] tensor and stays HF's pack_image_features
        self.tower_hip = None
        if (native_vision and config is not None and config.vision_feature_select_strategy == "default"
                and isinstance(config.vision_feature_layer, int)):
            from .vision import ClipTowerHIP
            self.tower_hip = ClipTowerHIP.from_hf(hf_vision.vision_tower, hf_vision.multi_modal_projector,
                                                  feature_layer=config.vision_feature_layer)

    # reference models/llavanext.py:388-443 (per-tile CLIP, projector, pack_image_features with image_newline)
    def _visual_embeds(self, pixel_values: Optional[torch.Tensor] = None, image_sizes: Optional[torch.Tensor] = None, **_):
        if pixel_values is None or image_sizes is None:
            raise ValueError("pixel_values and image_sizes are required")
        if self.tower_hip is not None:
            from transformers.models.llava_next.modeling_llava_next import image_size_to_num_patches
            n_tiles = image_size_to_num_patches(image_size=image_sizes[0], grid_pinpoints=self.config.image_grid_pinpoints,
                                                patch_size=self.config.vision_config.image_size)
            pv = pixel_values.to(self.device)
            tiles = pv[0][:n_tiles] if pv.dim() == 5 else pv[:n_tiles]
            feats = self.tower_hip(tiles)                                  # [tiles, 576, d] fp32
            packed, _ = self._hf.pack_image_features([feats], image_sizes.to(self.device),
                                                     vision_feature_select_strategy="default",
                                                     image_newline=self._hf.image_newline.float())
            return torch.cat(list(packed), dim=0) if isinstance(packed, (list, tuple)) else packed
        dt = next(self._hf.vision_tower.parameters()).dtype
        out = self._hf.get_image_features(pixel_values.to(self.device, dt), image_sizes.to(self.device),
                                          vision_feature_layer=self.config.vision_feature_layer,
                                          vision_feature_select_strategy=self.config.vision_feature_select_strategy)
        feats = out.pooler_output if hasattr(out, "pooler_output") else out
        if isinstance(feats, (list, tuple)):
            feats = torch.cat(list(feats), dim=0)
        return feats

    def _visual_embeds_batch(self, inputs_list):
        """One tower call for the anyres tiles of SEVERAL images (dd_vit_forward takes up to 16 tiles as one matrix: three 5-tile images
        = 9,600 rows instead of 3,200 per call, so the tower's GEMMs launch three times the workgroups — an N = 1024 projection over one
        image's tiles is 40 workgroups on 256 CUs); every tile goes through the tower on its own rows, so each image's tokens are bit for
        bit those of its own call.  The unpad / newline packing stays per image (HF index arithmetic)."""
        if self.tower_hip is None or any(inp.get("pixel_values") is None or inp.get("image_sizes") is None for inp in inputs_list):
            return [self._visual_embeds(**inp) for inp in inputs_list]
        from transformers.models.llava_next.modeling_llava_next import image_size_to_num_patches
        tiles = []
        for inp in inputs_list:
            n = image_size_to_num_patches(image_size=inp["image_sizes"][0], grid_pinpoints=self.config.image_grid_pinpoints,
                                          patch_size=self.config.vision_config.image_size)
            pv = inp["pixel_values"].to(self.device)
            tiles.append(pv[0][:n] if pv.dim() == 5 else pv[:n])
        out, i = [None] * len(inputs_list), 0
        while i < len(inputs_list):
            j, total = i, 0
            while j < len(inputs_list) and total + tiles[j].shape[0] <= 16 and (j == i or tiles[j].shape[1:] == tiles[i].shape[1:]):
                total += tiles[j].shape[0]
                j += 1
            if j == i:                                                    # (an image of more than 16 tiles: its own path)
                out[i] = self._visual_embeds(**inputs_list[i])
                i += 1
                continue
            feats = self.tower_hip(torch.cat(tiles[i:j], dim=0))          # [tiles of images i..j-1, 576, d] fp32
            at = 0
            for k in range(i, j):
                f = feats[at:at + tiles[k].shape[0]]
                at += tiles[k].shape[0]
                packed, _ = self._hf.pack_image_features([f], inputs_list[k]["image_sizes"].to(self.device),
                                                         vision_feature_select_strategy="default",
                                                         image_newline=self._hf.image_newline.float())
                out[k] = torch.cat(list(packed), dim=0) if isinstance(packed, (list, tuple)) else packed
            i = j
        return out

    def _decode_loop(self, n_new, eos, chunk: int = 16):
        # settings['use_random'] is read at every step in the reference (llavanext.py:547); the engine's mode is
        # fixed per sequence, which is equivalent as long as the flag does not change mid-generation.
        return super()._decode_loop(n_new, eos, chunk)

    @classmethod
    def from_hf_model(cls, hf, max_new_tokens: int = 1024, max_visual: int = 2944, original: bool = False, tp=None, tp_group=None):
        cfg = hf.config
        sd = lm_state_dict_from_hf(hf)
        lm_cfg = LMConfig.from_hf(cfg.text_config)
        eng = build_engine(lm_cfg, cls.family, checkpoint_dtype=sd["lm_head.weight"].dtype, max_visual=max_visual, max_new_tokens=max_new_tokens,
                           use_random=bool(settings["use_random"][0]), seed=_config.effective_seed, tp=tp, tp_group=tp_group)
        eng.load_state_dict(sd)
        dev = eng.device
        embed = sd["model.embed_tokens.weight"].to(dev, torch.float16 if eng.weight_format == "fp16" else torch.bfloat16)   # the engine's 16-bit type
        inner = getattr(hf, "model", hf)
        inner.language_model = None                      # the LM now lives in the engine
        inner = inner.to(dev).eval()
        gen = getattr(hf, "generation_config", None)
        eos = getattr(gen, "eos_token_id", None) if gen is not None else None
        if eos is None:
            eos = getattr(cfg.text_config, "eos_token_id", None)
        m = cls(eng, embed, inner, getattr(cfg, "image_token_index", None) or cfg.image_token_id, eos, cfg)
        m.original = original
        return m

    @classmethod
    def from_synthetic(cls, lm_cfg: Optional[LMConfig] = None, seed: int = 0, max_new_tokens: int = 256, weight_format: str = "fp8"):
        """Random-init weights of the real LLaVA-NeXT-Mistral-7B shapes (bench.py --config 5: no network, no checkpoints): CLIP-L/14-336
        tower + projector over the anyres tiles (a 672 x 672 image: base + 2 x 2 tiles -> 576 + 48 x 48 + 48 newline = 2928 visual
        tokens), Mistral-7B language model (GQA 4) with fp8 matrices by default (BASELINE config 5).  HF's LlavaNextModel supplies the
        unpad / image_newline packing (index arithmetic); its language model is a one-layer stand-in that is dropped."""
        from transformers import CLIPVisionConfig, LlavaNextConfig, MistralConfig
        from transformers.models.llava_next.modeling_llava_next import LlavaNextModel
        from .lm import MISTRAL_7B
        lm_cfg = lm_cfg or MISTRAL_7B
        vc = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                              patch_size=14, projection_dim=768)
        tc = MistralConfig(vocab_size=64, hidden_size=lm_cfg.hidden_size, intermediate_size=256, num_hidden_layers=1, num_attention_heads=32,
                           num_key_value_heads=8, head_dim=128)
        cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=32000, vision_feature_layer=-2,
                              vision_feature_select_strategy="default",
                              image_grid_pinpoints=[[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]])
        saved = settings.get("weight_format", None)
        settings["weight_format"] = weight_format
        try:
            eng = build_engine(lm_cfg, cls.family, max_visual=2944, max_new_tokens=max_new_tokens, prompt_tokens=64,
                               use_random=bool(settings["use_random"][0]), seed=_config.effective_seed)
        finally:
            if saved is None:
                settings.pop("weight_format", None)
            else:
                settings["weight_format"] = saved
        eng.load_synthetic(seed, 0.02)
        dev = eng.device
        torch.manual_seed(seed)
        with torch.device(dev):
            inner = LlavaNextModel(cfg).to(torch.bfloat16).eval()
        inner.language_model = None
        g = torch.Generator(device="cpu").manual_seed(seed)
        embed = (torch.randn(lm_cfg.vocab_size, 64, generator=g).repeat(1, lm_cfg.hidden_size // 64)).to(dev, torch.bfloat16)
        return cls(eng, embed, inner, 32000, None, cfg)

    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.float16, device_map="auto", max_new_tokens: int = 1024, **kw):
        from transformers import LlavaNextForConditionalGeneration
        hf = LlavaNextForConditionalGeneration.from_pretrained(model_path, torch_dtype=torch_dtype, low_cpu_mem_usage=True)
        return cls.from_hf_model(hf, max_new_tokens=max_new_tokens, tp=kw.get("tp"), tp_group=kw.get("tp_group"))
