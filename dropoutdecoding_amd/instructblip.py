"""Drop-in for the reference's models/instructblip.py: CustomInstructBlipForConditionalGeneration.

Reference behaviour carried over: the 32 Q-Former query embeddings are the visual span
at positions 0..31 (models/instructblip.py:599-602, 648-651); deterministic top-quantile masks reset per member
(:121, 447-460); the vote is over the argmax of the members' final HIDDEN state (:125-137) and lm_head is applied to
the winner; the last member's zeros leak into the next step's un-masked pass (:111-113, 121-122); generate() returns
`[BOS(2)] ‖ new ids` (:686-695).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import config as _config
from .lm import FAMILY_IBLIP, LMConfig
from .vlm import DropoutVLM, build_engine, lm_state_dict_from_hf

seed = 5217                        # reference models/instructblip.py:17
_config._module_imported(seed)


class CustomInstructBlipForConditionalGeneration(DropoutVLM):
    family = FAMILY_IBLIP

    def __init__(self, engine, embed_tokens, hf_front, eos_token_id=None, config=None):
        super().__init__(engine, embed_tokens, image_token_index=-1, eos_token_id=eos_token_id, config=config)
        self._hf = hf_front             # vision_model + qformer + language_projection (+ query_tokens)
        self.tower_hip = None           # EVA ViT-g/14 on own kernels (vision.ClipTowerHIP.from_hf_instructblip)
        self.qformer_hip = None         # Q-Former + language_projection on own kernels (vision.QFormerHIP.from_hf)

    # reference models/instructblip.py:607-633
    def _visual_embeds(self, pixel_values=None, qformer_input_ids=None, qformer_attention_mask=None,
                       interpolate_pos_encoding: bool = False, **_):
        if pixel_values is None:
            raise ValueError("pixel_values is required")
        hf, dev = self._hf, self.device
        if self.tower_hip is not None and not interpolate_pos_encoding:
            qdt = next(hf.qformer.parameters()).dtype
            image_embeds = self.tower_hip(pixel_values.to(dev).float()).to(qdt)          # [1, 257, 1408]: fp32-grade, own kernels
        else:
            dt = next(hf.vision_model.parameters()).dtype
            image_embeds = hf.vision_model(pixel_values.to(dev, dt), return_dict=True,
                                           interpolate_pos_encoding=interpolate_pos_encoding).last_hidden_state
        if (self.qformer_hip is not None and image_embeds.shape[0] == 1 and image_embeds.shape[1] <= self.qformer_hip.max_encoder_tokens
                and qformer_input_ids.shape[0] == 1):
            ids = qformer_input_ids[0].to(dev)
            if qformer_attention_mask is not None:
                ids = ids[qformer_attention_mask[0].to(dev).bool()]      # masked keys contribute nothing: drop the padding
            if ids.numel() <= self.qformer_hip.max_text_tokens:
                return self.qformer_hip(ids, image_embeds[0].float())    # [Q, d_lm] fp32
        image_attention_mask = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=dev)
        query_tokens = hf.query_tokens.expand(image_embeds.shape[0], -1, -1)
        query_attention_mask = torch.ones(query_tokens.size()[:-1], dtype=torch.long, device=dev)
        qformer_input_ids = qformer_input_ids.to(dev)
        if qformer_attention_mask is None:
            qformer_attention_mask = torch.ones_like(qformer_input_ids)
        qam = torch.cat([query_attention_mask, qformer_attention_mask.to(dev)], dim=1)
        q = hf.qformer(input_ids=qformer_input_ids, attention_mask=qam, query_embeds=query_tokens,
                       encoder_hidden_states=image_embeds, encoder_attention_mask=image_attention_mask, return_dict=True)
        query_output = q.last_hidden_state[:, : query_tokens.size(1), :]
        return hf.language_projection(query_output)[0]

    def _visual_embeds_batch(self, inputs_list):
        """One EVA tower call for several images (dd_vit_forward runs them as one matrix: the same bits as one call per image),
        then the Q-Former per image (its instruction tokens differ per image)."""
        pvs = [inp.get("pixel_values") for inp in inputs_list]
        ok = (self.tower_hip is not None and self.qformer_hip is not None and len(inputs_list) > 1
              and all(p is not None and p.dim() == 4 and p.shape[0] == 1 for p in pvs) and len({tuple(p.shape) for p in pvs}) == 1
              and all(inp.get("qformer_input_ids") is not None and inp["qformer_input_ids"].shape[0] == 1
                      and not inp.get("interpolate_pos_encoding", False) for inp in inputs_list))
        if not ok:
            return [self._visual_embeds(**inp) for inp in inputs_list]
        dev = self.device
        qdt = next(self._hf.qformer.parameters()).dtype
        embeds = self.tower_hip(torch.cat([p.to(dev).float() for p in pvs], dim=0)).to(qdt)       # [n, 257, 1408]
        out = []
        for inp, e in zip(inputs_list, embeds):
            ids = inp["qformer_input_ids"][0].to(dev)
            m = inp.get("qformer_attention_mask")
            if m is not None:
                ids = ids[m[0].to(dev).bool()]
            if ids.numel() > self.qformer_hip.max_text_tokens or e.shape[0] > self.qformer_hip.max_encoder_tokens:
                out.append(self._visual_embeds(**inp))
            else:
                out.append(self.qformer_hip(ids, e.float()))
        return out

    # reference models/instructblip.py:661-664: [query embeds ; prompt embeds], span = positions 0..Q-1
    def _merge(self, input_ids, visual):
        ids = input_ids[0]
        tid = getattr(self.config, "image_token_id", None) if self.config is not None else None
        if tid is not None:
            ids = ids[ids != tid]               # newer processors prepend Q placeholders; the reference's did not
        emb = torch.nn.functional.embedding(ids, self.embed_tokens).float()
        return torch.cat([visual.float(), emb], dim=0), 0

    def _format_output(self, input_ids, new):
        bos = 2                                  # instructblip.py:686-692 (LLaMA tokenizer files: </s> id 2)
        if self.config is not None:
            arch = (getattr(self.config.text_config, "architectures", None) or ["LLaMAForCausalLM"])[0]
            if arch != "LLaMAForCausalLM":
                bos = self.config.text_config.bos_token_id
        return torch.cat([torch.tensor([[bos]], dtype=torch.long, device=new.device), new], dim=1)

    @classmethod
    def from_hf_model(cls, hf, max_new_tokens: int = 1024, original: bool = False, tp=None, tp_group=None):
        cfg = hf.config
        sd = lm_state_dict_from_hf(hf)
        lm_cfg = LMConfig.from_hf(cfg.text_config)
        eng = build_engine(lm_cfg, cls.family, checkpoint_dtype=sd["lm_head.weight"].dtype, max_visual=cfg.num_query_tokens, max_new_tokens=max_new_tokens,
                           seed=_config.effective_seed, tp=tp, tp_group=tp_group)
        eng.load_state_dict(sd)
        dev = eng.device
        embed = sd["model.embed_tokens.weight"].to(dev, torch.float16 if eng.weight_format == "fp16" else torch.bfloat16)   # the engine's 16-bit type
        inner = getattr(hf, "model", hf)
        if not hasattr(inner, "qformer"):
            inner = hf
        inner.language_model = None
        inner = inner.to(dev).eval()
        gen = getattr(hf, "generation_config", None)
        eos = getattr(gen, "eos_token_id", None) if gen is not None else None
        if eos is None:
            eos = getattr(cfg.text_config, "eos_token_id", None)
        m = cls(eng, embed, inner, eos, cfg)
        m.original = original
        vc = cfg.vision_config
        if vc.hidden_size % 64 == 0 and vc.intermediate_size % 64 == 0 and vc.hidden_size // vc.num_attention_heads in (64, 88):
            from .vision import ClipTowerHIP
            m.tower_hip = ClipTowerHIP.from_hf_instructblip(inner.vision_model)
        qc = cfg.qformer_config
        if (qc.hidden_size == 64 * qc.num_attention_heads and qc.intermediate_size % 64 == 0 and qc.encoder_hidden_size % 64 == 0
                and lm_cfg.hidden_size % 16 == 0 and getattr(qc, "hidden_act", "gelu") == "gelu"):
            from .vision import QFormerHIP
            n_enc = (vc.image_size // vc.patch_size) ** 2 + 1
            m.qformer_hip = QFormerHIP.from_hf(inner.qformer, inner.query_tokens, inner.language_projection, max_encoder_tokens=n_enc)
        return m

    @classmethod
    def from_synthetic(cls, lm_cfg: Optional[LMConfig] = None, seed: int = 0, max_new_tokens: int = 256):
        """Random-init weights of the real InstructBLIP-Vicuna-7B shapes (bench.py --config 4: no network, no checkpoints): EVA
        ViT-g/14-224 tower (1408 wide, 39 layers, 16 heads of 88), Q-Former (768 wide, 12 layers, cross-attention every second layer,
        32 queries), language projection 768 -> 4096, Vicuna-7B language model — the front-end on the dd_vit_* / dd_qformer_* kernels."""
        import types
        from transformers import InstructBlipQFormerConfig, InstructBlipVisionConfig
        from transformers.models.instructblip.modeling_instructblip import InstructBlipQFormerModel, InstructBlipVisionModel
        from .lm import VICUNA_7B
        from .vision import ClipTowerHIP, QFormerHIP
        lm_cfg = lm_cfg or VICUNA_7B
        vc, qc = InstructBlipVisionConfig(), InstructBlipQFormerConfig()
        eng = build_engine(lm_cfg, cls.family, max_visual=32, max_new_tokens=max_new_tokens, prompt_tokens=64, seed=_config.effective_seed)
        eng.load_synthetic(seed, 0.02)
        dev = eng.device
        torch.manual_seed(seed)
        with torch.device(dev):
            front = types.SimpleNamespace(vision_model=InstructBlipVisionModel(vc).eval(), qformer=InstructBlipQFormerModel(qc).eval(),
                                          language_projection=torch.nn.Linear(qc.hidden_size, lm_cfg.hidden_size),
                                          query_tokens=torch.nn.Parameter(torch.randn(1, 32, qc.hidden_size)))
        g = torch.Generator(device="cpu").manual_seed(seed)
        embed = (torch.randn(lm_cfg.vocab_size, 64, generator=g).repeat(1, lm_cfg.hidden_size // 64)).to(dev, torch.bfloat16)
        m = cls(eng, embed, front, None, None)
        m.tower_hip = ClipTowerHIP.from_hf_instructblip(front.vision_model)
        m.qformer_hip = QFormerHIP.from_hf(front.qformer, front.query_tokens, front.language_projection,
                                           max_encoder_tokens=(vc.image_size // vc.patch_size) ** 2 + 1)
        return m

    @classmethod
    def from_pretrained(cls, model_path, torch_dtype=torch.float16, device_map="auto", max_new_tokens: int = 1024, **kw):
        from transformers import InstructBlipForConditionalGeneration
        hf = InstructBlipForConditionalGeneration.from_pretrained(model_path, torch_dtype=torch_dtype, low_cpu_mem_usage=True)
        return cls.from_hf_model(hf, max_new_tokens=max_new_tokens, tp=kw.get("tp"), tp_group=kw.get("tp_group"))
