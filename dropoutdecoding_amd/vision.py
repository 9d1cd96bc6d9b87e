"""ClipTowerHIP — the CLIP-style vision tower + LLaVA projector on the dd_vit_* kernels.

`ClipTowerHIP.from_hf(vision_tower, projector, feature_layer=-2)` copies an HF CLIPVisionModel (and optionally a
LlavaMultiModalProjector) into the library; `tower(pixel_values)` returns the projected visual tokens
[n_images, P, proj_dim] fp32 — what reference models/llava.py:233-246 computes with third-party modules
(hidden_states[-2][:, 1:] -> multi_modal_projector).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib
from .dropout import _stream

(VT_PATCH, VT_CLASS, VT_POS, VT_PRE_LN_W, VT_PRE_LN_B, VT_LN1_W, VT_LN1_B, VT_WQ, VT_WK, VT_WV, VT_BQ, VT_BK, VT_BV, VT_WO,
 VT_BO, VT_LN2_W, VT_LN2_B, VT_FC1_W, VT_FC1_B, VT_FC2_W, VT_FC2_B, VT_PROJ1_W, VT_PROJ1_B, VT_PROJ2_W, VT_PROJ2_B,
 VT_PATCH_B, VT_POST_LN_W, VT_POST_LN_B) = range(28)
VIT_NO_PRE_LN, VIT_POST_LN, VIT_KEEP_CLASS = 1, 2, 4


class ClipTowerHIP:
    def __init__(self, image_size: int, patch_size: int, hidden: int, intermediate: int, run_layers: int, heads: int,
                 proj_dim: int = 0, act: str = "quick_gelu", ln_eps: float = 1e-5, flags: int = 0):
        if not torch.cuda.is_available():
            raise _lib.DDError("ClipTowerHIP needs a GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.P = (image_size // patch_size) ** 2
        self.n_out = self.P + 1 if flags & VIT_KEEP_CLASS else self.P
        self.image_size, self.hidden, self.proj_dim = image_size, hidden, proj_dim
        self.kp = (3 * patch_size * patch_size + 63) // 64 * 64
        c = _lib.VitConfigC(image_size, patch_size, hidden, intermediate, run_layers, heads, proj_dim,
                            {"quick_gelu": 0, "gelu": 1}[act], ln_eps, flags)
        self._h = C.c_void_p()
        self._use_lock, self._last_use = threading.Lock(), None
        _lib.check(self.lib.dd_vit_create(C.byref(c), C.byref(self._h)), "dd_vit_create")

    def _load(self, tid: int, layer: int, t: torch.Tensor) -> None:
        t = t.detach()
        if t.dim() == 1:
            t = t[None]
        t = t.reshape(t.shape[0], -1).to(torch.bfloat16).contiguous()
        _lib.check(self.lib.dd_vit_load_tensor(self._h, tid, layer, t.view(torch.int16).data_ptr(), t.shape[0], t.shape[1],
                                               1 if t.is_cuda else 0), f"dd_vit_load_tensor({tid},{layer})")

    @classmethod
    def from_hf(cls, vision_tower, projector=None, feature_layer: int = -2) -> "ClipTowerHIP":
        vm = getattr(vision_tower, "vision_model", vision_tower)
        vc = vision_tower.config
        n_total = vc.num_hidden_layers
        run = n_total + feature_layer + 1 if feature_layer < 0 else feature_layer     # hidden_states[-2] = after n-1 layers
        proj_dim = projector.linear_2.out_features if projector is not None else 0
        t = cls(vc.image_size, vc.patch_size, vc.hidden_size, vc.intermediate_size, run, vc.num_attention_heads, proj_dim,
                "quick_gelu" if vc.hidden_act == "quick_gelu" else "gelu", vc.layer_norm_eps)
        emb = vm.embeddings
        w = emb.patch_embedding.weight.detach().reshape(vc.hidden_size, -1)
        t._load(VT_PATCH, 0, torch.nn.functional.pad(w, (0, t.kp - w.shape[1])))
        t._load(VT_CLASS, 0, emb.class_embedding)
        t._load(VT_POS, 0, emb.position_embedding.weight.reshape(1, -1))
        pre = getattr(vm, "pre_layrnorm", None) or getattr(vm, "pre_layernorm")
        t._load(VT_PRE_LN_W, 0, pre.weight)
        t._load(VT_PRE_LN_B, 0, pre.bias)
        for i in range(run):
            l = vm.encoder.layers[i]
            a, m = l.self_attn, l.mlp
            for tid, p in ((VT_LN1_W, l.layer_norm1.weight), (VT_LN1_B, l.layer_norm1.bias), (VT_WQ, a.q_proj.weight),
                           (VT_WK, a.k_proj.weight), (VT_WV, a.v_proj.weight), (VT_BQ, a.q_proj.bias), (VT_BK, a.k_proj.bias),
                           (VT_BV, a.v_proj.bias), (VT_WO, a.out_proj.weight), (VT_BO, a.out_proj.bias),
                           (VT_LN2_W, l.layer_norm2.weight), (VT_LN2_B, l.layer_norm2.bias), (VT_FC1_W, m.fc1.weight),
                           (VT_FC1_B, m.fc1.bias), (VT_FC2_W, m.fc2.weight), (VT_FC2_B, m.fc2.bias)):
                t._load(tid, i, p)
        if projector is not None:
            t._load(VT_PROJ1_W, 0, projector.linear_1.weight)
            t._load(VT_PROJ1_B, 0, projector.linear_1.bias)
            t._load(VT_PROJ2_W, 0, projector.linear_2.weight)
            t._load(VT_PROJ2_B, 0, projector.linear_2.bias)
        return t

    @classmethod
    def from_hf_instructblip(cls, vision_model) -> "ClipTowerHIP":
        """InstructBLIP's EVA ViT-g/14 (HF InstructBlipVisionModel: no pre-LayerNorm, patch bias, fused qkv with bias, GELU,
        post_layernorm over all P + 1 tokens) -> `tower(pixel_values)` = `vision_model(pixel_values).last_hidden_state` in
        fp32: what reference models/instructblip.py:607-612 feeds the Q-Former."""
        vc = vision_model.config
        t = cls(vc.image_size, vc.patch_size, vc.hidden_size, vc.intermediate_size, vc.num_hidden_layers, vc.num_attention_heads, 0,
                "gelu", vc.layer_norm_eps, VIT_NO_PRE_LN | VIT_POST_LN | VIT_KEEP_CLASS)
        d = vc.hidden_size
        emb = vision_model.embeddings
        w = emb.patch_embedding.weight.detach().reshape(d, -1)
        t._load(VT_PATCH, 0, torch.nn.functional.pad(w, (0, t.kp - w.shape[1])))
        t._load(VT_PATCH_B, 0, emb.patch_embedding.bias)
        t._load(VT_CLASS, 0, emb.class_embedding.reshape(-1))
        t._load(VT_POS, 0, emb.position_embedding.reshape(1, -1))
        for i, l in enumerate(vision_model.encoder.layers):
            a, m = l.self_attn, l.mlp
            qkv_w = a.qkv.weight.detach()
            if getattr(a.qkv, "bias", None) is not None:
                qkv_b = a.qkv.bias.detach()
            elif getattr(a, "q_bias", None) is not None:          # transformers 4.44 layout: q_bias / v_bias, no k bias
                qkv_b = torch.cat([a.q_bias.detach(), torch.zeros_like(a.v_bias), a.v_bias.detach()])
            else:
                qkv_b = torch.zeros(3 * d)
            for tid, p in ((VT_LN1_W, l.layer_norm1.weight), (VT_LN1_B, l.layer_norm1.bias), (VT_WQ, qkv_w[:d]), (VT_WK, qkv_w[d:2 * d]),
                           (VT_WV, qkv_w[2 * d:]), (VT_BQ, qkv_b[:d]), (VT_BK, qkv_b[d:2 * d]), (VT_BV, qkv_b[2 * d:]),
                           (VT_WO, a.projection.weight), (VT_BO, a.projection.bias), (VT_LN2_W, l.layer_norm2.weight),
                           (VT_LN2_B, l.layer_norm2.bias), (VT_FC1_W, m.fc1.weight), (VT_FC1_B, m.fc1.bias), (VT_FC2_W, m.fc2.weight),
                           (VT_FC2_B, m.fc2.bias)):
                t._load(tid, i, p)
        t._load(VT_POST_LN_W, 0, vision_model.post_layernorm.weight)
        t._load(VT_POST_LN_B, 0, vision_model.post_layernorm.bias)
        return t

    def __call__(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [n, 3, H, W] (normalised) on the GPU -> [n, P, proj_dim or hidden] fp32."""
        if not pixel_values.is_cuda:
            raise ValueError("pixel_values must be on the GPU")
        px = pixel_values.float().contiguous()
        n = px.shape[0]
        if tuple(px.shape[1:]) != (3, self.image_size, self.image_size):
            raise ValueError(f"expected [n, 3, {self.image_size}, {self.image_size}], got {tuple(px.shape)}")
        out = torch.empty(n, self.n_out, self.proj_dim or self.hidden, dtype=torch.float32, device=px.device)
        with self._use_lock:          # one scratch per tower: calls from several streams / host threads run one after the other
            cur = torch.cuda.current_stream(px.device)
            if self._last_use is not None:
                cur.wait_event(self._last_use)
            _lib.check(self.lib.dd_vit_forward(self._h, px.data_ptr(), n, out.data_ptr(), _stream()), "dd_vit_forward")
            self._last_use = torch.cuda.Event()
            self._last_use.record(cur)
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.dd_vit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# tensor ids of dd_qformer_load_tensor (include/dropdec.h)
(QF_WORD_EMB, QF_POS_EMB, QF_EMB_LN_W, QF_EMB_LN_B, QF_QUERY_TOKENS, QF_PROJ_W, QF_PROJ_B) = range(7)
(QF_SA_WQ, QF_SA_WK, QF_SA_WV, QF_SA_BQ, QF_SA_BK, QF_SA_BV, QF_SA_WO, QF_SA_BO, QF_SA_LN_W, QF_SA_LN_B,
 QF_CA_WQ, QF_CA_BQ, QF_CA_WK, QF_CA_BK, QF_CA_WV, QF_CA_BV, QF_CA_WO, QF_CA_BO, QF_CA_LN_W, QF_CA_LN_B,
 QF_FFQ_W1, QF_FFQ_B1, QF_FFQ_W2, QF_FFQ_B2, QF_FFQ_LN_W, QF_FFQ_LN_B, QF_FFT_W1, QF_FFT_B1, QF_FFT_W2, QF_FFT_B2,
 QF_FFT_LN_W, QF_FFT_LN_B) = range(10, 42)


class QFormerHIP:
    """InstructBLIP's Q-Former + language_projection on the dd_qformer_* kernels: `qf(qformer_input_ids, image_embeds)` =
    `language_projection(qformer(input_ids, query_embeds=query_tokens, encoder_hidden_states=image_embeds)
    .last_hidden_state[:, :Q])[0]` in fp32 — what reference models/instructblip.py:613-633 computes with third-party modules."""

    def __init__(self, hidden: int, heads: int, layers: int, intermediate: int, encoder_hidden: int, cross_freq: int, n_query: int,
                 vocab: int, max_pos: int, proj_dim: int, ln_eps: float = 1e-12, max_text_tokens: int = 128, max_encoder_tokens: int = 257):
        if not torch.cuda.is_available():
            raise _lib.DDError("QFormerHIP needs a GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.hidden, self.n_query, self.proj_dim, self.encoder_hidden = hidden, n_query, proj_dim, encoder_hidden
        self.max_text_tokens, self.max_encoder_tokens = min(max_text_tokens, max_pos), max_encoder_tokens
        c = _lib.QFormerConfigC(hidden, heads, layers, intermediate, encoder_hidden, cross_freq, n_query, vocab, max_pos, proj_dim,
                                self.max_text_tokens, max_encoder_tokens, ln_eps)
        self._h = C.c_void_p()
        self._use_lock, self._last_use = threading.Lock(), None
        _lib.check(self.lib.dd_qformer_create(C.byref(c), C.byref(self._h)), "dd_qformer_create")

    def _load(self, tid: int, layer: int, t: torch.Tensor) -> None:
        t = t.detach()
        if t.dim() == 1:
            t = t[None]
        t = t.reshape(t.shape[0], -1).to(torch.bfloat16).contiguous()
        _lib.check(self.lib.dd_qformer_load_tensor(self._h, tid, layer, t.view(torch.int16).data_ptr(), t.shape[0], t.shape[1],
                                                   1 if t.is_cuda else 0), f"dd_qformer_load_tensor({tid},{layer})")

    @classmethod
    def from_hf(cls, qformer, query_tokens, language_projection, max_text_tokens: int = 128, max_encoder_tokens: int = 257) -> "QFormerHIP":
        qc = qformer.config
        if getattr(qc, "hidden_act", "gelu") != "gelu":
            raise ValueError(f"Q-Former activation {qc.hidden_act!r} is not built (gelu only)")
        if getattr(qc, "position_embedding_type", "absolute") != "absolute":
            raise ValueError("only absolute position embeddings are built")
        t = cls(qc.hidden_size, qc.num_attention_heads, qc.num_hidden_layers, qc.intermediate_size, qc.encoder_hidden_size,
                qc.cross_attention_frequency, query_tokens.shape[-2], qc.vocab_size, qc.max_position_embeddings,
                language_projection.out_features, qc.layer_norm_eps, max_text_tokens, max_encoder_tokens)
        emb = qformer.embeddings
        t._load(QF_WORD_EMB, 0, emb.word_embeddings.weight)
        t._load(QF_POS_EMB, 0, emb.position_embeddings.weight)
        t._load(QF_EMB_LN_W, 0, emb.layernorm.weight)
        t._load(QF_EMB_LN_B, 0, emb.layernorm.bias)
        t._load(QF_QUERY_TOKENS, 0, query_tokens.reshape(query_tokens.shape[-2], -1))
        t._load(QF_PROJ_W, 0, language_projection.weight)
        t._load(QF_PROJ_B, 0, language_projection.bias)
        for i, l in enumerate(qformer.encoder.layer):
            sa, so = l.attention.attention, l.attention.output
            items = [(QF_SA_WQ, sa.query.weight), (QF_SA_WK, sa.key.weight), (QF_SA_WV, sa.value.weight), (QF_SA_BQ, sa.query.bias),
                     (QF_SA_BK, sa.key.bias), (QF_SA_BV, sa.value.bias), (QF_SA_WO, so.dense.weight), (QF_SA_BO, so.dense.bias),
                     (QF_SA_LN_W, so.LayerNorm.weight), (QF_SA_LN_B, so.LayerNorm.bias),
                     (QF_FFQ_W1, l.intermediate_query.dense.weight), (QF_FFQ_B1, l.intermediate_query.dense.bias),
                     (QF_FFQ_W2, l.output_query.dense.weight), (QF_FFQ_B2, l.output_query.dense.bias),
                     (QF_FFQ_LN_W, l.output_query.LayerNorm.weight), (QF_FFQ_LN_B, l.output_query.LayerNorm.bias),
                     (QF_FFT_W1, l.intermediate.dense.weight), (QF_FFT_B1, l.intermediate.dense.bias),
                     (QF_FFT_W2, l.output.dense.weight), (QF_FFT_B2, l.output.dense.bias),
                     (QF_FFT_LN_W, l.output.LayerNorm.weight), (QF_FFT_LN_B, l.output.LayerNorm.bias)]
            if getattr(l, "has_cross_attention", False):
                ca, co = l.crossattention.attention, l.crossattention.output
                items += [(QF_CA_WQ, ca.query.weight), (QF_CA_BQ, ca.query.bias), (QF_CA_WK, ca.key.weight), (QF_CA_BK, ca.key.bias),
                          (QF_CA_WV, ca.value.weight), (QF_CA_BV, ca.value.bias), (QF_CA_WO, co.dense.weight), (QF_CA_BO, co.dense.bias),
                          (QF_CA_LN_W, co.LayerNorm.weight), (QF_CA_LN_B, co.LayerNorm.bias)]
            for tid, p in items:
                t._load(tid, i, p)
        return t

    def __call__(self, text_ids: Optional[torch.Tensor], image_embeds: torch.Tensor, return_hidden: bool = False):
        """text_ids [n] integer instruction tokens (padding already dropped; may be empty / None); image_embeds [n_enc, enc_hidden]
        on the GPU -> [Q, proj_dim] fp32 (and the Q-Former's last hidden state [Q + n, hidden] when asked)."""
        if not image_embeds.is_cuda:
            raise ValueError("image_embeds must be on the GPU")
        enc = image_embeds.float().contiguous()
        if enc.dim() != 2 or enc.shape[1] != self.encoder_hidden:
            raise ValueError(f"expected image_embeds [n, {self.encoder_hidden}], got {tuple(enc.shape)}")
        n = 0 if text_ids is None else int(text_ids.numel())
        ids = None if n == 0 else text_ids.reshape(-1).to(enc.device, torch.int32).contiguous()
        out = torch.empty(self.n_query, self.proj_dim, dtype=torch.float32, device=enc.device)
        hid = torch.empty(self.n_query + n, self.hidden, dtype=torch.float32, device=enc.device) if return_hidden else None
        with self._use_lock:          # one scratch per handle (see ClipTowerHIP.__call__)
            cur = torch.cuda.current_stream(enc.device)
            if self._last_use is not None:
                cur.wait_event(self._last_use)
            _lib.check(self.lib.dd_qformer_forward(self._h, None if ids is None else ids.data_ptr(), n, enc.data_ptr(), enc.shape[0],
                                                   out.data_ptr(), None if hid is None else hid.data_ptr(), _stream()), "dd_qformer_forward")
            self._last_use = torch.cuda.Event()
            self._last_use.record(cur)
        return (out, hid) if return_hidden else out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.dd_qformer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
