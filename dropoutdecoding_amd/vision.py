"""ClipTowerHIP — the CLIP-style vision tower + LLaVA projector on the dd_vit_* kernels.

`ClipTowerHIP.from_hf(vision_tower, projector, feature_layer=-2)` copies an HF CLIPVisionModel (and optionally a
LlavaMultiModalProjector) into the library; `tower(pixel_values)` returns the projected visual tokens
[n_images, P, proj_dim] fp32 — what reference models/llava.py:233-246 computes with third-party modules
(hidden_states[-2][:, 1:] -> multi_modal_projector).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .dropout import _stream

(VT_PATCH, VT_CLASS, VT_POS, VT_PRE_LN_W, VT_PRE_LN_B, VT_LN1_W, VT_LN1_B, VT_WQ, VT_WK, VT_WV, VT_BQ, VT_BK, VT_BV, VT_WO,
 VT_BO, VT_LN2_W, VT_LN2_B, VT_FC1_W, VT_FC1_B, VT_FC2_W, VT_FC2_B, VT_PROJ1_W, VT_PROJ1_B, VT_PROJ2_W, VT_PROJ2_B) = range(25)


class ClipTowerHIP:
    def __init__(self, image_size: int, patch_size: int, hidden: int, intermediate: int, run_layers: int, heads: int,
                 proj_dim: int = 0, act: str = "quick_gelu", ln_eps: float = 1e-5):
        if not torch.cuda.is_available():
            raise _lib.DDError("ClipTowerHIP needs a GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.P = (image_size // patch_size) ** 2
        self.image_size, self.hidden, self.proj_dim = image_size, hidden, proj_dim
        self.kp = (3 * patch_size * patch_size + 63) // 64 * 64
        c = _lib.VitConfigC(image_size, patch_size, hidden, intermediate, run_layers, heads, proj_dim,
                            {"quick_gelu": 0, "gelu": 1}[act], ln_eps)
        self._h = C.c_void_p()
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

    def __call__(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [n, 3, H, W] (normalised) on the GPU -> [n, P, proj_dim or hidden] fp32."""
        if not pixel_values.is_cuda:
            raise ValueError("pixel_values must be on the GPU")
        px = pixel_values.float().contiguous()
        n = px.shape[0]
        if tuple(px.shape[1:]) != (3, self.image_size, self.image_size):
            raise ValueError(f"expected [n, 3, {self.image_size}, {self.image_size}], got {tuple(px.shape)}")
        out = torch.empty(n, self.P, self.proj_dim or self.hidden, dtype=torch.float32, device=px.device)
        _lib.check(self.lib.dd_vit_forward(self._h, px.data_ptr(), n, out.data_ptr(), _stream()), "dd_vit_forward")
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
