"""Writes a LLaVA-1.5 checkpoint directory the way the released `llava-hf/llava-1.5-7b-hf` lies on disk — sharded fp16 safetensors with
the hub's key names (`language_model.model.layers.N...`, `vision_tower.vision_model...`, `multi_modal_projector.linear_K`), a
`model.safetensors.index.json` and a `config.json` — from seeded values, so that the path the reference's harness takes
(`CustomLlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, device_map="auto")`, chair_test/chair_test.py:185-214)
can be exercised without network or real weights.  Test infrastructure only."""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, Tuple

import torch


def llava_config_dict(text: dict, vision: dict, image_token_index: int) -> dict:
    return {
        "architectures": ["LlavaForConditionalGeneration"], "model_type": "llava", "ignore_index": -100,
        "image_token_index": image_token_index, "projector_hidden_act": "gelu", "torch_dtype": "float16",
        "vision_feature_layer": -2, "vision_feature_select_strategy": "default", "tie_word_embeddings": False,
        "text_config": dict({"model_type": "llama", "architectures": ["LlamaForCausalLM"], "rms_norm_eps": 1e-5, "rope_theta": 10000.0,
                             "hidden_act": "silu", "tie_word_embeddings": False, "torch_dtype": "float16", "max_position_embeddings": 4096}, **text),
        "vision_config": dict({"model_type": "clip_vision_model", "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5, "num_channels": 3}, **vision),
    }


LLAVA15_7B_TEXT = {"vocab_size": 32064, "hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 32, "num_attention_heads": 32,
                   "num_key_value_heads": 32}
CLIP_L_336 = {"hidden_size": 1024, "intermediate_size": 4096, "num_hidden_layers": 24, "num_attention_heads": 16, "image_size": 336,
              "patch_size": 14, "projection_dim": 768}


def tensor_specs(text: dict, vision: dict) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """(hub key, shape, kind) of every tensor; kind: 'w' matrix / embedding (N(0, 0.02)), 'n' norm weight (1 + N(0, 0.02)), 'b' bias (N(0, 0.01))."""
    d, ff, V = text["hidden_size"], text["intermediate_size"], text["vocab_size"]
    kvd = d // text["num_attention_heads"] * text["num_key_value_heads"]
    yield "language_model.model.embed_tokens.weight", (V, d), "w"
    for i in range(text["num_hidden_layers"]):
        p = f"language_model.model.layers.{i}."
        yield p + "self_attn.q_proj.weight", (d, d), "w"
        yield p + "self_attn.k_proj.weight", (kvd, d), "w"
        yield p + "self_attn.v_proj.weight", (kvd, d), "w"
        yield p + "self_attn.o_proj.weight", (d, d), "w"
        yield p + "mlp.gate_proj.weight", (ff, d), "w"
        yield p + "mlp.up_proj.weight", (ff, d), "w"
        yield p + "mlp.down_proj.weight", (d, ff), "w"
        yield p + "input_layernorm.weight", (d,), "n"
        yield p + "post_attention_layernorm.weight", (d,), "n"
    yield "language_model.model.norm.weight", (d,), "n"
    yield "language_model.lm_head.weight", (V, d), "w"
    vd, vff = vision["hidden_size"], vision["intermediate_size"]
    n_pos = (vision["image_size"] // vision["patch_size"]) ** 2 + 1
    v = "vision_tower.vision_model."
    yield v + "embeddings.class_embedding", (vd,), "w"
    yield v + "embeddings.patch_embedding.weight", (vd, 3, vision["patch_size"], vision["patch_size"]), "w"
    yield v + "embeddings.position_embedding.weight", (n_pos, vd), "w"
    yield v + "pre_layrnorm.weight", (vd,), "n"
    yield v + "pre_layrnorm.bias", (vd,), "b"
    for i in range(vision["num_hidden_layers"]):
        p = v + f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{nm}.weight", (vd, vd), "w"
            yield p + f"self_attn.{nm}.bias", (vd,), "b"
        yield p + "layer_norm1.weight", (vd,), "n"
        yield p + "layer_norm1.bias", (vd,), "b"
        yield p + "mlp.fc1.weight", (vff, vd), "w"
        yield p + "mlp.fc1.bias", (vff,), "b"
        yield p + "mlp.fc2.weight", (vd, vff), "w"
        yield p + "mlp.fc2.bias", (vd,), "b"
        yield p + "layer_norm2.weight", (vd,), "n"
        yield p + "layer_norm2.bias", (vd,), "b"
    yield v + "post_layernorm.weight", (vd,), "n"
    yield v + "post_layernorm.bias", (vd,), "b"
    yield "multi_modal_projector.linear_1.weight", (d, vd), "w"
    yield "multi_modal_projector.linear_1.bias", (d,), "b"
    yield "multi_modal_projector.linear_2.weight", (d, d), "w"
    yield "multi_modal_projector.linear_2.bias", (d,), "b"


def seeded_tensor(name: str, shape, kind: str, seed: int, device="cpu") -> torch.Tensor:
    """The fp16 values of tensor `name`: a function of (seed, name) only — the test regenerates single tensors to compare."""
    import zlib
    g = torch.Generator(device=device).manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    x = torch.randn(*shape, generator=g, device=device, dtype=torch.float32)
    x = x * (0.02 if kind in ("w", "n") else 0.01)
    if kind == "n":
        x = x + 1.0
    return x.to(torch.float16)


def write_llava_checkpoint(path: str, text: dict = LLAVA15_7B_TEXT, vision: dict = CLIP_L_336, image_token_index: int = 32000, seed: int = 0,
                           shard_bytes: int = 5 * 10 ** 9, device: str = "cpu") -> Dict[str, str]:
    """-> weight_map (key -> shard file).  Tensors are generated on `device` (the GPU for full size) and written shard by shard."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(llava_config_dict(text, vision, image_token_index), f, indent=1)
    specs = list(tensor_specs(text, vision))
    shards, cur, cur_b = [], [], 0
    for name, shape, kind in specs:
        nb = 2
        for s in shape:
            nb *= s
        if cur and cur_b + nb > shard_bytes:
            shards.append(cur)
            cur, cur_b = [], 0
        cur.append((name, shape, kind))
        cur_b += nb
    shards.append(cur)
    weight_map, total = {}, 0
    for i, sh in enumerate(shards):
        fn = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors" if len(shards) > 1 else "model.safetensors"
        blob = {}
        for name, shape, kind in sh:
            t = seeded_tensor(name, shape, kind, seed, device).cpu().contiguous()
            blob[name] = t
            weight_map[name] = fn
            total += t.numel() * 2
        save_file(blob, os.path.join(path, fn), metadata={"format": "pt"})
        del blob
    if len(shards) > 1:
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    return weight_map
