"""The checkpoint writer of tests/ckpt_util.py produces what transformers loads as the released LLaVA-1.5 layout (hub key names, shards +
index), and `lm_state_dict_from_hf` hands the language model's tensors to the engine loader under LlamaForCausalLM names.  CPU only; the
GPU half is tests/test_gpu_checkpoint_load.py."""
import json
import os

import torch

import ckpt_util as cu


def test_hub_named_sharded_checkpoint_loads_and_maps(tmp_path):
    from transformers import LlavaForConditionalGeneration
    from dropoutdecoding_amd.vlm import lm_state_dict_from_hf
    from dropoutdecoding_amd.lm import LMConfig
    text = {"vocab_size": 200, "hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 2, "num_attention_heads": 1, "num_key_value_heads": 1}
    vis = {"hidden_size": 32, "intermediate_size": 64, "num_hidden_layers": 3, "num_attention_heads": 2, "image_size": 56, "patch_size": 14, "projection_dim": 16}
    path = str(tmp_path / "ckpt")
    wm = cu.write_llava_checkpoint(path, text, vis, image_token_index=199, seed=3, shard_bytes=100_000)
    files = sorted(os.listdir(path))
    assert "model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) > 2
    idx = json.load(open(os.path.join(path, "model.safetensors.index.json")))
    assert idx["weight_map"] == wm and all(k.split(".")[0] in ("language_model", "vision_tower", "multi_modal_projector") for k in wm)
    hf, info = LlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, low_cpu_mem_usage=True, output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info["mismatched_keys"]
    sd = lm_state_dict_from_hf(hf)
    specs = {n: (s, k) for n, s, k in cu.tensor_specs(text, vis)}
    lm_names = [n for n in specs if n.startswith("language_model.")]
    assert len(sd) == len(lm_names)
    for n in lm_names:
        t = sd[n[len("language_model."):]]
        assert t.dtype == torch.float16 and torch.equal(t, cu.seeded_tensor(n, *specs[n], 3)), n
    c = LMConfig.from_hf(hf.config.text_config)
    assert (c.vocab_size, c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads, c.num_kv_heads, c.head_dim) == (200, 128, 256, 2, 1, 1, 128)
