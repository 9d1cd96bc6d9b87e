"""The reference harness' model load at the released layout: a sharded fp16 safetensors checkpoint with the hub's LLaVA-1.5 key names
(seeded values, written on the box — no network, no real weights) through
`models.llava.CustomLlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, device_map="auto")`
exactly as chair_test/chair_test.py:185-214 calls it, then one generate().

The matrices have LLaVA-1.5-7B's shapes (d = 4096, d_ff = 11008, V = 32064, CLIP-L/14-336); the LAYER COUNT is DD_CKPT_LAYERS (default
8: a 4.4 GB checkpoint, half a minute; 32 = the full 13.5 GB model, run once per round: profiles/r04_checkpoint_load_32_layers.log)."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

import ckpt_util as cu                                  # tests/ckpt_util.py

LAYERS = int(os.environ.get("DD_CKPT_LAYERS", "8"))


def test_sharded_fp16_checkpoint_through_from_pretrained(tmp_path):
    from dropoutdecoding_amd import build, config as ddc
    build.build()
    import models.llava as M                             # the drop-in module the harness imports (chair_test.py:9)
    from dropoutdecoding_amd import lm
    text = dict(cu.LLAVA15_7B_TEXT, num_hidden_layers=LAYERS)
    path = str(tmp_path / "llava-1.5-7b-layout")
    t0 = time.time()
    wm = cu.write_llava_checkpoint(path, text, cu.CLIP_L_336, image_token_index=32000, seed=11, device="cuda", shard_bytes=2 * 10 ** 9)
    t_write = time.time() - t0
    files = sorted(os.listdir(path))
    shards = [f for f in files if f.endswith(".safetensors")]
    assert "config.json" in files and len(shards) >= 1
    if LAYERS >= 8:
        assert "model.safetensors.index.json" in files and len(shards) >= 2          # sharded, with the index the hub ships
        idx = json.load(open(os.path.join(path, "model.safetensors.index.json")))
        assert set(idx["weight_map"]) == set(wm)
    ddc.settings["voting_numbers"] = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    ddc._module_imported(24)
    t0 = time.time()
    m = M.CustomLlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, device_map="auto")   # chair_test.py:192-194
    t_load = time.time() - t0
    eng = m.engine
    assert eng.weight_format == "fp16"                   # an fp16 checkpoint keeps its values exactly (weight_format 'auto')
    assert eng.cfg.num_layers == LAYERS and eng.cfg.hidden_size == 4096 and eng.cfg.intermediate_size == 11008 and eng.cfg.vocab_size == 32064
    assert m.tower_hip is not None                        # CLIP tower + projector on own kernels
    # the embedding table the wrapper looks tokens up in is the checkpoint's, bit for bit
    want_embed = cu.seeded_tensor("language_model.model.embed_tokens.weight", (32064, 4096), "w", 11, "cuda")
    assert m.embed_tokens.dtype == torch.float16 and torch.equal(m.embed_tokens, want_embed)
    # one generate() as the harness issues it (chair_test.py:341-346): 32-token prompt with one <image>, 336 x 336 image
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 31999, (1, 32), generator=g)
    ids[0, 0], ids[0, 5] = 1, 32000
    pv = torch.randn(1, 3, 336, 336, generator=g)
    n_new = 6
    t0 = time.time()
    out = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv, max_new_tokens=n_new, num_beams=1, pad_token_id=0,
                     eos_token_id=[])
    t_gen = time.time() - t0
    assert out.shape == (1, 32 + n_new) and out[0, :32].tolist() == ids[0].tolist()
    got_logits = eng.logits().copy()
    got_epi = m.vision_uncert_dict["epis_uncert_per_token"].cpu().numpy()[0].copy()
    assert m.start_image_pos == [5] and m.end_image_pos == [5 + 575] and m.start_generation_pos == 31 + 576
    # an engine loaded WITHOUT the checkpoint files or transformers — the same seeded tensors handed to lm.py's loader under HF's
    # LlamaForCausalLM names — must agree bit for bit: the file format, the index, the key mapping and the dtype handling delivered
    # exactly these values to the packer
    vis = m._visual_embeds(pixel_values=pv)
    emb, start = m._merge(ids.cuda(), vis)
    assert start == 5 and emb.shape == (31 + 576, 4096)
    eng2 = lm.DropoutEngine(eng.cfg, family=lm.FAMILY_LLAVA, max_seq=eng.max_seq, max_visual=576, seed=eng.seed, weight_format="fp16",
                            kv_format=eng.kv_format)
    sd = {}
    for name, shape, kind in cu.tensor_specs(text, cu.CLIP_L_336):
        if name.startswith("language_model."):
            hf_name = name[len("language_model."):]                                   # model.layers..., lm_head.weight
            sd[hf_name] = cu.seeded_tensor(name, shape, kind, 11, "cuda")
    eng2.load_state_dict(sd)
    del sd
    eng2.prefill(emb, start, 576)
    toks2 = eng2.generate(n_new, mprobs=ddc.settings["voting_numbers"], eos=[])
    assert out[0, 32:].tolist() == toks2
    assert np.array_equal(eng2.logits(), got_logits)
    assert np.array_equal(eng2.vision_uncert_dict()["epis_uncert_per_token"].reshape(-1), got_epi.reshape(-1))
    gb = sum(os.path.getsize(os.path.join(path, f)) for f in shards) / 1e9
    print(f"\ncheckpoint: {LAYERS} layers, {len(shards)} shards, {gb:.2f} GB fp16; written in {t_write:.1f} s; from_pretrained {t_load:.1f} s "
          f"({gb / t_load:.2f} GB/s incl. transformers' load and the tile packing); generate({n_new}) {t_gen:.2f} s; tokens {toks2}")
    eng2.close()
