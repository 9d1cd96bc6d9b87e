"""Host-side logic of the drop-in wrappers on CPU (no GPU, no library calls): image-span bookkeeping for both
processor layouts, error behaviour, return layouts, EOS handling, seed bookkeeping, fp8 quantiser round trip."""
import os

import numpy as np
import pytest
import torch

from dropoutdecoding_amd import config as ddc
from dropoutdecoding_amd.vlm import DropoutVLM, lm_state_dict_from_hf


class StubEngine:
    """Plays DropoutEngine for the wrapper: emits a scripted token sequence."""
    device = torch.device("cpu")

    def __init__(self, script):
        self.script, self.n, self.calls = list(script), 0, []

    def prefill(self, embeds, start, L, first_step_ensemble=False, stream=None):
        self.calls.append(("prefill", tuple(embeds.shape), start, L))
        self.n = 1

    def generate(self, n_new, eos=None, dropout=True, **kw):
        eos = set(eos or [])
        out = []
        for t in self.script[:n_new]:
            out.append(t)
            if t in eos:
                break
        self.calls.append(("generate", n_new, tuple(sorted(eos)), dropout))
        return out

    def tokens(self):
        return self.script[:self.n]

    def vision_uncert_dict(self):
        return {"epis_uncert_per_token": np.zeros((1, 4), np.float32)}

    def topk(self):
        return np.zeros((4, 5), np.float32), np.zeros((4, 5), np.int32)


class Wrap(DropoutVLM):
    family = "llava-1.5"

    def _visual_embeds(self, pixel_values=None, **_):
        return torch.full((4, 8), 7.0)


def make(script=(5, 6, 9, 3, 2)):
    emb = torch.arange(20 * 8, dtype=torch.float32).reshape(20, 8)
    return Wrap(StubEngine(script), emb, image_token_index=19, eos_token_id=9)


def test_merge_single_placeholder_and_expanded_run():
    m = make()
    e1, s1 = m._merge(torch.tensor([[1, 19, 3]]), torch.full((4, 8), 7.0))
    e2, s2 = m._merge(torch.tensor([[1, 19, 19, 19, 19, 3]]), torch.full((4, 8), 7.0))
    assert s1 == s2 == 1 and e1.shape == e2.shape == (6, 8) and torch.equal(e1, e2)
    assert torch.equal(e1[1:5], torch.full((4, 8), 7.0)) and torch.equal(e1[0], m.embed_tokens[1]) and torch.equal(e1[5], m.embed_tokens[3])


@pytest.mark.parametrize("ids", [[[1, 2, 3]], [[1, 19, 19, 3]], [[19, 1, 19, 19, 19]]])
def test_merge_rejects_wrong_image_token_counts(ids):
    with pytest.raises(ValueError, match="number of image tokens"):
        make()._merge(torch.tensor(ids), torch.full((4, 8), 7.0))       # reference llava.py:134-138


def test_generate_surface_and_state():
    m = make()
    ids = torch.tensor([[1, 19, 3]])
    out = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=torch.zeros(1), max_new_tokens=5,
                     num_beams=1, pad_token_id=0)
    assert out.tolist() == [[1, 19, 3, 5, 6, 9]]                      # stops on the model's EOS (9), prompt ids first
    assert m.start_image_pos == [1] and m.end_image_pos == [4] and m.start_generation_pos == 6
    assert m.engine.calls[0] == ("prefill", (6, 8), 1, 4)
    assert m.image_features[1].shape == (1, 4, 5) and "epis_uncert_per_token" in m.vision_uncert_dict
    out = m.generate(input_ids=ids, pixel_values=torch.zeros(1), max_new_tokens=4, eos_token_id=[])
    assert out.tolist() == [[1, 19, 3, 5, 6, 9, 3]]                   # EOS ignored, exactly max_new_tokens
    m.original = True
    m.generate(input_ids=ids, pixel_values=torch.zeros(1), max_new_tokens=2)
    assert m.engine.calls[-1][3] is False                            # --original: stock greedy steps


def test_generate_rejects_what_the_reference_cannot_do():
    m = make()
    with pytest.raises(ValueError):
        m.generate(input_ids=torch.tensor([[1, 19], [1, 19]]), pixel_values=torch.zeros(1))      # batch 2
    with pytest.raises(ValueError):
        m.generate(input_ids=torch.tensor([[1, 19]]), pixel_values=torch.zeros(1), num_beams=3)
    with pytest.raises(ValueError):
        m.generate(input_ids=torch.tensor([[1, 19]]), pixel_values=torch.zeros(1), do_sample=True)


def test_instructblip_output_layout():
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration as IB
    emb = torch.arange(20 * 8, dtype=torch.float32).reshape(20, 8)
    m = IB(StubEngine((5, 6, 9)), emb, hf_front=None, eos_token_id=9, config=None)
    m._visual_embeds = lambda **kw: torch.full((4, 8), 7.0)
    out = m.generate(input_ids=torch.tensor([[1, 3, 4]]), pixel_values=torch.zeros(1), max_new_tokens=8)
    assert out.tolist() == [[2, 5, 6, 9]]                             # BOS(2) + new ids only (instructblip.py:686-695)
    assert m.start_image_pos == [0] and m.end_image_pos == [3] and m.engine.calls[0] == ("prefill", (7, 8), 0, 4)


def test_effective_seed_is_the_last_imported_module():
    import importlib
    import dropoutdecoding_amd.instructblip as a
    import dropoutdecoding_amd.llava as b
    import dropoutdecoding_amd.llavanext as c
    for mod, seed in ((b, 24), (c, 506), (a, 5217)):                  # chair_test's import order (chair_test.py:9,12,19)
        importlib.reload(mod)
        assert ddc.effective_seed == seed == mod.seed
    import models.config
    assert models.config.settings is ddc.settings and ddc.settings["voting_numbers"]


def test_lm_state_dict_layouts():
    w = {k: torch.zeros(1) for k in ("language_model.model.embed_tokens.weight", "language_model.model.layers.0.mlp.up_proj.weight",
                                     "language_model.lm_head.weight", "vision_tower.x")}

    class M:
        def state_dict(self):
            return w
    sd = lm_state_dict_from_hf(M())
    assert set(sd) == {"model.embed_tokens.weight", "model.layers.0.mlp.up_proj.weight", "lm_head.weight"}
    w2 = {"model.language_model.embed_tokens.weight": torch.ones(1), "model.language_model.norm.weight": torch.ones(1),
          "model.vision_tower.y": torch.ones(1)}

    class M2:
        def state_dict(self):
            return w2
    sd2 = lm_state_dict_from_hf(M2())
    assert set(sd2) == {"model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"}     # tied head filled in


def test_fp8_quantiser_round_trip():
    from dropoutdecoding_amd.lm import dequantize_fp8, quantize_fp8
    w = torch.randn(32, 64, generator=torch.Generator().manual_seed(0)) * 0.05
    q, s = quantize_fp8(w)
    assert q.dtype == torch.uint8 and s.shape == (32,)
    d = dequantize_fp8(q, s)
    assert float((d - w).abs().max()) <= float(w.abs().max()) / 16 + 1e-9          # e4m3: 3 mantissa bits
    assert float(d.abs().max(dim=1).values.sub(w.abs().max(dim=1).values).abs().max()) < 1e-6   # row maxima are exact


def test_bench_gpus_n_starts_its_own_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` outside a launcher: N ranks through torch.distributed.run on 127.0.0.1, spawned as child processes
    before the parent touches the GPU; inside a launcher (WORLD_SIZE set) nothing is spawned."""
    import importlib
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: (_ for _ in ()).throw(AssertionError("GPU touched before the spawn")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert bench.main() == 7                                     # the children's status is the parent's
    (cmd, env), = calls
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_takes_its_committed_profile_from_the_rounds_own_files(tmp_path, monkeypatch):
    """`roofline.traffic` / `kernel_stats_avg_us` come from profiles/rNN_pmc_summary.json and rNN_kernel_stats.csv of the newest round (config 5:
    rNN_c5_*) — not from any other file that merely ends in the same words (round 4: a tool's r04_prefill_ab_*_kernel_stats.csv sorted last and
    the bench line carried a null)."""
    import importlib
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = "k_gemv_slices_seq<9, 4, 16, 3, 0, 2>"
    for name, avg in (("r03_kernel_stats.csv", 48400.0), ("r04_kernel_stats.csv", 47850.0), ("r04_c5_kernel_stats.csv", 30390.0),
                      ("r04_prefill_ab_mistral_kernel_stats.csv", 1.0)):
        (prof / name).write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                                 f'"void {kern}(SliceArgs)",10,{avg * 10},{avg},1.0,1,2,0.0\n')
    for name, rd in (("r03_pmc_summary.json", 1), ("r04_pmc_summary.json", 184000000), ("r04_c5_pmc_summary.json", 55000000)):
        (prof / name).write_text(json.dumps({"kernels": {f"void {kern}(SliceArgs)": {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": 1000}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench._profile_files() == (os.path.join("profiles", "r04_pmc_summary.json"), os.path.join("profiles", "r04_kernel_stats.csv"))
    assert bench._profile_files("c5_") == (os.path.join("profiles", "r04_c5_pmc_summary.json"), os.path.join("profiles", "r04_c5_kernel_stats.csv"))
    got = bench.committed_profile(kern)
    assert got["stats_avg_us"] == 47.85 and got["traffic"] == 184001000 and got["stats_file"].endswith("r04_kernel_stats.csv")
    assert bench.committed_profile(kern, "c5_")["stats_avg_us"] == 30.39
    (prof / "r04_c5_kernel_stats.csv").unlink()                      # no config-5 trace this round: fall back to the default collection
    assert bench._profile_files("c5_")[1] == os.path.join("profiles", "r04_kernel_stats.csv")


def test_tensor_parallel_shards_reassemble_the_layer():
    """lm.tp_local_config / tp_shard_state_dict: head-aligned slices, d_ff padded to a multiple of 256 with zero rows / columns —
    the ranks' partial MLP and attention-output products add up to the un-sharded layer's."""
    import torch.nn.functional as F
    from dropoutdecoding_amd import lm
    cfg = lm.LMConfig(64, 512, 1280, 1, 4, 2, 128, 1e-5, 10000.0)
    g = torch.Generator().manual_seed(0)
    sd = {"model.embed_tokens.weight": torch.randn(64, 512, generator=g), "model.norm.weight": torch.ones(512), "lm_head.weight": torch.randn(64, 512, generator=g)}
    p = "model.layers.0."
    sd[p + "input_layernorm.weight"] = sd[p + "post_attention_layernorm.weight"] = torch.ones(512)
    sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.o_proj.weight"] = torch.randn(512, 512, generator=g), torch.randn(512, 512, generator=g)
    sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = torch.randn(256, 512, generator=g), torch.randn(256, 512, generator=g)
    sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = torch.randn(1280, 512, generator=g), torch.randn(1280, 512, generator=g)
    sd[p + "mlp.down_proj.weight"] = torch.randn(512, 1280, generator=g)
    local = lm.tp_local_config(cfg, 2)
    assert (local.num_heads, local.num_kv_heads, local.intermediate_size) == (2, 1, 768)      # 640 columns padded to 768
    x = torch.randn(5, 512, generator=g)
    full = F.linear(F.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) * F.linear(x, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
    o_full = F.linear(x, sd[p + "self_attn.o_proj.weight"])
    part, o_part = 0, 0
    for r in range(2):
        sh = lm.tp_shard_state_dict(sd, cfg, r, 2)
        assert sh[p + "mlp.gate_proj.weight"].shape == (768, 512) and sh[p + "mlp.down_proj.weight"].shape == (512, 768)
        assert sh[p + "self_attn.q_proj.weight"].shape == (256, 512) and sh[p + "self_attn.k_proj.weight"].shape == (128, 512)
        assert sh["lm_head.weight"] is sd["lm_head.weight"]
        part = part + F.linear(F.silu(F.linear(x, sh[p + "mlp.gate_proj.weight"])) * F.linear(x, sh[p + "mlp.up_proj.weight"]), sh[p + "mlp.down_proj.weight"])
        o_part = o_part + F.linear(x[:, 256 * r:256 * (r + 1)], sh[p + "self_attn.o_proj.weight"])
    assert torch.allclose(part, full, rtol=1e-4, atol=1e-3) and torch.allclose(o_part, o_full, rtol=1e-4, atol=1e-3)
    with pytest.raises(ValueError):
        lm.tp_local_config(cfg, 4)                        # 2 kv heads do not split over 4 ranks
