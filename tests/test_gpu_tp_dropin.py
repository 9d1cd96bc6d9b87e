"""Tensor parallelism BEHIND THE BOUNDARY: the drop-in class builds a sharded engine —
`models.llava.CustomLlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, device_map="auto", tp=(rank, world))` — from the
same checkpoint directory the harness loads (chair_test/chair_test.py:185-214), one process per rank, and `generate()` is called unchanged on every
rank.  Two rank processes on the one GPU of a box (gloo, the seams' all-gather staged through the host; on a node the group is "nccl" = RCCL over
xGMI): both must return the tokens of the un-sharded model loaded from the same files, with the scorer's outputs equal to 1e-4."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

import ckpt_util as cu                                  # tests/ckpt_util.py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXT = {"vocab_size": 32064, "hidden_size": 512, "intermediate_size": 1280, "num_hidden_layers": 2, "num_attention_heads": 4, "num_key_value_heads": 4}

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DD_ROOT"])
import torch
import torch.distributed as dist
torch.set_grad_enabled(False)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from dropoutdecoding_amd import config as ddc
ddc.settings["voting_numbers"] = [0.1, 0.3, 0.5, 0.7]
ddc._module_imported(24)
import models.llava as M
m = M.CustomLlavaForConditionalGeneration.from_pretrained(os.environ["DD_CKPT"], torch_dtype=torch.float16, device_map="auto", tp=(rank, world))
assert m.engine.tp_rank is not None and m.engine.tp_rank.world == world
g = torch.Generator().manual_seed(5)
ids = torch.randint(3, 31999, (1, 32), generator=g)
ids[0, 0], ids[0, 5] = 1, 32000
pv = torch.randn(1, 3, 336, 336, generator=g)
out = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv, max_new_tokens=7, num_beams=1, pad_token_id=0, eos_token_id=[])
out2 = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv, max_new_tokens=4, num_beams=1, pad_token_id=0, eos_token_id=[])
res = {"rank": rank, "tokens": out[0, 32:].tolist(), "tokens2": out2[0, 32:].tolist(), "exchanges": m.engine.tp_rank.exchanges,
       "epi": m.vision_uncert_dict["epis_uncert_per_token"].cpu().numpy().reshape(-1).tolist()}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
'''


def test_drop_in_class_builds_a_sharded_engine_and_generates(tmp_path):
    from dropoutdecoding_amd import build, config as ddc
    build.build()
    path = str(tmp_path / "llava-small-layout")
    cu.write_llava_checkpoint(path, TEXT, cu.CLIP_L_336, image_token_index=32000, seed=13, device="cuda", shard_bytes=2 * 10 ** 9)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), DD_ROOT=ROOT, DD_CKPT=path)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=1200)
        assert p.returncode == 0, so[-2000:] + se[-4000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    # the un-sharded model from the same files, this process
    ddc.settings["voting_numbers"] = [0.1, 0.3, 0.5, 0.7]
    ddc._module_imported(24)
    import models.llava as M
    m = M.CustomLlavaForConditionalGeneration.from_pretrained(path, torch_dtype=torch.float16, device_map="auto")
    assert m.engine.tp_rank is None
    m.engine.set_speculation("never")                   # the sharded step is the two-sweep form
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 31999, (1, 32), generator=g)
    ids[0, 0], ids[0, 5] = 1, 32000
    pv = torch.randn(1, 3, 336, 336, generator=g)
    out = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv, max_new_tokens=7, num_beams=1, pad_token_id=0, eos_token_id=[])
    out2 = m.generate(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pv, max_new_tokens=4, num_beams=1, pad_token_id=0, eos_token_id=[])
    epi = m.vision_uncert_dict["epis_uncert_per_token"].cpu().numpy().reshape(-1)
    for o in outs:
        assert o["tokens"] == out[0, 32:].tolist(), (o["rank"], o["tokens"], out[0, 32:].tolist())
        assert o["tokens2"] == out2[0, 32:].tolist()      # the rng stream continued identically into the second image
        np.testing.assert_allclose(np.array(o["epi"]), epi, rtol=2e-3, atol=1e-6)
        assert o["exchanges"] > 0
    assert outs[0]["tokens"] == outs[1]["tokens"] and outs[0]["epi"] == outs[1]["epi"]      # the ranks agree bit for bit
    print(f"\n[tp drop-in] 2 ranks: tokens {outs[0]['tokens']} = un-sharded; {outs[0]['exchanges']} exchanges per rank")
