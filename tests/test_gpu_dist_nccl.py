"""K-shard decoding over torch.distributed's "nccl" backend (= RCCL on ROCm) with REAL engines: the collectives run on the
engine's stream between its phase kernels.  One GPU per box here, so world 1 (every rank-count-independent line of the RCCL
code path: process group on the device, all_reduce of the id and winner records on a side stream, ordering against the phase
kernels); the protocol at world 2 / 3 is covered with gloo in tests/test_dist_gloo.py and the phase kernels of two engines on
one GPU in tests/test_gpu_wrappers.py::test_kshard_phase_kernels_two_engines.  Runs in a child process so that the process
group never outlives the test."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DD_ROOT"])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from dropoutdecoding_amd import build
build.build()
from dropoutdecoding_amd import lm
from dropoutdecoding_amd.dist import KShardDecoder
from oracle.decode_ref import FAMILY_LLAVA, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights
rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
w = random_weights(rc, 1234, 0.05)
probs = [0.1, 0.3, 0.5, 0.7]
emb = torch.randn(30, 256, generator=torch.Generator().manual_seed(5))
want = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=5217).generate(emb, 2, 20, 9)
eng = lm.DropoutEngine(lm.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=lm.FAMILY_LLAVA, max_seq=128, max_visual=32, seed=5217)
eng.load_state_dict(w)
eng.prefill(emb.cuda(), 2, 20)
ks = KShardDecoder(eng, dist.get_rank(), dist.get_world_size(), time_exchange=True)
got = ks.generate(9, probs)
x = ks.exchange_ms()
logits = eng.logits()
eng.rng.manual_seed(5217)
eng.prefill(emb.cuda(), 2, 20)
solo = eng.generate(9, mprobs=probs)
same_logits = bool((eng.logits() == logits).all())
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"want": want, "got": got, "solo": solo, "same_logits": same_logits, "exchange": x}))
'''


def test_kshard_decoder_over_rccl_world_1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DD_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res["got"] == res["want"] == res["solo"]
    assert res["same_logits"]
    x = res["exchange"]
    assert x["backend"] == "nccl" and x["world"] == 1 and x["tokens"] >= 8 and x["ms_per_token"] > 0
    print(f"\n[K-shard over RCCL, world 1] {x['ms_per_token']:.3f} ms per token for the two all-reduces + export / import kernels "
          f"({x['bytes_per_token']} bytes per token)")


TP_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DD_ROOT"])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from dropoutdecoding_amd import lm
from dropoutdecoding_amd.dist import TensorParallelRank
from oracle.decode_ref import FAMILY_LLAVA, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights
rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
w = random_weights(rc, 1234, 0.05)
probs = [0.1, 0.3, 0.5, 0.7]
emb = torch.randn(30, 256, generator=torch.Generator().manual_seed(5))
want = RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=5217).generate(emb, 2, 20, 9)
tp = TensorParallelRank(lm.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), dist.get_rank(), dist.get_world_size(), family=lm.FAMILY_LLAVA,
                        max_seq=128, max_visual=32, seed=5217)
tp.load_state_dict(w)
tp.prefill(emb.cuda(), 2, 20)
got = tp.generate(9, probs)
logits = tp.engine.logits()
eng = lm.DropoutEngine(lm.LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0), family=lm.FAMILY_LLAVA, max_seq=128, max_visual=32, seed=5217)
eng.load_state_dict(w)
eng.set_speculation("never")
eng.prefill(emb.cuda(), 2, 20)
solo = eng.generate(9, mprobs=probs)
same = bool((eng.logits() == logits).all())
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps({"want": want, "got": got, "solo": solo, "same_logits": same, "exchanges": tp.exchanges}))
'''


def test_tensor_parallel_rank_exchange_over_rccl_world_1():
    """The distributed tensor-parallel driver with its RCCL exchange (in-place all_gather_into_tensor on the engine's stream, called
    by the engine at every seam) at world 1 — the world the box offers: 2 layers x 2 seams for the prefill and for each of the two
    sweeps of 8 steps; a world of one adds one slot, so the result is the un-sharded engine's bit for bit."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DD_ROOT=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", TP_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["got"] == res["want"] == res["solo"] and res["same_logits"]
    assert res["exchanges"] == 4 * (1 + 2 * 8)


@pytest.mark.parametrize("mode", ["replicas", "kshard"])
def test_bench_gpus_2_on_one_device(mode):
    """`python bench.py --gpus 2` with world = 2 on the ONE GPU a box has (DD_BENCH_SHARE_DEVICE=1: both ranks on cuda:0 over gloo; RCCL
    refuses two ranks on one device): bench.py spawns its ranks, each decodes its own image shard (replicas) or its half of the members
    (kshard, real engines exchanging ids + the winner's record every token), the barriers and the all_reduce(MAX) of the time run at
    world 2 and rank 0 alone prints the JSON line.  Not a scaling measurement — the line carries `shared_device`."""
    env = dict(os.environ, DD_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--n-new", "12", "--images-per-gpu", "16",
           "--single-images", "0", "--no-roofline", "--no-cpu-baseline", "--mode", mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 alone reports
    line = json.loads(lines[0])
    pg = line["process_group"]
    assert line["n_gpus"] == 2 and pg["ranks"] == 2 and pg["backend"] == "gloo" and pg["shared_device"] is True and pg["rccl_ranks"] == 0
    assert line["value"] > 0 and line["scaling"] == ("weak" if mode == "replicas" else "strong")
    if mode == "replicas":
        assert line["config"]["images_per_step_per_gpu"] == 16
        # whole job = both ranks' tokens over the slower rank's time
        assert abs(line["value"] - 2 * 16 * 12 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"] + 0.02
    else:
        x = line["kshard_exchange"]
        assert x["world"] == 2 and x["backend"] == "gloo" and x["tokens"] >= 10 and x["collectives_per_token"] == 2
    print(f"\n[bench.py --gpus 2 --mode {mode}, both ranks on one GPU over gloo] value {line['value']} tok/s, {line['ms_per_step']} ms per step")
