"""Tensor-parallel decode in the engine (csrc/dd_tp.hip, include/dropdec.h dd_lm_tp_*): the LM's matrices
sharded over `world` ranks, two exchanges per layer, everything else replicated.

* world = 1 through the tensor-parallel path == the un-sharded engine, bit for bit;
* world = 2 / 4 (all ranks linked in one process on the one GPU there is; MHA and GQA; d_ff that needs padding) against the fp32
  oracle: tokens, masks, votes exact, logits within 1e-3 (observed ~1e-6: fp32 reassociation at the seams), every rank holding the
  same bits;
* one process per rank (torch.distributed, two processes on the one GPU with gloo staging the exchange through the host): the
  registered exchange runs at every seam and the result equals the linked run of the same world size bit for bit;
* LLaVA-1.5-7B matrices (8 layers) over 8 linked ranks against the un-sharded engine on the same weights."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, FAMILY_NEXT, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, random_weights


@pytest.fixture(scope="module")
def E():
    from dropoutdecoding_amd import build
    build.build()
    from dropoutdecoding_amd import lm
    return lm


def _cfg(E, rc):
    return E.LMConfig(rc.vocab_size, rc.hidden_size, rc.intermediate_size, rc.num_layers, rc.num_heads, rc.num_kv_heads, rc.head_dim,
                      rc.rms_eps, rc.rope_theta)


def test_world_1_through_the_tp_path_is_the_unsharded_engine_bitwise(E):
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
    w = random_weights(rc, 11, 0.05)
    probs = [0.2, 0.4, 0.6, 0.8]
    emb = (torch.randn(40, 256, generator=torch.Generator().manual_seed(3)) * 0.8).cuda()
    eng = E.DropoutEngine(_cfg(E, rc), family=FAMILY_LLAVA, max_seq=128, max_visual=32, seed=9)
    eng.load_state_dict(w)
    eng.set_speculation("never")
    eng.prefill(emb, 3, 30)
    want = []
    for _ in range(10):
        eng.decode_step(probs)
        st = eng.last_step()
        want.append((st["drop"].copy(), st["member_argmax"].tolist(), st["winner"], eng.logits().copy(), eng.base_logits().copy()))
    tp = E.TensorParallelGroup(_cfg(E, rc), 1, family=FAMILY_LLAVA, max_seq=128, max_visual=32, seed=9)
    tp.load_state_dict(w)
    tp.prefill(emb, 3, 30)
    for s in range(10):
        tp.decode_step(probs)
        st = tp.last_step()
        np.testing.assert_array_equal(st["drop"], want[s][0])
        assert st["member_argmax"].tolist() == want[s][1] and st["winner"] == want[s][2]
        np.testing.assert_array_equal(tp.logits(), want[s][3])
        np.testing.assert_array_equal(tp.ranks[0].base_logits(), want[s][4])
    assert tp.tokens() == eng.tokens()
    np.testing.assert_array_equal(tp.ranks[0].kv_sums(), eng.kv_sums())
    tp.close()
    eng.close()


@pytest.mark.parametrize("family,world,heads,kv_heads,dff,K", [(FAMILY_LLAVA, 2, 4, 4, 1280, 8), (FAMILY_LLAVA, 4, 8, 8, 1280, 3),
                                                                (FAMILY_NEXT, 2, 8, 2, 1024, 4), (FAMILY_IBLIP, 2, 4, 4, 768, 8)])
def test_linked_ranks_against_the_oracle(E, family, world, heads, kv_heads, dff, K):
    d = heads * 128
    rc = RefCfg(512, d, dff, 2, heads, kv_heads, 128, 1e-5, 10000.0)
    w = random_weights(rc, 21, 0.04)
    L = 32 if family == FAMILY_IBLIP else 40
    s0 = 0 if family == FAMILY_IBLIP else 4
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8][:K]
    emb = torch.randn(L + 11, d, generator=torch.Generator().manual_seed(5)) * 0.8
    n_new = 13
    ref = RefDecoder(family, rc, w, probs, seed=77)
    want = ref.generate(emb, s0, L, n_new)
    tp = E.TensorParallelGroup(_cfg(E, rc), world, family=family, max_seq=192, max_visual=L, seed=77)
    assert tp.ranks[0].cfg.num_heads == heads // world and tp.ranks[0].cfg.intermediate_size % 256 == 0
    tp.load_state_dict(w)
    tp.prefill(emb.cuda(), s0, L)
    np.testing.assert_allclose(tp.ranks[0].vision_uncert_dict()["epis_uncert_per_token"][0], ref.epi.numpy(), rtol=5e-3, atol=1e-6)
    worst = 0.0
    for s in range(n_new - 1):
        tp.decode_step(probs)
        st, r = tp.last_step(), ref.records[s]
        np.testing.assert_array_equal(st["drop"], r.drop, err_msg=f"step {s}")
        assert st["member_argmax"].tolist() == r.member_argmax and st["winner"] == r.winner, f"step {s}"
        lg = tp.logits()
        worst = max(worst, float(np.abs(lg - r.logits).max() / np.abs(r.logits).max()))
        for e in tp.ranks[1:]:                                  # replicated state: every rank holds the same bits
            np.testing.assert_array_equal(e.logits(), lg)
            np.testing.assert_array_equal(e.last_step()["drop"], st["drop"])
    assert tp.tokens() == want and all(e.tokens() == want for e in tp.ranks)
    assert worst < 1e-3, worst
    print(f"\n[tensor-parallel, {world} linked ranks, {family}, {heads}/{kv_heads} heads, d_ff {dff} -> {tp.ranks[0].cfg.intermediate_size} per rank] "
          f"logits vs the fp32 oracle: {worst:.1e}")
    with pytest.raises(Exception, match="tensor-parallel shard"):
        tp.ranks[0].decode_step(probs)                      # a shard refuses the un-sharded entry points
    # stock greedy (K = 0) and EOS through the same path
    tp.manual_seed(77)
    tp.prefill(emb.cuda(), s0, L)
    g = tp.generate(6, mprobs=probs, dropout=False)
    assert g == RefDecoder(family, rc, w, [], dropout=False).generate(emb, s0, L, 6)
    tp.manual_seed(77)
    tp.prefill(emb.cuda(), s0, L)
    eos = want[5]
    got = tp.generate(n_new, mprobs=probs, eos=eos)
    assert got == want[:want.index(eos) + 1]
    tp.close()


WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DD_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from dropoutdecoding_amd import lm
from dropoutdecoding_amd.dist import TensorParallelRank
from oracle.lm_ref import LMConfig as RefCfg, random_weights
rc = RefCfg(512, 512, 1280, 2, 4, 4, 128, 1e-5, 10000.0)
w = random_weights(rc, 21, 0.04)
cfg = lm.LMConfig(512, 512, 1280, 2, 4, 4, 128, 1e-5, 10000.0)
probs = [0.1, 0.3, 0.5, 0.7]
emb = (torch.randn(51, 512, generator=torch.Generator().manual_seed(5)) * 0.8).cuda()
tp = TensorParallelRank(cfg, rank, world, family=lm.FAMILY_LLAVA, max_seq=192, max_visual=40, seed=77)
tp.load_state_dict(w)
tp.prefill(emb, 4, 40)
toks = tp.generate(9, probs)
out = {"rank": rank, "tokens": toks, "exchanges": tp.exchanges, "logits": tp.engine.logits().tobytes().hex()}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_one_process_per_rank_equals_the_linked_run(E):
    """Two rank processes (both on the one GPU; gloo, the exchange staged through the host): the engine calls the registered
    all-gather at every seam — 2 per layer and sweep, prefill included — and ends on the linked run's bits."""
    from dropoutdecoding_amd import build
    build.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), DD_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, so[-2000:] + se[-4000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    rc = RefCfg(512, 512, 1280, 2, 4, 4, 128, 1e-5, 10000.0)
    w = random_weights(rc, 21, 0.04)
    probs = [0.1, 0.3, 0.5, 0.7]
    emb = torch.randn(51, 512, generator=torch.Generator().manual_seed(5)) * 0.8
    tp = E.TensorParallelGroup(_cfg(E, rc), 2, family=FAMILY_LLAVA, max_seq=192, max_visual=40, seed=77)
    tp.load_state_dict(w)
    tp.prefill(emb.cuda(), 4, 40)
    linked = tp.generate(9, probs)
    assert linked == RefDecoder(FAMILY_LLAVA, rc, w, probs, seed=77).generate(emb, 4, 40, 9)
    for o in outs:
        assert o["tokens"] == linked
        assert o["logits"] == tp.logits().tobytes().hex()              # the same reduction order: the same bits
        # 2 layers x 2 seams: once for the prefill, and for the un-masked + member sweep of each of the 8 steps
        assert o["exchanges"] == 4 * (1 + 2 * 8)
    tp.close()


def test_llava7b_matrices_over_8_linked_ranks(E):
    """LLaVA-1.5-7B's matrices (8 of the 32 layers, random bf16-valued weights) over 8 ranks — 4 heads and 1376 -> 1536 d_ff columns
    each — against the un-sharded engine on the same weights: same tokens and masks, logits within 1e-4; prints the step time of the
    8 ranks' kernels run back to back on the one GPU (what each of 8 GPUs would do concurrently, plus the exchanges)."""
    cfg = E.LMConfig(32064, 4096, 11008, 8, 32, 32, 128, 1e-5, 10000.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *sh: (torch.randn(*sh, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    sd = {"model.embed_tokens.weight": (torch.randn(32064, 4096, device="cuda", generator=g)).to(torch.bfloat16),
          "model.norm.weight": torch.ones(4096, device="cuda", dtype=torch.bfloat16), "lm_head.weight": rnd(32064, 4096)}
    for i in range(8):
        p = f"model.layers.{i}."
        sd[p + "input_layernorm.weight"] = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
        sd[p + "post_attention_layernorm.weight"] = torch.ones(4096, device="cuda", dtype=torch.bfloat16)
        for n_ in ("q", "k", "v", "o"):
            sd[p + f"self_attn.{n_}_proj.weight"] = rnd(4096, 4096)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = rnd(11008, 4096), rnd(11008, 4096), rnd(4096, 11008)
    probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    emb = (torch.randn(608, 4096, generator=torch.Generator().manual_seed(2)) * 0.5).cuda()
    eng = E.DropoutEngine(cfg, family=FAMILY_LLAVA, max_seq=704, max_visual=576, seed=24, kv_format="fp16")
    eng.load_state_dict(sd)
    eng.set_speculation("never")
    eng.prefill(emb, 5, 576)
    ref = []
    for _ in range(6):
        eng.decode_step(probs)
        ref.append((eng.last_step()["drop"].copy(), eng.logits().copy()))
    ref_toks = eng.tokens()
    eng.torch_stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.decode_step(probs)
    eng.torch_stream.synchronize()
    ms_full = (time.perf_counter() - t0) / 10 * 1e3
    eng.close()
    tp = E.TensorParallelGroup(cfg, 8, family=FAMILY_LLAVA, max_seq=704, max_visual=576, seed=24, kv_format="fp16")
    assert tp.ranks[0].cfg.num_heads == 4 and tp.ranks[0].cfg.intermediate_size == 1536
    tp.load_state_dict(sd)
    del sd
    tp.prefill(emb, 5, 576)
    worst = 0.0
    for s in range(6):
        tp.decode_step(probs)
        np.testing.assert_array_equal(tp.last_step()["drop"], ref[s][0])
        worst = max(worst, float(np.abs(tp.logits() - ref[s][1]).max() / np.abs(ref[s][1]).max()))
    assert tp.tokens() == ref_toks and worst < 1e-4, worst
    tp.ranks[0].torch_stream.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tp.decode_step(probs)
    tp.ranks[0].torch_stream.synchronize()
    ms_tp = (time.perf_counter() - t0) / 10 * 1e3
    print(f"\n[tensor-parallel, LLaVA-1.5-7B matrices x 8 layers, K=8, T~620] un-sharded two-sweep step {ms_full:.2f} ms; 8 linked ranks run back to back "
          f"on one GPU {ms_tp:.2f} ms = {ms_tp / 8:.2f} ms per rank (one hipGraph per step, no exchange cost); logits vs un-sharded {worst:.1e}")
    tp.close()
