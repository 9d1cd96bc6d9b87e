"""World-size-2 (and 3) gloo runs of the K-shard protocol on CPU.

The HIP engine cannot run here, so the phased engine API is implemented by a CPU stand-in built from the oracle;
what is under test is the host logic of dropoutdecoding_amd/dist.py: member partition, the two all-reduces,
zero-contribution broadcast of the winner record, identical caches/tokens on every rank.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dropoutdecoding_amd.dist import KShardDecoder, member_range, shard_images
from oracle import dropout_ref as DR
from oracle.decode_ref import FAMILY_LLAVA, RefDecoder
from oracle.lm_ref import LMConfig, lm_hidden, lm_logits, random_weights

CFG = LMConfig(128, 256, 256, 2, 2, 2, 128, 1e-5, 10000.0)
PROBS = [0.2, 0.4, 0.6, 0.8, 0.5]


def test_member_range_partitions():
    for K in range(1, 9):
        for world in (1, 2, 3, 4, 8):
            got = []
            for r in range(world):
                lo, hi = member_range(K, r, world)
                got += list(range(lo, hi))
            assert got == list(range(K)), (K, world)
    assert member_range(16, 1, 2) == (8, 16)
    assert shard_images(10, 1, 4) == [1, 5, 9]


class CpuPhasedEngine:
    """Same phase methods as DropoutEngine, computed with the oracle (CPU)."""

    def __init__(self, dec: RefDecoder, first_token: int):
        self.d, self.toks = dec, [first_token]

    def new_xchg_buffers(self):
        n = CFG.vocab_size + CFG.num_layers * 2 * CFG.kv_dim
        return torch.zeros(32, dtype=torch.int32), torch.zeros(n)

    def step_base(self, mprobs, uniforms=None):
        d = self.d
        self.K = len(mprobs)
        self.x = d.embed(self.toks[-1])
        self.T = d.cache.length
        self.orig = d.cache.clone()
        base = lm_logits(d.cfg, d.w, lm_hidden(d.cfg, d.w, self.x, torch.tensor([self.T]), d.cache.clone()))[0]
        keep = DR.overlap_keep(base, d.topk_ids)
        uni = torch.from_numpy(np.stack([d.rng.rand_f32(d.L) for _ in range(self.K)]))   # every rank draws all K
        self.drop = DR.sample_masks(d.epi, list(mprobs), keep, DR.MODE_LLAVA_CUMULATIVE, uni)
        self.logits, self.rows, self.ids_all = {}, {}, None
        return self.K

    def step_members(self, lo, hi):
        d = self.d
        for k in range(lo, hi):
            c = self.orig.clone()
            km = torch.ones(self.T + 1, dtype=torch.long)
            km[d.span_start:d.span_start + d.L][self.drop[k]] = 0
            self.logits[k] = lm_logits(d.cfg, d.w, lm_hidden(d.cfg, d.w, self.x, torch.tensor([self.T]), c, km))[0]
            self.rows[k] = torch.cat([torch.cat([c.k[i][:, -1].reshape(-1), c.v[i][:, -1].reshape(-1)]) for i in range(d.cfg.num_layers)])

    def export_ids(self, lo, hi, ids):
        ids.zero_()
        for k in range(lo, hi):
            ids[2 * k] = ids[2 * k + 1] = int(torch.argmax(self.logits[k]))

    def import_ids(self, ids):
        self.ids_all = [int(ids[2 * k + 1]) for k in range(self.K)]

    def export_winner(self, lo, hi, rec):
        self.win, _ = DR.vote(self.ids_all)
        rec.zero_()
        if lo <= self.win < hi:
            rec.copy_(torch.cat([self.logits[self.win], self.rows[self.win]]))

    def import_winner(self, rec):
        self.win_logits = rec[:CFG.vocab_size].clone()
        self.win_rows = rec[CFG.vocab_size:].clone()

    def step_commit(self):
        d, kv = self.d, CFG.kv_dim
        for i in range(d.cfg.num_layers):
            r = self.win_rows[i * 2 * kv:(i + 1) * 2 * kv]
            d.cache.k[i] = torch.cat([d.cache.k[i], r[:kv].reshape(CFG.num_kv_heads, 1, 128)], dim=1)
            d.cache.v[i] = torch.cat([d.cache.v[i], r[kv:].reshape(CFG.num_kv_heads, 1, 128)], dim=1)
        self.toks.append(int(torch.argmax(self.win_logits)))

    def tokens(self):
        return list(self.toks)


def _setup():
    torch.manual_seed(0)
    w = random_weights(CFG, 7, 0.06)
    emb = torch.randn(24, CFG.hidden_size, generator=torch.Generator().manual_seed(1))
    return w, emb


def _worker(rank, world, port, n_new, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    w, emb = _setup()
    dec = RefDecoder(FAMILY_LLAVA, CFG, w, PROBS, seed=99)
    first = dec.prefill(emb, 2, 16)
    ks = KShardDecoder(CpuPhasedEngine(dec, first), rank, world)
    toks = ks.generate(n_new, PROBS)
    ksum = float(sum(k.double().sum() for k in dec.cache.k))
    q.put((rank, toks, ksum))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_kshard_protocol_matches_single_process(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_new = 6
    w, emb = _setup()
    ref = RefDecoder(FAMILY_LLAVA, CFG, w, PROBS, seed=99)
    want = ref.generate(emb, 2, 16, n_new)
    want_ksum = float(sum(k.double().sum() for k in ref.cache.k))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_new, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, toks, ksum in out:
        assert toks == want, (rank, toks, want)
        assert abs(ksum - want_ksum) < 1e-6 * max(1.0, abs(want_ksum))      # caches identical on every rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ---- tensor-parallel plan: the sharded forward with two all-reduces per layer equals the unsharded one
def _tp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dropoutdecoding_amd.dist import TensorParallelPlan
        import torch.nn.functional as F
        from oracle.lm_ref import _rotate_half, rms_norm, rope_cos_sin
        cfg = LMConfig(128, 256, 512, 2, 4, 4, 64, 1e-5, 10000.0)      # 4 kv heads of 64: splits over 2 and 4 ranks
        w = random_weights(cfg, 3, 0.05)
        plan = TensorParallelPlan(cfg.num_heads, cfg.num_kv_heads, cfg.head_dim, cfg.hidden_size, cfg.intermediate_size, world)
        T = 12
        x = torch.randn(T, cfg.hidden_size, generator=torch.Generator().manual_seed(5))
        pos = torch.arange(T)
        cos, sin = rope_cos_sin(cfg, pos, torch.float32)
        k0, k1 = plan.kv_heads(rank)
        nh = (k1 - k0) * plan.G
        causal = torch.ones(T, T, dtype=torch.bool).tril()
        h = x
        for i in range(cfg.num_layers):
            sh = plan.shard_layer(w, f"model.layers.{i}.", rank)
            hn = rms_norm(h, sh["input_layernorm.weight"], cfg.rms_eps)
            qh = F.linear(hn, sh["self_attn.q_proj.weight"]).view(T, nh, cfg.head_dim).transpose(0, 1)
            kh = F.linear(hn, sh["self_attn.k_proj.weight"]).view(T, k1 - k0, cfg.head_dim).transpose(0, 1)
            vh = F.linear(hn, sh["self_attn.v_proj.weight"]).view(T, k1 - k0, cfg.head_dim).transpose(0, 1)
            qh = qh * cos[None] + _rotate_half(qh) * sin[None]
            kh = kh * cos[None] + _rotate_half(kh) * sin[None]
            att = (qh @ kh.repeat_interleave(plan.G, 0).transpose(1, 2)) * cfg.head_dim ** -0.5
            att = torch.softmax(att.masked_fill(~causal, torch.finfo(torch.float32).min), -1)
            o = (att @ vh.repeat_interleave(plan.G, 0)).transpose(0, 1).reshape(T, nh * cfg.head_dim)
            part = F.linear(o, sh["self_attn.o_proj.weight"])                 # row-parallel: partial sums of [T, d]
            dist.all_reduce(part)                                             # all-reduce #1
            h = h + part
            hn = rms_norm(h, sh["post_attention_layernorm.weight"], cfg.rms_eps)
            part = F.linear(F.silu(F.linear(hn, sh["mlp.gate_proj.weight"])) * F.linear(hn, sh["mlp.up_proj.weight"]), sh["mlp.down_proj.weight"])
            dist.all_reduce(part)                                             # all-reduce #2
            h = h + part
        hid = rms_norm(h, w["model.norm.weight"], cfg.rms_eps)
        from oracle.lm_ref import KVCache
        ref = lm_hidden(cfg, w, x, pos, KVCache())
        err = float((hid - ref).abs().max() / ref.abs().max())
        v0, v1 = rank * cfg.vocab_size // world, (rank + 1) * cfg.vocab_size // world
        loc = F.linear(hid[-1:], w["lm_head.weight"][v0:v1])[0]               # column-parallel lm_head: local (value, index) best
        best = torch.tensor([float(loc.max()), float(v0 + int(loc.argmax()))], dtype=torch.float64)
        allb = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allb, best)
        tok = int(max(allb, key=lambda t: (float(t[0]), -float(t[1])))[1])    # larger value first, then the lower index
        want = int(torch.argmax(lm_logits(cfg, w, ref[-1:])[0]))
        c = plan.collectives_per_sweep(cfg.num_layers, T)
        q.put((rank, err, tok == want, c["all_reduce"], plan.weight_bytes_per_rank(cfg.num_layers, cfg.vocab_size)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tensor_parallel_plan_equals_the_unsharded_forward(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, err, tok_ok, n_ar, wbytes in res:
        assert err < 1e-5 and tok_ok and n_ar == 4
    assert len({r[4] for r in res}) == 1                      # every rank streams the same share of the weights


def _tp_flag_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dropoutdecoding_amd import dist as dd
    assert dd.tp_peer_failed(None) == ""
    dist.barrier()
    if rank == 1:
        # a rank whose exchange raised: what TensorParallelRank._abort_peers does, on an object without an engine
        obj = dd.TensorParallelRank.__new__(dd.TensorParallelRank)
        obj.group, obj.rank, obj.error, obj.generation = None, rank, RuntimeError("exchange broke"), 3
        obj._abort_peers()
        q.put(("aborter", repr(obj.abort_error)))          # gloo has no abort: nothing raised, nothing swallowed
    else:
        import time
        t0 = time.time()
        msg = ""
        while not msg and time.time() - t0 < 20:
            msg = dd.tp_peer_failed(None, 3)
            time.sleep(0.05)
        # the flag belongs to ONE job — another job on the same group (generation 4), or one on a sub-group sharing the
        # default store, does not see it; a rank polls once per step (rate-limited), not once per exchange; teardown removes the key
        other_job, sub_group_key = dd.tp_peer_failed(None, 4), dd.tp_error_key(dist.new_group([0]) if False else None, 3)
        obj = dd.TensorParallelRank.__new__(dd.TensorParallelRank)
        obj.group, obj.rank, obj.generation, obj._last_poll = None, rank, 3, 0.0
        raised = ""
        try:
            obj._poll_peers()
        except RuntimeError as ex:
            raised = str(ex)
        obj.generation = 4
        obj._poll_peers(force=True)                        # a healthy job on the same group: nothing raised
        cleared = dd.tp_clear_error(None, 3)
        q.put(("peer", (msg, other_job, sub_group_key, raised, cleared, dd.tp_peer_failed(None, 3))))
    dist.barrier()
    dist.destroy_process_group()


def test_tp_failed_exchange_is_flagged_to_the_peers_over_gloo():
    """`_abort_peers` was untested and swallowed everything.  On gloo (no communicator abort) a failing rank flags the job in
    the rendezvous store and its peers see the flag before they enter their next exchange."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_tp_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got["aborter"] == "None"
    msg, other_job, key, raised, cleared, after = got["peer"]
    assert "rank 1" in msg and "exchange broke" in msg
    assert other_job == "" and key.endswith("/0-1/3")
    assert "a peer's exchange failed" in raised and "exchange broke" in raised
    assert cleared and after == ""
