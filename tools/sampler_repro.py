"""Unit-level reproducers and probes of the two co-residency faults.

  sampler(form, beside, rounds):  the group step's mask sampler (8 sequences, L = 576, K = 8, each drawing from its own mt19937 stream) launched
      back to back on its own stream — alone, or BESIDE 72-row slice-resident GEMVs (gate/up and qkv of LLaVA-1.5-7B) looping on two other streams
      — and every launch's masks compared with the oracle sampler over the host mt19937.  form: "wave" = the product's one-wave kernel (round 5),
      "block" / False = round 4's 1,024-thread kernel, "scratch" / True = round 3's (616 bytes of private scratch per lane); the block forms live
      in libdropdec_tools.so only (csrc/dd_sampler_block.h).
  sampler_streams(n_streams, rounds, lib, company_lanes, ...):  THE reproducer of the sampler fault: the sampler from one or more (high-priority)
      streams beside a group of `company_lanes` sequences taking rider steps (StepCompany) — wrong masks about once in 50,000 launches with the
      1,024-thread forms requesting the LDS they use (lds_kib = 0), none with the one-wave form.  n_seq / lds_kib / company_mode / dbg: round 5's
      experiments (workgroups per launch, the LDS request sweep, the company cut down, the checking form that dumps the wrong words;
      tools/r05_sampler_fault.py runs them one process per leg: a process that builds several StepCompanies once hung).
  analyse_dump(d, ...):  where the wrong words of a dump come from (which stream, which generation, zeros, the launch-start state, elsewhere in LDS).
  twist_probe / barrier_probe:  only the regeneration sweeps / only store - barrier - read-the-other-waves - barrier, in workgroups of the block
      sampler's shape, beside the same company: both clean.
  probe / lds_probe / lds_full_probe / hold_probe / pk_probe:  round 4's probes (private scratch, LDS exchange, LDS isolation, registers and loads in
      flight, packed FP32 chains and the P.V step).

    python tools/sampler_repro.py [rounds]      # prints one JSON line per configuration
"""
import json, os, sys, threading, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.dropout import TorchCpuCompatRNG
from oracle import dropout_ref as DR
from oracle.mt19937 import TorchCpuMT19937

N_SEQ, L_VIS, K_TOP, STEPS = 8, 576, 5, 24
PROBS = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]


class Company:
    """72-row slice-resident GEMVs looping on two streams of their own (host threads: the timing hook synchronises)."""

    def __init__(self, lib):
        cfg = lm.LMConfig(32064, 4096, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
        self.eng = lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=128, max_visual=32, kv_format="fp16", lib=lib)
        self.eng.load_synthetic(0, 0.02)
        # (the timing hook records into events of the handle it is given: one handle per thread, over the same weights)
        self.engs = [self.eng, lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=128, max_visual=32, kv_format="fp16", lib=lib, share_weights_with=self.eng)]
        self.lib, self.stop, self.threads, self.launches = lib, False, [], [0, 0]
        self.rows = 72
        self.streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def _loop(self, i, which):
        ms, by = C.c_float(), C.c_double()
        while not self.stop:
            rc = self.lib.dd_lm_time_gemv(self.engs[i]._h, which, self.rows, 64, C.byref(ms), C.byref(by), self.streams[i].cuda_stream)
            assert rc == 0, self.lib.dd_last_error()
            self.launches[i] += 64 + 2

    def __enter__(self):
        self.threads = [threading.Thread(target=self._loop, args=(0, 2)), threading.Thread(target=self._loop, args=(1, 0))]
        for t in self.threads:
            t.start()
        time.sleep(0.05)
        return self

    def __exit__(self, *a):
        self.stop = True
        for t in self.threads:
            t.join()


def _inputs():
    rs = np.random.RandomState(7)
    seqs = []
    for i in range(N_SEQ):
        epi = (rs.rand(L_VIS) * 2).astype(np.float32)
        topk = rs.randint(0, 400, (L_VIS, K_TOP)).astype(np.int32)
        tok = int(topk[3 + i, 2])
        keep = (topk == tok).any(1)
        ref = TorchCpuMT19937(100 + i)
        want, unis = [], []
        for _ in range(STEPS):
            uni = torch.from_numpy(np.stack([ref.rand_f32(L_VIS) for _ in PROBS]))
            unis.append(uni.numpy())
            want.append(DR.sample_masks(torch.from_numpy(epi), PROBS, torch.from_numpy(keep), DR.MODE_LLAVA_CUMULATIVE, uni).numpy())
        seqs.append({"epi": epi, "topk": topk, "tok": tok, "keep": keep, "want": np.stack(want), "uniforms": np.stack(unis)})
    return seqs


def sampler(scratch: bool, beside: bool, rounds: int, lib=None, seqs=None, rows: int = 72) -> dict:
    lib = lib or _lib.load_tools()
    # scratch: True / "scratch" = round 3's form (private scratch), False / "block" = round 4's 1,024-thread kernel, "wave" = the product's one-wave sampler
    form = scratch if isinstance(scratch, str) else ("scratch" if scratch else "block")
    lib.dd_tools_set_tuning(34, {"wave": 0, "scratch": 1, "block": 3}[form])
    seqs = seqs or _inputs()
    K = len(PROBS)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    epi = [dev(s["epi"]) for s in seqs]
    topk = [dev(s["topk"]) for s in seqs]
    argmax = [torch.tensor([s["tok"]], dtype=torch.int32, device="cuda") for s in seqs]
    keep = [torch.zeros(L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
    drop = [torch.zeros(STEPS, K, L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
    n_drop = [torch.zeros(STEPS, K, dtype=torch.int32, device="cuda") for _ in seqs]
    bits = [torch.zeros(STEPS, L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
    rngs = [TorchCpuCompatRNG(100 + i, lib=lib) for i in range(N_SEQ)]
    want = [s["want"] for s in seqs]
    want_bits = [np.bitwise_or.reduce(w.astype(np.uint8) << np.arange(K, dtype=np.uint8)[None, :, None], axis=1) for w in want]
    st = torch.cuda.Stream()
    arr = lambda ts: (C.c_void_p * N_SEQ)(*[t.data_ptr() if torch.is_tensor(t) else t for t in ts])
    Ls = (C.c_int32 * N_SEQ)(*([L_VIS] * N_SEQ))
    pr = (C.c_double * K)(*PROBS)
    bad_launches, launches, first_bad = 0, 0, None
    t0 = time.time()

    def body():
        nonlocal bad_launches, launches, first_bad
        for r in range(rounds):
            torch.cuda.synchronize()
            for i, g in enumerate(rngs):
                g.manual_seed(100 + i)
            torch.cuda.synchronize()
            for s in range(STEPS):
                rc = lib.dd_tools_sample_masks_lanes(N_SEQ, arr(epi), Ls, arr(keep), arr(argmax), arr(topk), arr([g.handle.value for g in rngs]),
                                                     arr([d[s].data_ptr() for d in drop]), arr([n[s].data_ptr() for n in n_drop]),
                                                     arr([b[s].data_ptr() for b in bits]), K_TOP, pr, K, DR.MODE_LLAVA_CUMULATIVE, st.cuda_stream)
                assert rc == 0, lib.dd_last_error()
            st.synchronize()
            launches += STEPS
            for i in range(N_SEQ):
                got = drop[i].cpu().numpy().astype(bool)
                gb = bits[i].cpu().numpy()
                gn = n_drop[i].cpu().numpy()
                for s in range(STEPS):
                    if not (np.array_equal(got[s], want[i][s]) and np.array_equal(gb[s], want_bits[i][s]) and np.array_equal(gn[s], want[i][s].sum(1))):
                        bad_launches += 1
                        if first_bad is None:
                            first_bad = {"round": r, "sequence": i, "launch": s, "wrong_mask_bytes": int((got[s] != want[i][s]).sum())}
                        break

    if beside:
        co = Company(lib)
        co.rows = rows            # 72 / 64 rows: 144 / 128 KiB of LDS per GEMV workgroup — the sampler's 76-KiB workgroups cannot share their CUs; 32 / 16 rows can
        with co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    lib.dd_tools_set_tuning(34, 0)
    return {"test": "sampler", "sampler_form": form, "beside_72_row_gemvs": beside, "company_rows": rows if beside else 0, "sampler_launches": launches, "workgroups_per_launch": N_SEQ,
            "company_gemv_launches": company, "sequences_with_a_wrong_launch": bad_launches, "first_bad": first_bad, "seconds": round(time.time() - t0, 1)}


class StepCompany:
    """A group of `lanes` sequences (LLaVA-1.5-7B matrices, 2 layers) taking rider steps in a loop on the engines' own stream: the company a
    branch-local sampler has inside a group step — attention tile passes (MFMA, a few KiB of LDS, 95 VGPRs), combine and finishing kernels, which
    fit on a CU beside the sampler's 76-KiB workgroup where the 144-KiB slice GEMVs do not."""

    def __init__(self, lib, lanes=32, mode="full"):
        """mode: "full" = prefills + 120 rider steps in turn (round 4's company); "steps" = rider steps only beside the test (the prompts are
        prefilled again every 120 steps under self.lock, which the test's workers hold around their launches); "prefill" = batched prefills only."""
        cfg = lm.LMConfig(32064, 4096, 11008, 2, 32, 32, 128, 1e-5, 10000.0)
        self.mode = mode
        self.engs = []
        for i in range(lanes):
            self.engs.append(lm.DropoutEngine(cfg, family=lm.FAMILY_LLAVA, max_seq=608 + 140, max_visual=576, seed=7 + i, kv_format="fp16", lib=lib,
                                              share_weights_with=self.engs[0] if self.engs else None))
        self.engs[0].load_synthetic(0, 0.02)
        g = torch.Generator().manual_seed(3)
        self.embs = [(torch.randn(608, 4096, generator=g) * 0.5).cuda() for _ in range(lanes)]
        self.stop, self.steps, self.thread, self.prefills = False, 0, None, 0
        self.lock = threading.Lock()

    def _prefill(self):
        lm.prefill_group(self.engs[:16], self.embs[:16], [(5, 576)] * 16)
        if len(self.engs) > 16:
            lm.prefill_group(self.engs[16:32], self.embs[16:32], [(5, 576)] * (len(self.engs[16:32])))

    def _loop(self):
        torch.cuda.set_device(0)
        while not self.stop:
            if self.mode == "steps":                 # the prompts are prefilled while no sampler launch of the test is in flight
                with self.lock:
                    self._prefill()
                    torch.cuda.synchronize()
            else:
                self._prefill()
            self.prefills += 1
            if self.mode == "prefill":
                torch.cuda.synchronize()
                continue
            grp = lm.EngineGroup(self.engs)
            for _ in range(120):
                if self.stop:
                    break
                grp.decode_step(PROBS)
                self.steps += 1
            torch.cuda.synchronize()

    def __enter__(self):
        self.thread = threading.Thread(target=self._loop)
        self.thread.start()
        time.sleep(2.0)
        return self

    def __exit__(self, *a):
        self.stop = True
        self.thread.join()
        for e in reversed(self.engs):
            e.close()


_GEN_CACHE = {}


def _generations(seed: int, n: int):
    """[n][624] uint32: the first n regenerated blocks of the mt19937 stream seeded with `seed` (oracle/mt19937.py)."""
    from oracle import mt19937 as MT
    if (seed, n) not in _GEN_CACHE:
        st = MT.seed_state(seed)
        out = []
        for _ in range(n):
            MT._twist(st[:MT.N])
            out.append(st[:MT.N].copy())
        _GEN_CACHE[(seed, n)] = np.stack(out)
    return _GEN_CACHE[(seed, n)]


def analyse_dump(d: np.ndarray, test_seeds, company_seeds=()) -> dict:
    """Provenance of the words the checking sampler found wrong (dd_dropout.hip mt_dbg_check; layout in include/dropdec_tools.h)."""
    d = d.astype(np.uint32)
    hdr = {"failed_checks": int(d[1]), "workgroup": int(d[2]), "member": int(np.int32(d[3])), "regenerations_before": int(d[4]),
           "check_site": {1: "before a regeneration", 3: "member start", 4: "after the member's uniforms", 5: "member end"}.get(int(d[5]), int(d[5])),
           "launch_tag": int(d[6]), "read_index": int(d[7]), "checks_before": int(d[8]), "hw_id": hex(int(d[10])), "xcc_id": int(d[11]) & 0xF,
           "lds_alloc": hex(int(d[12]))}
    shadow, seen, later, src = (d[16 + i * 640:16 + i * 640 + 624] for i in range(4))
    nw = int(d[9])
    lds = d[16 + 5 * 640:16 + 5 * 640 + nw]
    W = np.nonzero(shadow != seen)[0]
    runs, start = [], None
    for i in W.tolist() + [None]:
        if start is None:
            start = prev = i
        elif i is None or i != prev + 1:
            runs.append([int(start), int(prev)])
            start = prev = i
        else:
            prev = i
    out = dict(hdr)
    out["wrong_words"] = int(len(W))
    out["wrong_word_runs"] = runs[:8]
    out["still_wrong_microseconds_later"] = int((later[W] != shadow[W]).sum())
    out["later_read_equals_first_read"] = bool(np.array_equal(later[W], seen[W]))
    if not len(W):
        return out
    # which stream / generation is the block the registers hold?
    n_gen = 200
    owner = None
    for kind, seeds in (("test", test_seeds), ("company", company_seeds)):
        for sd in seeds:
            G = _generations(sd, n_gen)
            hit = np.nonzero((G == shadow[None, :]).sum(1) > 500)[0]
            if len(hit):
                owner = (kind, sd, int(hit[0]))
                break
        if owner:
            break
    out["block_in_registers_is"] = None if owner is None else {"stream": owner[0], "seed": owner[1], "generation": owner[2]}
    # provenance of the wrong values
    prov = {"zero": int((seen[W] == 0).sum()), "launch_start_state_same_index": int((seen[W] == src[W]).sum())}
    if owner:
        G = _generations(owner[1], n_gen)
        g = owner[2]
        for dg in (-3, -2, -1, 1, 2):
            if 0 <= g + dg < n_gen:
                prov[f"own_generation_{dg:+d}_same_index"] = int((seen[W] == G[g + dg][W]).sum())
        seedwords = __import__("oracle.mt19937", fromlist=["x"]).seed_state(owner[1])[:624]
        prov["own_seed_state_same_index"] = int((seen[W] == seedwords[W]).sum())
    table = {}
    for kind, seeds in (("test", test_seeds), ("company", company_seeds)):
        for sd in seeds:
            G = _generations(sd, n_gen)
            for v in np.unique(seen[W]):
                gi, wi = np.nonzero(G == v)
                if len(gi):
                    table.setdefault(f"{kind}:{sd}", []).append([int(v), int(gi[0]), int(wi[0])])
    prov["values_found_in_any_generation_of"] = {k: {"count": len(v), "first": v[:4]} for k, v in table.items()}
    # the workgroup's own LDS: do the wrong values, or the right ones, sit anywhere else in it?
    mt_off = (8192 * 4 * 2 + 8192) // 4
    other = np.concatenate([lds[:mt_off], lds[mt_off + 624:]])
    prov["wrong_values_elsewhere_in_own_lds"] = int(np.isin(seen[W], other).sum())
    prov["right_values_elsewhere_in_own_lds"] = int(np.isin(shadow[W], other).sum())
    u_bits = lds[8192:8192 + 8192]
    prov["wrong_values_equal_uniform_buffer_words"] = int(np.isin(seen[W], u_bits).sum())
    # are the wrong values the tempered / untempered neighbours of something?  crude tests
    prov["wrong_equals_right_shifted_by_words"] = {str(k): int((seen[W] == np.roll(shadow, k)[W]).sum()) for k in (-64, -1, 1, 64)}
    out["provenance_of_wrong_values"] = prov
    out["sample"] = [{"word": int(i), "want": hex(int(shadow[i])), "got": hex(int(seen[i])), "later": hex(int(later[i])), "launch_start": hex(int(src[i]))} for i in W[:6]]
    return out


def sampler_streams(n_streams: int, rounds: int, lib=None, company_lanes: int = 0, sync_each: bool = False, n_seq: int = N_SEQ, lds_kib: int = -1,
                    dbg: bool = False, company_mode: str = "full", dump_prefix: str = "", max_events: int = 4, seconds: float = 0.0,
                    form: str = "block") -> dict:
    """The lanes sampler from `n_streams` host threads at once, each on a stream of its own with sequences (and rng streams) of its own — the
    branch-local form's launch pattern —, optionally beside a group of `company_lanes` sequences taking rider steps; every launch against the
    oracle.  n_seq: workgroups (sequences) per launch; lds_kib: the kernel's dynamic-LDS request (tools key 48: 0 = the 76 KiB it uses, 1 = the
    product's 156 KiB, n = n KiB; -1 = leave as set); dbg: the checking sampler (tools key 34 = 2) — the generator block mirrored in registers and
    verified, the first failure of every round dumped and analysed (analyse_dump); seconds > 0: stop after that long even if rounds remain."""
    lib = lib or _lib.load_tools()
    # form: "wave" = the product's one-wave sampler (round 5); "block" = round 4's 1,024-thread kernel; "scratch" = round 3's; dbg: the checking block form
    lib.dd_tools_set_tuning(34, 2 if dbg else {"wave": 0, "scratch": 1, "block": 3}[form])
    if lds_kib >= 0:
        lib.dd_tools_set_tuning(48, lds_kib)
    seqs = _inputs()[:n_seq]
    K = len(PROBS)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    want = [s_["want"] for s_ in seqs]
    want_bits = [np.bitwise_or.reduce(w.astype(np.uint8) << np.arange(K, dtype=np.uint8)[None, :, None], axis=1) for w in want]
    bad, launches, first_bad = [0] * n_streams, [0] * n_streams, [None] * n_streams
    events, dumps = [], []
    t0 = time.time()
    dbg_buf = None
    if dbg:
        assert n_streams == 1, "the checking sampler has one dump buffer"
        dbg_buf = torch.zeros(int(lib.dd_tools_sampler_dbg_words()), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        lib.dd_tools_sampler_dbg_attach(dbg_buf.data_ptr())
    co_ref = [None]

    def worker(t):
        torch.cuda.set_device(0)
        epi = [dev(s_["epi"]) for s_ in seqs]
        topk = [dev(s_["topk"]) for s_ in seqs]
        argmax = [torch.tensor([s_["tok"]], dtype=torch.int32, device="cuda") for s_ in seqs]
        keep = [torch.zeros(L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
        drop = [torch.zeros(STEPS, K, L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
        n_drop = [torch.zeros(STEPS, K, dtype=torch.int32, device="cuda") for _ in seqs]
        bits = [torch.zeros(STEPS, L_VIS, dtype=torch.uint8, device="cuda") for _ in seqs]
        rngs = [TorchCpuCompatRNG(100 + i, lib=lib) for i in range(n_seq)]
        # a high-priority stream: HIP gives those hardware queues of their own, so the sampler's launches run BESIDE the company's kernels whatever
        # queue the company's streams were dealt (with many streams alive in the process a normal-priority stream can land on the company's queue,
        # and the launches then run between its kernels instead of beside them: 3,000 instead of 200,000 launches in 40 s, and no co-residency)
        st = torch.cuda.Stream(priority=-1)
        arr = lambda ts: (C.c_void_p * n_seq)(*[x.data_ptr() if torch.is_tensor(x) else x for x in ts])
        Ls = (C.c_int32 * n_seq)(*([L_VIS] * n_seq))
        pr = (C.c_double * K)(*PROBS)
        import contextlib
        for r in range(rounds):
            if seconds and time.time() - t0 > seconds:
                break
            lock = co_ref[0].lock if co_ref[0] is not None and co_ref[0].mode == "steps" else contextlib.nullcontext()
            with lock:
                for i, g in enumerate(rngs):
                    g.manual_seed(100 + i)
                torch.cuda.current_stream().synchronize()
                for s_ in range(STEPS):
                    rc = lib.dd_tools_sample_masks_lanes(n_seq, arr(epi), Ls, arr(keep), arr(argmax), arr(topk), arr([g.handle.value for g in rngs]),
                                                         arr([d[s_].data_ptr() for d in drop]), arr([n[s_].data_ptr() for n in n_drop]),
                                                         arr([b[s_].data_ptr() for b in bits]), K_TOP, pr, K, DR.MODE_LLAVA_CUMULATIVE, st.cuda_stream)
                    assert rc == 0, lib.dd_last_error()
                    if sync_each:
                        st.synchronize()          # (experiment) no two sampler launches back to back: the host waits for each
                st.synchronize()
            launches[t] += STEPS
            with torch.cuda.stream(st):
                got_all = [d.cpu().numpy().astype(bool) for d in drop]
                gb_all = [b.cpu().numpy() for b in bits]
                gn_all = [n.cpu().numpy() for n in n_drop]
                if dbg and int(dbg_buf[0].item()) != 0:
                    d = dbg_buf.cpu().numpy().copy()
                    dbg_buf.zero_()
                    st.synchronize()
                    if len(dumps) < max_events:
                        a = analyse_dump(d, [100 + i for i in range(n_seq)], [7 + i for i in range(company_lanes)])
                        a["round"] = r
                        dumps.append(a)
                        if dump_prefix:
                            np.save(f"{dump_prefix}_{len(dumps)}.npy", d[:16 + 5 * 640 + int(d[9])])
            for i in range(n_seq):
                for s_ in range(STEPS):
                    if not (np.array_equal(got_all[i][s_], want[i][s_]) and np.array_equal(gb_all[i][s_], want_bits[i][s_]) and
                            np.array_equal(gn_all[i][s_], want[i][s_].sum(1))):
                        bad[t] += 1
                        k_bad = [int(k) for k in range(K) if not np.array_equal(got_all[i][s_][k], want[i][s_][k])]
                        if k_bad and len(events) < 12:
                            k0 = k_bad[0]
                            pos = np.nonzero(got_all[i][s_][k0] != want[i][s_][k0])[0]
                            events.append({"stream": t, "round": r, "sequence": i, "launch": s_, "first_wrong_member": k0, "wrong_positions": pos[:48].tolist(),
                                           "n_wrong_positions": int(len(pos)), "waves_of_wrong_positions": sorted({int(x) // 64 for x in pos})})
                        if first_bad[t] is None:
                            first_bad[t] = {"stream": t, "round": r, "sequence": i, "launch": s_, "members_with_wrong_masks": k_bad,
                                            "wrong_mask_bytes": int((got_all[i][s_] != want[i][s_]).sum()),
                                            "n_drop_got": gn_all[i][s_].tolist(), "n_drop_want": want[i][s_].sum(1).tolist()}
                        break

    def run():
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_streams)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()

    steps = prefills = 0
    if company_lanes:
        with StepCompany(lib, company_lanes, company_mode) as co:
            co_ref[0] = co
            run()
            steps, prefills = co.steps, co.prefills
    else:
        run()
    lib.dd_tools_set_tuning(34, 0)
    if dbg:
        lib.dd_tools_sampler_dbg_attach(None)
    return {"test": "sampler_streams", "sampler_form": "checking block form" if dbg else form, "streams": n_streams, "host_waits_for_every_launch": sync_each, "sampler_launches": sum(launches), "workgroups_per_launch": n_seq,
            "lds_request_key_48": lds_kib, "checking_sampler": dbg, "company_mode": company_mode if company_lanes else None,
            "company_rider_steps": steps, "company_prefill_rounds": prefills, "company_lanes": company_lanes, "sequences_with_a_wrong_launch": sum(bad),
            "first_bad": [f for f in first_bad if f], "events": events, "generator_block_dumps": dumps, "seconds": round(time.time() - t0, 1)}


def twist_probe(rounds: int, lib=None, company_lanes: int = 32, lds_bytes: int = 77856, wgs: int = 8, iters: int = 64) -> dict:
    """dd_tools_twist_probe (csrc/dd_tools.hip): only the mt19937 regeneration of the sampler, in workgroups of the sampler's shape, checked in the
    kernel against an in-order recomputation — on a stream of its own, alone (company_lanes = 0) or beside a group taking rider steps.  Written at
    the end of round 4 and not yet run on a GPU."""
    lib = lib or _lib.load_tools()
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    t0 = time.time()
    launches = 0

    def body():
        nonlocal launches
        for r in range(rounds):
            rc = lib.dd_tools_twist_probe(16, wgs, iters, lds_bytes, out.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()
            launches += 16

    steps = 0
    if company_lanes:
        with StepCompany(lib, company_lanes) as co:
            body()
            steps = co.steps
    else:
        body()
    o = out.cpu().numpy().astype(np.uint32)
    return {"test": "mt19937_regeneration_probe", "probe_launches": launches, "workgroups_per_launch": wgs, "regenerations_per_workgroup": iters,
            "lds_bytes": lds_bytes, "company_lanes": company_lanes, "company_rider_steps": steps, "differing_words": int(o[0]),
            "first": None if o[0] == 0 else {"workgroup": int(o[1]), "iteration": int(o[2]), "word": int(o[3]), "got": int(o[4]), "want": int(o[5]),
                                              "word_before": int(o[6]), "word_after": int(o[7])},
            "seconds": round(time.time() - t0, 1)}


def barrier_probe(rounds: int, lib=None, company_lanes: int = 32, threads: int = 1024, lds_bytes: int = 77856, wgs: int = 8, iters: int = 600,
                  seconds: float = 0.0, company_mode: str = "full") -> dict:
    """dd_tools_barrier_probe: s_barrier + LDS visibility of a workgroup of the sampler's shape, alone or beside a group taking rider steps."""
    lib = lib or _lib.load_tools()
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    t0 = time.time()
    launches = 0

    def body():
        nonlocal launches
        for r in range(rounds):
            if seconds and time.time() - t0 > seconds:
                break
            rc = lib.dd_tools_barrier_probe(16, wgs, threads, iters, lds_bytes, out.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()
            launches += 16

    steps = 0
    if company_lanes:
        with StepCompany(lib, company_lanes, company_mode) as co:
            body()
            steps = co.steps
    else:
        body()
    o = out.cpu().numpy().astype(np.uint32)
    return {"test": "barrier_probe", "threads_per_workgroup": threads, "lds_bytes": lds_bytes, "probe_launches": launches, "workgroups_per_launch": wgs,
            "barrier_pairs_per_workgroup": iters, "company_lanes": company_lanes, "company_rider_steps": steps,
            "words_not_of_this_iteration": int(o[0]), "one_iteration_old": int(o[1]), "of_a_later_iteration": int(o[2]),
            "first": None if o[0] == 0 else {"workgroup": int(o[3]), "iteration": int(o[4]), "reader_wave": int(o[5]), "writer_wave": int(o[6]), "value": int(o[7])},
            "seconds": round(time.time() - t0, 1)}


def probe(beside: bool, rounds: int, lib=None, wgs: int = 8, spin: int = 200) -> dict:
    lib = lib or _lib.load_tools()
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for r in range(rounds):
            rc = lib.dd_tools_scratch_probe(32, wgs, spin, err.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        with Company(lib) as co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    return {"test": "scratch_probe", "beside_72_row_gemvs": beside, "probe_launches": rounds * 32, "workgroups_per_launch": wgs, "spin": spin,
            "company_gemv_launches": company, "mismatching_words": int(err[0].item()), "seconds": round(time.time() - t0, 1)}


def lds_probe(beside: bool, rounds: int, lib=None, wgs: int = 3072, which=(2, 0), rows: int = 64) -> dict:
    """The fp32-cache attention tile pass's LDS exchange pattern (dd_tools_lds_barrier_probe) alone / beside slice GEMVs of `rows` rows."""
    lib = lib or _lib.load_tools()
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for r in range(rounds):
            rc = lib.dd_tools_lds_barrier_probe(16, wgs, 4, err.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        co = Company(lib)
        co.rows = rows
        with co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    return {"test": "lds_barrier_probe", "beside_gemvs_of_rows": rows if beside else 0, "probe_launches": rounds * 16, "workgroups_per_launch": wgs,
            "company_gemv_launches": company, "mismatching_words": int(err[0].item()), "seconds": round(time.time() - t0, 1)}


def lds_full_probe(beside: bool, rounds: int, lib=None, rows: int = 64, lds_bytes: int = 30720) -> dict:
    """Workgroups that hold and verify a pattern in ALL of their dynamic LDS (k_lds_hold) alone / beside slice GEMVs of `rows` rows."""
    lib = lib or _lib.load_tools()
    err = torch.tensor([0, 0, -1, 0], dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for r in range(rounds):
            rc = lib.dd_tools_lds_overlap_probe(0, 0, 0, lds_bytes, 3072, 8, 16, err.data_ptr(), st.cuda_stream, st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        co = Company(lib)
        co.rows = rows
        with co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    e = [int(x) & 0xFFFFFFFF for x in err.tolist()]
    return {"test": "lds_full_pattern_probe", "lds_bytes": lds_bytes, "beside_gemvs_of_rows": rows if beside else 0, "probe_launches": rounds * 16,
            "workgroups_per_launch": 3072, "company_gemv_launches": company, "corrupted_words": e[1],
            "first_bad_byte": None if not e[1] else e[2] * 4, "last_bad_byte": None if not e[1] else e[3] * 4 + 3, "seconds": round(time.time() - t0, 1)}


def pk_probe(beside: bool, rounds: int, lib=None, rows: int = 64, pv=False) -> dict:
    """Packed vs scalar FP32 multiply-add chains (dd_tools_pk_probe; pv: the tile pass's own P.V step, dd_tools_pv_probe) alone / beside slice GEMVs."""
    lib = lib or _lib.load_tools()
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for r in range(rounds):
            rc = (lib.dd_tools_pkadd_gload_probe(16, 3072, 16, err.data_ptr(), st.cuda_stream) if pv == "gload" else
                  lib.dd_tools_pv_probe(16, 3072, 16, err.data_ptr(), st.cuda_stream) if pv
                  else lib.dd_tools_pk_probe(16, 3072, 512, err.data_ptr(), st.cuda_stream))
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        co = Company(lib)
        co.rows = rows
        with co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    return {"test": "packed_add_of_global_loads_probe" if pv == "gload" else ("pv_step_probe" if pv else "packed_fp32_probe"), "beside_gemvs_of_rows": rows if beside else 0, "probe_launches": rounds * 16, "workgroups_per_launch": 3072,
            "lanes_checked": rounds * 16 * 3072 * 256, "company_gemv_launches": company, "lanes_with_wrong_packed_result": int(err[0].item()),
            "seconds": round(time.time() - t0, 1)}


def hold_probe(kind: int, beside: bool, rounds: int, lib=None, rows: int = 64) -> dict:
    """dd_tools_hold_probe (0: registers, 1: outstanding global loads) alone / beside slice GEMVs of `rows` rows."""
    lib = lib or _lib.load_tools()
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    lib.dd_tools_hold_probe(kind, 1, 64, 1, err.data_ptr(), st.cuda_stream)       # (allocates / fills the load probe's buffer)
    torch.cuda.synchronize()
    err.zero_()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for r in range(rounds):
            rc = lib.dd_tools_hold_probe(kind, 16, 3072, 24, err.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        co = Company(lib)
        co.rows = rows
        with co:
            body()
            company = sum(co.launches)
    else:
        body()
        company = 0
    return {"test": ["vgpr_hold_probe", "global_load_hold_probe"][kind], "beside_gemvs_of_rows": rows if beside else 0, "probe_launches": rounds * 16,
            "workgroups_per_launch": 3072, "company_gemv_launches": company, "mismatching_words": int(err[0].item()), "seconds": round(time.time() - t0, 1)}


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    torch.cuda.set_device(0)
    lib = _lib.load_tools()
    seqs = _inputs()
    out = []
    for scratch in (False, True):
        for beside in (False, True):
            out.append(sampler(scratch, beside, rounds, lib, seqs))
            print(json.dumps(out[-1]), flush=True)
    for beside in (False, True):
        for wgs in (8, 256):
            out.append(probe(beside, rounds, lib, wgs=wgs))
            print(json.dumps(out[-1]), flush=True)
    for beside, rows in ((False, 0), (True, 64), (True, 32), (True, 16)):
        out.append(lds_probe(beside, rounds, lib, rows=rows or 64))
        print(json.dumps(out[-1]), flush=True)
    for pv in (True, "gload", False):
        for beside, rows in ((False, 0), (True, 64), (True, 32), (True, 16), (True, 72)):
            out.append(pk_probe(beside, rounds, lib, rows=rows or 64, pv=pv))
            print(json.dumps(out[-1]), flush=True)
    for beside, rows, nbytes in ((False, 0, 30720), (True, 64, 30720), (True, 32, 30720), (True, 16, 30720), (True, 64, 12288), (True, 32, 66560)):
        out.append(lds_full_probe(beside, rounds, lib, rows=rows or 64, lds_bytes=nbytes))
        print(json.dumps(out[-1]), flush=True)
    for kind in (0, 1):
        for beside, rows in ((False, 0), (True, 64), (True, 32)):
            out.append(hold_probe(kind, beside, rounds, lib, rows=rows or 64))
            print(json.dumps(out[-1]), flush=True)
    if os.environ.get("DD_REPRO_LOG"):
        with open(os.environ["DD_REPRO_LOG"], "w") as f:
            json.dump(out, f, indent=1)
