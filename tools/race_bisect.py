"""Where does a two-branch group step first differ from the one-branch step?  Per-stage checksums of every multi-group sweep
(dd_tools_sweep_trace: embed, q rows, new K rows, attention output, o_proj, gate/up, down) of the same decode run with 1 and with 2
branches; prints the first (step, sweep, layer, stage) whose sums differ, for a few repetitions of the two-branch run.

    python tools/race_bisect.py [kv=fp32] [lanes=12] [steps=3] ["key=value,..." applied to both runs]"""
import os, sys
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm

torch.cuda.set_device(0)
KV = sys.argv[1] if len(sys.argv) > 1 else "fp32"
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 12
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = _lib.load()
L.dd_tools_set_tuning(8, 0)           # eager launches: the trace numbers sweeps in host order
L.dd_tools_set_tuning(37, 1)          # branches for fp32 caches too
for kv in sys.argv[4:]:
    for p in kv.split(","):
        k_, v_ = p.split("=")
        L.dd_tools_set_tuning(int(k_), int(v_))
probs = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
shapes = [(608, 5, 576), (640, 9, 576), (600, 1, 576), (615, 20, 576), (609, 5, 576), (700, 60, 576),
          (610, 3, 576), (633, 7, 576), (655, 11, 576), (602, 2, 576), (690, 33, 576), (611, 4, 576), (644, 8, 576), (603, 2, 576),
          (620, 6, 576), (699, 30, 576)][:NL]
engs = []
for i in range(len(shapes)):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=768, max_visual=576, seed=5217, kv_format=KV,
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(1, 0.02)
embs = [torch.randn(T0, 4096, generator=torch.Generator().manual_seed(50 + i)).cuda() for i, (T0, _, _) in enumerate(shapes)]
CAP = 64
STAGE = ["embed", "q rows (qkv GEMV + finish)", "new K rows (qkv GEMV + finish)", "attention -> o_proj operand", "o_proj -> residual rows",
         "gate/up -> down operand", "down -> residual rows", "down -> next operand", "attention tiles: statistics (partial kernel)",
         "attention tiles: outputs (partial kernel)", "new V rows", "2nd launch: tile statistics", "2nd launch: tile outputs"] + ["-"] * 3
ORDER = [0, 1, 2, 10, 8, 9, 11, 12, 3, 4, 5, 6, 7]          # pipeline order of the stages within a layer
trace = torch.zeros(CAP, 32, 16, dtype=torch.int32, device="cuda")
ATTN_WGS = 32 * 12 * 8                     # workgroups of the widest attention tile pass (kv heads x key tiles x groups)
REPLAY = os.environ.get("DD_BISECT_REPLAY", "0") not in ("", "0")      # every traced attention launched twice: do the two launches agree?
ATTN = os.environ.get("DD_BISECT_ATTN", "1") not in ("", "0") and KV == "fp32"
CAP_A = 2 * STEPS
attn = torch.zeros(CAP_A, 32, ATTN_WGS * 8, dtype=torch.int32, device="cuda") if ATTN else None
SLOT = ["K registers (global loads)", "V registers (global loads)", "q rows read from LDS", "scores read from LDS", "p written", "p read from LDS",
        "outputs read from LDS", "-"]


def run(branches):
    L.dd_tools_set_tuning(23, branches)
    for e, x, (T0, s0, Lv) in zip(engs, embs, shapes):
        e.rng.manual_seed(5217)
        e.prefill(x, s0, Lv)
    torch.cuda.synchronize()
    trace.zero_()
    if ATTN:
        attn.zero_()
    torch.cuda.synchronize()
    L.dd_tools_sweep_trace(trace.data_ptr(), (CAP if not ATTN else CAP_A) * (-1 if REPLAY else 1))
    if ATTN:
        L.dd_tools_attn_trace(attn.data_ptr(), ATTN_WGS * 8)
    grp = lm.EngineGroup(engs)
    for s in range(STEPS):
        grp.decode_step(probs)
    torch.cuda.synchronize()
    L.dd_tools_sweep_trace(None, 0)
    L.dd_tools_attn_trace(None, 0)
    return trace.cpu().numpy().copy(), [e.tokens() for e in engs], [e.logits().copy() for e in engs], (attn.cpu().numpy().copy() if ATTN else None)


ref, rtoks, rlog, ra = run(1)
ref2, _, _, ra2 = run(1)
if ATTN:
    print('one branch twice: attention traces', 'equal' if np.array_equal(ra, ra2) else 'DIFFER (!)')
print("one branch twice: traces", "equal" if np.array_equal(ref, ref2) else "DIFFER (!)")
n_sweeps = int((ref.reshape(CAP, -1) != 0).any(1).sum())
print(f"{n_sweeps} multi-group sweeps traced over {STEPS} steps")
for rep in range(4):
    got, toks, logs, ga = run(2)
    bad = np.argwhere(got != ref)
    lanes_bad = [i for i in range(len(engs)) if not np.array_equal(logs[i], rlog[i])]
    if len(bad) == 0:
        print(f"rep {rep}: two branches: trace equal; lanes with different final logits: {lanes_bad}")
        continue
    by_sweep = {}
    for sw, ly, stg in bad:
        by_sweep.setdefault(int(sw), []).append((int(ly), ORDER.index(int(stg))))
    msg = "; ".join(f"sweep {sw}: first at layer {min(v)[0]} stage '{STAGE[ORDER[min(v)[1]]]}' ({len(v)} cells)" for sw, v in sorted(by_sweep.items()))
    print(f"rep {rep}: two branches: {len(bad)} trace cells differ; {msg}; lanes with different final logits: {lanes_bad}", flush=True)
    if REPLAY:
        g9, g12, r9 = got[:, :, 9], got[:, :, 12], ref[:, :, 9]
        live = r9 != 0
        print(f"      tile outputs over {int(live.sum())} (sweep, layer) cells: 1st launch != reference in {int((g9 != r9)[live].sum())}, 2nd launch != reference in "
              f"{int((g12 != r9)[live].sum())}, 1st != 2nd in {int((g9 != g12)[live].sum())}; in the one-branch run 1st != 2nd in {int((ref[:, :, 9] != ref[:, :, 12])[live].sum())}", flush=True)
    if ATTN:
        # inside the tile pass of each sweep's FIRST differing layer: which of the workgroup's checksums differ
        for sw, v in sorted(by_sweep.items()):
            ly = min(v)[0]
            a1, a2 = ra[sw, ly].reshape(-1, 8), ga[sw, ly].reshape(-1, 8)
            wg_bad = np.argwhere((a1 != a2).any(1)).flatten()
            slots = sorted({int(c) for w in wg_bad for c in np.argwhere(a1[w] != a2[w]).flatten()})
            wgs = [(int(w) % 32, (int(w) // 32) % 12, int(w) // (32 * 12)) for w in wg_bad[:6]]
            print(f"      sweep {sw} layer {ly}: {len(wg_bad)} of {int((a1 != 0).any(1).sum())} tile workgroups differ; checksums that differ: {[SLOT[c] for c in slots]}; "
                  f"first (kv head, tile, group): {wgs}", flush=True)
