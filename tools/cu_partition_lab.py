"""Prefill beside decode by CU partition: the go / no-go measurement.

The headline batch is 127 rider steps of 64 lanes (HBM-bound, one 144-KiB workgroup per CU) plus the vision tower and four batched
prefills of 16 prompts (matrix-core-bound, one 144-KiB workgroup per CU): GroupPipeline enqueues the next batch's prefill on a second
stream, but the two kinds of workgroup cannot share a CU, so the launches time-slice the chip.  Here the prefill stream is confined to P
of the 256 CUs (hipExtStreamCreateWithCUMask) and the decode steps keep running unmasked on whatever is free:

  python tools/cu_partition_lab.py probe            where the mask bits land (XCC_ID / HW_ID per workgroup)
  python tools/cu_partition_lab.py run [lanes]      ms per decode step and ms per 16-prompt prefill pass: apart, together unmasked
                                                    (what the product does today), together with the prefill on P = 32 ... 128 CUs
"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.config import VOTING_NUMBERS_K8

torch.cuda.set_device(0)
L = _lib.load()


def masked_stream(bits):
    """a torch stream over a HIP stream restricted to the CUs in `bits` (driver numbering 0..255)"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    h = C.c_void_p()
    _lib.check(L.dd_tools_stream_create_cu_mask(words, 8, C.byref(h)), "dd_tools_stream_create_cu_mask")
    return torch.cuda.ExternalStream(h.value), h


def probe(bits, wgs=2048, hold=200):
    st, h = masked_stream(bits)
    out = torch.zeros(2 * wgs, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    _lib.check(L.dd_tools_cu_probe(out.data_ptr(), wgs, hold, h.value), "dd_tools_cu_probe")
    st.synchronize()
    o = out.cpu().numpy().astype(np.uint32).reshape(wgs, 2)
    xcc = o[:, 0] & 0xF
    cu, sh, se = (o[:, 1] >> 8) & 0xF, (o[:, 1] >> 12) & 0x1, (o[:, 1] >> 13) & 0x7
    L.dd_tools_stream_destroy(h)
    places = sorted({(int(x), int(s), int(a), int(c)) for x, s, a, c in zip(xcc, se, sh, cu)})
    per_xcc = {x: sum(1 for p in places if p[0] == x) for x in sorted({p[0] for p in places})}
    return places, per_xcc


def spread(P, layout):
    """P CUs, P / 8 from every XCD, under the bit layout the probe found: 'interleaved' (bit i -> XCD i % 8) or 'blocked' (i // 32)"""
    per = P // 8
    if layout == "interleaved":
        return [i for i in range(256) if (i >> 3) < per]
    return [i for i in range(256) if (i & 31) < per]


def find_layout():
    _, a = probe(range(0, 32))
    _, b = probe([i for i in range(256) if i % 8 == 0])
    print(f"mask bits 0..31      -> distinct CUs per XCC_ID {a}")
    print(f"mask bits i % 8 == 0 -> distinct CUs per XCC_ID {b}")
    # (measured, round 6: bits 0..31 give 4 CUs on each of the 8 XCCs — bit i is CU i // 8 of XCC i % 8; a mask that leaves whole XCCs
    # without a CU, like the second one, is not honoured: all 256 CUs show up)
    if len(a) == 8 and max(a.values()) <= 4:
        return "interleaved"
    if len(a) == 1 and len(b) == 8:
        return "blocked"
    return "unknown"


if len(sys.argv) < 2 or sys.argv[1] == "probe":
    lay = find_layout()
    print("layout:", lay)
    for P in (32, 64, 96):
        if lay != "unknown":
            places, per = probe(spread(P, lay))
            print(f"P = {P}: {len(places)} distinct (XCC, SE, SH, CU) places, per XCC {per}")
    sys.exit(0)

B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
PF = 16
steps = int(os.environ.get("DD_LAB_STEPS", "28"))
passes = int(os.environ.get("DD_LAB_PASSES", "4"))
lay = find_layout()
print("layout:", lay, flush=True)
engs = []
for i in range(B + PF):
    engs.append(lm.DropoutEngine(lm.LLAVA15_7B, family=lm.FAMILY_LLAVA, max_seq=784, max_visual=576, kv_format="fp16",
                                 share_weights_with=engs[0] if engs else None))
engs[0].load_synthetic(0, 0.02)
dec, pre = engs[:B], engs[B:]
embs = [torch.randn(608, 4096, generator=torch.Generator().manual_seed(i)).cuda() for i in range(PF)]
spans = [(5, 576)] * PF
for c in range(0, B, PF):
    lm.prefill_group(dec[c:c + PF], embs, spans)
for e in dec:
    e.rng.manual_seed(24)
grp = lm.EngineGroup(dec)
for _ in range(6):
    grp.decode_step(VOTING_NUMBERS_K8)
torch.cuda.synchronize()
dstream = dec[0].torch_stream


def refill():
    """the decode lanes back to their prompts (a lane's cache holds 784 tokens: 608 + the steps of two or three runs)"""
    torch.cuda.synchronize()
    for c in range(0, B, PF):
        lm.prefill_group(dec[c:c + PF], embs, spans)
    for e in dec:
        e.rng.manual_seed(24)
    for _ in range(3):
        grp.decode_step(VOTING_NUMBERS_K8)
    torch.cuda.synchronize()


def run(pstream, n_steps, n_passes):
    """-> (ms per decode step, [ms per prefill pass]) with both streams fed at once"""
    if n_steps > 40:
        refill()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    d0, d1 = ev(), ev()
    pe = [ev() for _ in range(n_passes + 1)]
    torch.cuda.synchronize()
    if n_passes:
        pe[0].record(pstream)
        for k in range(n_passes):
            lm.prefill_group(pre, embs, spans, stream=pstream)
            pe[k + 1].record(pstream)
    if n_steps:
        d0.record(dstream)
        for _ in range(n_steps):
            grp.decode_step(VOTING_NUMBERS_K8)
        d1.record(dstream)
    torch.cuda.synchronize()
    return (d0.elapsed_time(d1) / n_steps if n_steps else None), [pe[k].elapsed_time(pe[k + 1]) for k in range(n_passes)]


plain = torch.cuda.Stream()
run(plain, 2, 1)                                           # warm-up: graphs captured, prefill scratch grown
d_alone, _ = run(plain, steps, 0)
_, p_alone = run(plain, 0, passes)
print(f"apart: decode {d_alone:.2f} ms per step of {B} lanes; prefill of {PF} prompts {np.mean(p_alone):.1f} ms per pass "
      f"({', '.join(f'{x:.0f}' for x in p_alone)})", flush=True)
d_t, p_t = run(plain, steps, passes)
print(f"together, unmasked (the product today): decode {d_t:.2f} ms per step over {steps} steps; prefill passes {', '.join(f'{x:.0f}' for x in p_t)} ms",
      flush=True)
# a batch = 127 steps + 4 passes (+ the vision tower, left out here): serial, and with the passes hidden behind the steps at the measured rates
serial = 127 * d_alone + 4 * np.mean(p_alone)
print(f"batch by these numbers: serial {serial:.0f} ms", flush=True)
if lay != "unknown":
    for P in (32, 64, 96, 128):
        st, h = masked_stream(spread(P, lay))
        run(st, 1, 1)
        _, p_only = run(st, 0, 2)
        refill()
        n_pass = 2 if P >= 64 else 1
        n_st = max(steps, int(1.2 * n_pass * np.mean(p_only) / (1.9 * d_alone)))   # decode steps to cover the passes
        d_m, p_m = run(st, min(n_st, 100), n_pass)
        # batch: the four passes run beside decode steps at the together-rate, the remaining steps at the alone-rate
        t_pf = 4 * np.mean(p_m)
        steps_beside = min(127.0, t_pf / d_m)
        batch = max(t_pf, steps_beside * d_m) + (127 - steps_beside) * d_alone if steps_beside < 127 else max(t_pf, 127 * d_m)
        print(f"prefill on {P} CUs ({P // 8} per XCD): alone {np.mean(p_only):.0f} ms per pass; together: decode {d_m:.2f} ms per step "
              f"({d_m / d_alone:.2f} x), prefill {', '.join(f'{x:.0f}' for x in p_m)} ms per pass ({np.mean(p_m) / np.mean(p_alone):.2f} x the whole chip); "
              f"batch {batch:.0f} ms = {serial / batch:.3f} x serial", flush=True)
        pass    # (the masked streams are left to the process exit: destroying one with torch still holding its ExternalStream crashed at exit)
