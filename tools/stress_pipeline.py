"""Determinism stress of the forms that SHIP, on libdropdec.so — not the tools library:

    python tools/stress_pipeline.py <config> <repetitions> [batches per repetition = 4] [images per batch]

config 3: `GroupPipeline` as bench.py drives it (LLaVA-1.5-7B shapes, K = 8, 64 images per batch, the next batch's vision tower + batched prefill
          on a second stream BESIDE the decode steps and their sampler);  2: K = 4 [0.1, 0.3, 0.5, 0.7], 56 images per batch (half planes, groups of
          fourteen);  5: LLaVA-NeXT-Mistral-7B shapes, fp8 matrices, 2928 visual tokens (the fp8 nine-plane rider step);  4: InstructBLIP.
A repetition = every lane's generator re-seeded, then `batches` batches through the pipeline (128 new tokens per image: 128 group steps per batch,
so both lane sets and the overlap are exercised from the second batch on).  Every repetition's token ids are compared with the first repetition's,
image by image; and per repetition ONE image of the first batch is decoded alone through `model.generate()` from the same generator state and
compared with its lane.  Exit code 1 when anything differs.  DD_STRESS_LOG=path appends one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dropoutdecoding_amd import _lib, lm
from dropoutdecoding_amd.config import settings, VOTING_NUMBERS_K8
from dropoutdecoding_amd.vlm import GroupPipeline

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_inputs            # the bench's inputs

torch.cuda.set_device(0)
cfg_no = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 4
B = int(sys.argv[4]) if len(sys.argv) > 4 else {2: 56, 3: 64, 4: 64, 5: 64}[cfg_no]
N_NEW = int(os.environ.get("DD_STRESS_NEW", "128"))
SEED = 5217
settings["voting_numbers"] = [0.1, 0.3, 0.5, 0.7] if cfg_no == 2 else list(VOTING_NUMBERS_K8)
if cfg_no == 4:
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration as M
elif cfg_no == 5:
    from dropoutdecoding_amd.llavanext import CustomLlavaNextForConditionalGeneration as M
else:
    from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration as M
model = M.from_synthetic(max_new_tokens=N_NEW + 8)
eng = model.engine
pipe = GroupPipeline(model, lanes=B)
pipe.prefill_chunk = 4 if cfg_no == 5 else 16
if cfg_no == 5:
    pipe.tower_chunk = 6              # as bench.py --config 5: the anyres tiles of three images per vision-tower call (round 6)
lanes_all = pipe.sets[0] + pipe.sets[1]
prompt_len = 32


def batch_inputs(i, n=B):
    out = []
    for b in range(n):
        if cfg_no == 4:
            rng = np.random.default_rng(i * B + b)
            px = torch.from_numpy(rng.standard_normal((1, 3, 224, 224), dtype=np.float32))
            ids = torch.from_numpy(rng.integers(3, 31999, size=prompt_len).astype(np.int64))[None]
            qids = torch.from_numpy(rng.integers(1000, 30000, size=12).astype(np.int64))[None]
            out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda(), qformer_input_ids=qids.cuda(), qformer_attention_mask=torch.ones_like(qids).cuda()))
            continue
        ids, px = synthetic_inputs(i * B + b, eng.cfg.vocab_size, model.image_token_index)
        if cfg_no == 5:
            rng = np.random.default_rng(7_000_000 + i * B + b)
            px = torch.from_numpy(rng.standard_normal((1, 5, 3, 336, 336), dtype=np.float32))
            out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda(), image_sizes=torch.tensor([[672, 672]])))
            continue
        out.append(dict(input_ids=ids.cuda(), pixel_values=px.cuda()))
    return out


def reseed():
    torch.cuda.synchronize()
    for m in lanes_all:
        m.engine.rng.manual_seed(SEED)
    torch.cuda.synchronize()


first, bad, bad_solo, events = None, 0, 0, []
t0 = time.time()
reps = 0
for rep in range(R):
    reseed()
    outs = [[o[0].tolist() for o in batch] for batch in pipe.run((batch_inputs(i) for i in range(NB)), max_new_tokens=N_NEW, eos_token_id=[])]
    assert len(outs) == NB and all(len(b) == B for b in outs)
    reps += 1
    # one image of the first batch alone, from the same generator state (a lane's first image starts from a freshly seeded stream)
    b = rep % B
    reseed()
    eng.set_speculation("never" if rep % 2 else "default")     # both single-sequence step forms give the group's tokens
    solo = model.generate(**batch_inputs(0, b + 1)[b], max_new_tokens=N_NEW, eos_token_id=[])[0].tolist()
    eng.set_speculation("default")
    if solo != outs[0][b]:
        bad_solo += 1
        k = next((j for j in range(min(len(solo), len(outs[0][b]))) if solo[j] != outs[0][b][j]), -1)
        events.append(f"rep {rep}: image {b} of batch 0 differs from its solo run at id {k}")
        print(events[-1], flush=True)
    if first is None:
        first = outs
        continue
    diff = [(bi, li) for bi in range(NB) for li in range(B) if outs[bi][li] != first[bi][li]]
    if diff:
        bad += 1
        bi, li = diff[0]
        k = next(j for j in range(len(first[bi][li])) if outs[bi][li][j] != first[bi][li][j])
        events.append(f"rep {rep}: {len(diff)} images differ from repetition 0; first: batch {bi} lane {li} at id {k}")
        print(events[-1], flush=True)
    if rep % 5 == 4:
        print(f"  ... rep {rep + 1}: {bad} differing, {bad_solo} solo checks differing, {time.time() - t0:.0f} s", flush=True)
summary = {"tool": "stress_pipeline", "library": os.path.basename(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else "libdropdec.so", "config": cfg_no,
           "images_per_batch": B, "batches_per_repetition": NB, "repetitions": reps, "group_steps_total": reps * NB * N_NEW, "K": len(settings["voting_numbers"]),
           "differ_from_first": bad, "solo_checks": reps, "differ_from_solo": bad_solo, "events": events[:20], "seconds": round(time.time() - t0, 1)}
print(json.dumps(summary))
if os.environ.get("DD_STRESS_LOG"):
    with open(os.environ["DD_STRESS_LOG"], "a") as f:
        f.write(json.dumps(summary) + "\n")
sys.exit(1 if bad or bad_solo else 0)
