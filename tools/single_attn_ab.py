"""One image at a time: tokens/s and ms per decode step with round 6's two single-sequence kernel forms on and off (same bits either way:
tests/test_gpu_single_stream_attn.py).  key 54 = the looping 8-row qkv GEMV (k_gemv_loop), key 55 = the attention merge launches' early loads.
python tools/single_attn_ab.py [n_images]        (profiles/r06_lab/attn_one_launch_ab.log: the run that also had the one-launch attention)"""
import os, sys, time
os.environ.setdefault("DD_USE_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_inputs
from dropoutdecoding_amd import config as ddcfg, _lib
from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration

torch.cuda.set_device(0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ddcfg.settings["voting_numbers"] = ddcfg.VOTING_NUMBERS_K8
model = CustomLlavaForConditionalGeneration.from_synthetic(max_new_tokens=136)
T = _lib.load()
eng = model.engine
n_new = 128


def one(i):
    ids, px = synthetic_inputs(i, eng.cfg.vocab_size, model.image_token_index)
    return model.generate(input_ids=ids.cuda(), pixel_values=px.cuda(), max_new_tokens=n_new, eos_token_id=[])


def decode_only(i, steps=120):
    ids, px = synthetic_inputs(i, eng.cfg.vocab_size, model.image_token_index)
    model._prepare(ids.cuda(), n_new, None, 1, [], False, dict(pixel_values=px.cuda()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.decode_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for spec in (0, 1):
    for k54, k55 in ((0, 0), (1, 0), (0, 1), (1, 1), (0, 0), (1, 1)):
        T.dd_tools_set_tuning(54, k54)
        T.dd_tools_set_tuning(55, k55)
        T.dd_tools_set_tuning(14, spec)
        one(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_img):
            one(1 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms = decode_only(50)
        print(f"spec={spec} looping qkv GEMV={k54} early merge loads={k55}: {n_img * n_new / dt:.1f} tok/s end to end, {ms:.3f} ms per decode step", flush=True)
