"""Golden vectors for the Philox (torch GPU generator) stream: run ON THE GPU BOX.

    python oracle/gen_golden_philox.py            -> gpurun_out/g8_philox.npz  (copy to tests/golden/)

Each case: torch.manual_seed(seed), then successive torch.rand(n, device="cuda") calls (what the reference's
torch.rand_like draws at models/llava.py:650 when the model is on a GPU).  Also prints whether oracle/philox.py
reproduces every case, so a mismatch in the float mapping shows up here first.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.philox import TorchGpuPhilox, philox4x32_10  # noqa: E402

CASES = [(0, [5, 576, 576]), (42, [576, 2928, 7]), (5217, [576, 576, 576, 2928]), ((1 << 33) + 5, [1025, 32]),
         (19, [8192]), (23, [300000, 4])]


def main():
    out = {"n_cases": np.int64(len(CASES)), "torch": np.array(torch.__version__), "device": np.array(torch.cuda.get_device_name(0))}
    ok = True
    for c, (seed, sizes) in enumerate(CASES):
        torch.manual_seed(seed)
        draws = [torch.rand(n, device="cuda").cpu().numpy() for n in sizes]
        ref = TorchGpuPhilox(seed)
        mine = [ref.rand_f32(n) for n in sizes]
        for n, a, b in zip(sizes, draws, mine):
            same = np.array_equal(a, b)
            ok &= same
            if not same:
                bad = np.flatnonzero(a != b)
                print(f"case {c} seed {seed} n {n}: {bad.size} differ, first {bad[:4]}, torch {a[bad[:4]]}, oracle {b[bad[:4]]}")
        out[f"c{c}_seed"] = np.uint64(seed)
        out[f"c{c}_sizes"] = np.array(sizes, dtype=np.int64)
        keep = [d if d.size <= 8192 else np.concatenate([d[:2048], d[-2048:]]) for d in draws]   # big cases: head and tail only
        out[f"c{c}_kept"] = np.array([k.size for k in keep], dtype=np.int64)
        out[f"c{c}_draws"] = np.concatenate(keep)
    print("oracle/philox.py == torch.rand(device='cuda'):", ok)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/g8_philox.npz", **out)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
