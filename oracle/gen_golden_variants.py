"""Golden vectors for the reference's DORMANT variants, produced by calling the reference's own
functions (build container only; /root/reference is read-only and never copied):

  * `get_image_attention_mask(..., method="epis_no_overlap")` of models/llava.py:663-683 driven the way llava.py:342-346
    drives "epis" (mask NOT reset between members) and of models/instructblip.py:486-505 with the reset of :121;
  * `select_by_average` (models/llava.py:37-52) on K random logit rows.

Inputs are the g3 cases already in tests/golden/g3_masks.npz, so the two files stay comparable.

    python -m oracle.gen_golden_variants        # writes tests/golden/g7_variants.npz
"""
from __future__ import annotations

import os
import types

from .gen_golden import OUT, REF, _import_reference


def main() -> int:
    if not os.path.isdir(REF):
        print("gen_golden_variants: /root/reference not present; nothing to do")
        return 0
    import numpy as np
    import torch
    RL, RN, RI, settings = _import_reference()
    torch.set_grad_enabled(False)
    NS = types.SimpleNamespace
    g3 = np.load(os.path.join(OUT, "g3_masks.npz"))
    out = {}
    n = int(g3["n_cases"])
    for ci in range(n):
        epi = torch.from_numpy(g3[f"c{ci}_epi"])
        probs = [float(p) for p in g3[f"c{ci}_probs"]]
        topk_ids = torch.from_numpy(g3[f"c{ci}_topk_ids"])[None]
        step_logits = torch.from_numpy(g3[f"c{ci}_step_logits"])[None, None]
        seed, start, T = int(g3[f"c{ci}_seed"]), int(g3[f"c{ci}_start"]), int(g3[f"c{ci}_T"])
        L = epi.numel()
        for fam, cls, reset in (("llava_no_overlap", RL.CustomLlavaForConditionalGeneration, False),
                                ("iblip_no_overlap", RI.CustomLlamaForCausalLM, True)):
            ns = NS(image_features=(None, topk_ids), start_image_pos=[start], end_image_pos=[start + L - 1],
                    vision_uncert_dict={"epis_uncert_per_token": epi[None]}, masked_numbers=[])
            ns.get_overlap_image_tokens = lambda lg, _c=cls, _n=ns: _c.get_overlap_image_tokens(_n, lg)
            torch.manual_seed(seed)
            mask = torch.ones(1, T, dtype=torch.long)
            masks = []
            for p in probs:
                if reset:
                    mask[:, :] = 1                                     # instructblip.py:121
                mask = cls.get_image_attention_mask(ns, step_logits, mask, method="epis_no_overlap", prob=p)
                masks.append(mask[0].clone())
            out[f"c{ci}_{fam}_masks"] = torch.stack(masks).numpy().astype(np.uint8)
    # epis_kl (instructblip.py:464-485 + lowest_percent_kl_indices :559-578), driven like the commented call at :123 with the
    # reset of :121; image_logits are random per case (the g3 cases carry none) and stored with the fixture
    for ci in range(n):
        epi = torch.from_numpy(g3[f"c{ci}_epi"])
        if epi.numel() > 100:
            continue                                     # keep the fixture small: the 576-token case is left out
        probs = [float(p) for p in g3[f"c{ci}_probs"]]
        step_logits = torch.from_numpy(g3[f"c{ci}_step_logits"])[None, None]
        seed, start, T = int(g3[f"c{ci}_seed"]), int(g3[f"c{ci}_start"]), int(g3[f"c{ci}_T"])
        L, V = epi.numel(), step_logits.shape[-1]
        img = (torch.randn(1, L, V, generator=torch.Generator().manual_seed(1000 + ci)) * 2.0).float()
        cls = RI.CustomLlamaForCausalLM
        ns = NS(image_features=(None, torch.from_numpy(g3[f"c{ci}_topk_ids"])[None]), start_image_pos=[start],
                end_image_pos=[start + L - 1], vision_uncert_dict={"epis_uncert_per_token": epi[None]}, masked_numbers=[],
                image_logits=img)
        ns.get_overlap_image_tokens = lambda lg, _c=cls, _n=ns: _c.get_overlap_image_tokens(_n, lg)
        ns.lowest_percent_kl_indices = lambda a, b, percent=0.1, _c=cls, _n=ns: _c.lowest_percent_kl_indices(_n, a, b, percent)
        torch.manual_seed(seed)
        mask = torch.ones(1, T, dtype=torch.long)
        masks = []
        for p in probs:
            mask[:, :] = 1                                             # instructblip.py:121
            mask = cls.get_image_attention_mask(ns, step_logits[0], mask, method="epis_kl", prob=p)
            masks.append(mask[0].clone())
        out[f"c{ci}_kl_image_logits"] = img[0].numpy()
        out[f"c{ci}_kl_lowest"] = cls.lowest_percent_kl_indices(ns, img, step_logits[0]).numpy().astype(np.int64)
        out[f"c{ci}_iblip_kl_masks"] = torch.stack(masks).numpy().astype(np.uint8)
    out["n_cases"] = np.int64(n)
    # select_by_average: outputs_all[k][0] is the member's logits [1, 1, V]
    for ai, (K, V, seed) in enumerate([(3, 200, 1), (8, 4099, 2), (5, 512, 3)]):
        gen = torch.Generator().manual_seed(seed)
        rows = (torch.randn(K, V, generator=gen) * 4.0).float()
        outs = [[rows[k].clone()[None, None]] for k in range(K)]
        avg = RL.select_by_average(outs)
        out[f"avg{ai}_rows"] = rows.numpy()
        out[f"avg{ai}_mean"] = avg[0][0, 0].numpy()
    out["n_avg"] = np.int64(3)
    np.savez_compressed(os.path.join(OUT, "g7_variants.npz"), **out)
    print("wrote", os.path.join(OUT, "g7_variants.npz"))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
