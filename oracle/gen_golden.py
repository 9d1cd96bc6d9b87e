"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

Imports kigb/DropoutDecoding from /root/reference (read-only; never copied) under a small
compatibility shim for the installed transformers 5.15 (the reference targets 4.44):
4.44 attribute layout on the wrapper objects + legacy cache subscripting
(reference models/llava.py:257).  Outputs are DATA ONLY: inputs, seeds and the arrays the
reference produced.  No-op (exit 0 with a message) where /root/reference is absent, e.g.
on the GPU box.

    python -m oracle.gen_golden            # writes tests/golden/g1..g6
"""
from __future__ import annotations

import json
import os
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for n in ("turtledemo", "turtledemo.forest"):          # stray import at reference models/instructblip.py:3
        m = types.ModuleType(n)
        m.start = None
        sys.modules[n] = m
    import models.llava as RL
    import models.llavanext as RN
    import models.instructblip as RI
    from models.config import settings
    return RL, RN, RI, settings


def main() -> int:
    if not os.path.isdir(REF):
        print("gen_golden: /root/reference not present; nothing to do")
        return 0
    import numpy as np
    import torch
    from transformers import (CLIPVisionConfig, CLIPVisionModel, DynamicCache, LlamaConfig, LlamaForCausalLM,
                              LlavaConfig, LlavaNextConfig, MistralConfig, MistralForCausalLM)
    from transformers.models.llava.modeling_llava import LlavaMultiModalProjector
    from transformers.models.llava_next.modeling_llava_next import LlavaNextMultiModalProjector

    RL, RN, RI, settings = _import_reference()
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from oracle.lm_ref import LMConfig, random_weights

    os.makedirs(OUT, exist_ok=True)
    torch.set_grad_enabled(False)
    meta = {"torch": torch.__version__, "transformers": __import__("transformers").__version__,
            "numpy": np.__version__, "reference": "kigb/DropoutDecoding @ 2024-12-20"}

    class LegacyCache(DynamicCache):                      # reference llava.py:257 subscripts the cache
        def __getitem__(self, i):
            return (self.layers[i].keys, self.layers[i].values)

    NS = types.SimpleNamespace

    # ---------------- G1: uncertainty + top-k -------------------------------------------
    g1 = {}
    for ci, (L, V, k, scale) in enumerate([(8, 200, 5, 3.0), (32, 512, 10, 5.0), (64, 512, 5, 1.0), (36, 1000, 10, 8.0)]):
        gen = torch.Generator().manual_seed(100 + ci)
        logits = (torch.randn(1, L, V, generator=gen) * scale).float()
        d = RL.CustomLlavaForConditionalGeneration.calculate_vision_uncertainty(None, logits)
        vals, ids = RL.CustomLlavaForConditionalGeneration.get_topk_token_id(None, logits, topk=k)
        g1[f"c{ci}_logits"] = logits.numpy()
        g1[f"c{ci}_k"] = np.int64(k)
        for key, t in d.items():
            g1[f"c{ci}_{key}"] = t.numpy()
        g1[f"c{ci}_topk_vals"], g1[f"c{ci}_topk_ids"] = vals.numpy(), ids.numpy()
    g1["n_cases"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "g1_uncertainty.npz"), **g1)

    # ---------------- G2: overlap keep --------------------------------------------------
    g2 = {}
    gen = torch.Generator().manual_seed(7)
    L, V, k, start = 24, 64, 5, 3
    topk_ids = torch.stack([torch.randperm(V, generator=gen)[:k] for _ in range(L)])[None]
    cases = []
    for want in ("many", "one", "none", "rand0", "rand1"):
        sl = torch.randn(1, 1, V, generator=gen)
        if want == "one":          # a token present in exactly one row
            cnt = torch.bincount(topk_ids.flatten(), minlength=V)
            sl[0, 0, int((cnt == 1).nonzero()[0])] = 50.0
        elif want == "none":
            cnt = torch.bincount(topk_ids.flatten(), minlength=V)
            sl[0, 0, int((cnt == 0).nonzero()[0])] = 50.0
        elif want == "many":
            cnt = torch.bincount(topk_ids.flatten(), minlength=V)
            sl[0, 0, int(cnt.argmax())] = 50.0
        ns = NS(image_features=(None, topk_ids), start_image_pos=[start])
        idx = RL.CustomLlavaForConditionalGeneration.get_overlap_image_tokens(ns, sl)
        cases.append((sl, idx))
    g2["topk_ids"], g2["start"] = topk_ids[0].numpy(), np.int64(start)
    for i, (sl, idx) in enumerate(cases):
        g2[f"c{i}_logits"] = sl[0, 0].numpy()
        g2[f"c{i}_idx"] = np.atleast_1d(idx.numpy()).astype(np.int64)
        g2[f"c{i}_idx_ndim"] = np.int64(idx.dim())
    g2["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "g2_overlap.npz"), **g2)

    # ---------------- G3: masks, three family variants ------------------------------------
    g3 = {}
    ci = 0
    for (L, seed, probs, n_keep, start, tail) in [(576, 24, [0.1, 0.3, 0.5, 0.7], 7, 5, 9), (36, 5217, [0.3, 0.5, 0.7], 2, 1, 4),
                                                  (88, 506, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8], 0, 1, 3),
                                                  (32, 11, [0.3, 0.5, 0.7], 3, 0, 6), (33, 12, [0.5], 1, 2, 2),
                                                  (16, 13, [0.3, 0.7], 0, 0, 1)]:
        gen = torch.Generator().manual_seed(1000 + ci)
        epi = torch.rand(L, generator=gen).float() * 3.0
        if ci == 5:
            epi[:] = 1.25                                              # hi == lo  => NaN probs => nothing dropped
        V, k = 97, 5
        topk_ids = torch.stack([torch.randperm(V, generator=gen)[:k] for _ in range(L)])[None]
        step_logits = torch.randn(1, 1, V, generator=gen)
        if n_keep == 0:
            cnt = torch.bincount(topk_ids.flatten(), minlength=V)
            z = (cnt == 0).nonzero()
            if len(z):
                step_logits[0, 0, int(z[0])] = 50.0
        T = start + L + tail
        for fam, mod in (("llava", RL), ("next", RN), ("next_no_overlap", RN), ("iblip", RI)):
            cls = {"llava": RL.CustomLlavaForConditionalGeneration, "next": RN.CustomLlavaNextForConditionalGeneration,
                   "next_no_overlap": RN.CustomLlavaNextForConditionalGeneration, "iblip": RI.CustomLlamaForCausalLM}[fam]
            ns = NS(image_features=(None, topk_ids), start_image_pos=[start], end_image_pos=[start + L - 1],
                    vision_uncert_dict={"epis_uncert_per_token": epi[None]}, masked_numbers=[])
            ns.get_overlap_image_tokens = lambda lg, _c=cls, _n=ns: _c.get_overlap_image_tokens(_n, lg)
            torch.manual_seed(seed)
            mask = torch.ones(1, T, dtype=torch.long)
            masks = []
            for p in probs:
                if fam != "llava":
                    mask[:, :] = 1                                     # llavanext.py:546 / instructblip.py:121
                method = "epis_no_overlap" if fam == "next_no_overlap" else "epis"
                mask = cls.get_image_attention_mask(ns, step_logits, mask, method=method, prob=p)
                masks.append(mask[0].clone())
            torch.manual_seed(seed)
            uni = torch.stack([torch.rand(L) for _ in probs])          # the same stream rand_like consumed
            g3[f"c{ci}_{fam}_masks"] = torch.stack(masks).numpy().astype(np.uint8)
            if fam == "llava":
                g3[f"c{ci}_{fam}_masked_numbers"] = np.array(ns.masked_numbers, dtype=np.int64)
        g3[f"c{ci}_epi"], g3[f"c{ci}_probs"] = epi.numpy(), np.array(probs, dtype=np.float64)
        g3[f"c{ci}_topk_ids"], g3[f"c{ci}_step_logits"] = topk_ids[0].numpy(), step_logits[0, 0].numpy()
        g3[f"c{ci}_uniforms"], g3[f"c{ci}_seed"] = uni.numpy(), np.int64(seed)
        g3[f"c{ci}_start"], g3[f"c{ci}_T"] = np.int64(start), np.int64(T)
        ci += 1
    g3["n_cases"] = np.int64(ci)
    np.savez_compressed(os.path.join(OUT, "g3_masks.npz"), **g3)

    # ---------------- G4: vote ---------------------------------------------------------
    g4 = {}
    pats = [[3, 5, 7], [3, 5, 5], [5, 3, 5], [9, 9, 9], [1, 2, 2, 1], [4], [7, 1, 1, 7, 2, 2, 7, 1], [2, 1, 1, 2, 3, 3]]
    V = 12
    for i, pat in enumerate(pats):
        outs = []
        for j, t in enumerate(pat):
            lg = torch.full((1, 1, V), -1.0)
            lg[0, 0, t] = 1.0 + 0.01 * j
            outs.append((lg,))
        o, idx = RL.select_by_vote(outs)
        o2 = RN.select_by_vote(outs)
        assert o2 is o
        g4[f"c{i}_ids"], g4[f"c{i}_winner"] = np.array(pat, dtype=np.int64), np.int64(idx)
    g4["n_cases"] = np.int64(len(pats))
    np.savez_compressed(os.path.join(OUT, "g4_vote.npz"), **g4)

    # ---------------- G6: RNG stream ------------------------------------------------------
    g6 = {}
    for i, (seed, ns_) in enumerate([(24, [1, 8, 15]), (5217, [576, 576, 100]), (506, [2928, 17]), (0, [700])]):
        torch.manual_seed(seed)
        g6[f"c{i}_seed"] = np.int64(seed)
        g6[f"c{i}_sizes"] = np.array(ns_, dtype=np.int64)
        g6[f"c{i}_draws"] = np.concatenate([torch.rand(n).numpy() for n in ns_])
    g6["n_cases"] = np.int64(4)
    np.savez_compressed(os.path.join(OUT, "g6_rng.npz"), **g6)

    # ---------------- G5: end-to-end tiny models through the reference forward -------------
    def load_lm(hf_lm, cfg: LMConfig, wseed: int, std: float):
        w = random_weights(cfg, wseed, std)
        missing, unexpected = hf_lm.load_state_dict(w, strict=False)
        assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
        return w

    def spy_lm(lm_module, calls):
        orig = lm_module.forward

        def fwd(*a, **kw):
            out = orig(*a, **kw)
            am = kw.get("attention_mask")
            calls.append({"mask": None if am is None else am.clone(), "out0": out[0].detach().clone(),
                          "embeds": kw.get("inputs_embeds"), "pos": kw.get("position_ids"),
                          "cache_position": kw.get("cache_position")})
            return out
        lm_module.forward = fwd

    def margins(x):
        t = torch.topk(x.flatten().float(), 2).values
        return float(t[0] - t[1])

    def run_llava_like(kind: str, wseed: int, probs, n_new: int, rseed: int, use_random=False):
        if kind == "llava":
            lc = LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-5, 10000.0)
            tc = LlamaConfig(vocab_size=lc.vocab_size, hidden_size=lc.hidden_size, intermediate_size=lc.intermediate_size,
                             num_hidden_layers=lc.num_layers, num_attention_heads=lc.num_heads,
                             num_key_value_heads=lc.num_kv_heads, head_dim=lc.head_dim, max_position_embeddings=512,
                             rms_norm_eps=lc.rms_eps, rope_theta=lc.rope_theta, attention_bias=False, mlp_bias=False,
                             tie_word_embeddings=False)
            vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                                  image_size=84, patch_size=14, projection_dim=16)
            cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-2,
                              vision_feature_select_strategy="default")
            torch.manual_seed(wseed)
            m = RL.CustomLlavaForConditionalGeneration(cfg)
            m.language_model = LlamaForCausalLM(tc)
            m.vision_tower = CLIPVisionModel(vc)
            m.multi_modal_projector = LlavaMultiModalProjector(cfg)
            m.pad_token_id = -1
            pv = torch.randn(1, 3, 84, 84)
            extra = {"pixel_values": pv}
        else:
            lc = LMConfig(512, 512, 512, 2, 4, 2, 128, 1e-5, 1000000.0)
            tc = MistralConfig(vocab_size=lc.vocab_size, hidden_size=lc.hidden_size, intermediate_size=lc.intermediate_size,
                               num_hidden_layers=lc.num_layers, num_attention_heads=lc.num_heads,
                               num_key_value_heads=lc.num_kv_heads, head_dim=lc.head_dim, max_position_embeddings=1024,
                               rms_norm_eps=lc.rms_eps, rope_theta=lc.rope_theta, sliding_window=None,
                               tie_word_embeddings=False)
            vc = CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                                  image_size=56, patch_size=14, projection_dim=16)
            cfg = LlavaNextConfig(vision_config=vc, text_config=tc, image_token_index=511, vision_feature_layer=-2,
                                  vision_feature_select_strategy="default",
                                  image_grid_pinpoints=[[56, 112], [112, 56], [112, 112]])
            torch.manual_seed(wseed)
            m = RN.CustomLlavaNextForConditionalGeneration(cfg)
            m.language_model = MistralForCausalLM(tc)
            m.vision_tower = CLIPVisionModel(vc)
            m.multi_modal_projector = LlavaNextMultiModalProjector(cfg)
            m.image_newline = torch.nn.Parameter(torch.randn(tc.hidden_size) * 0.5)
            m.padding_side = "left"
            _p = m.model.pack_image_features

            def pack(feats, sizes, image_newline=None):
                f, l = _p(feats, sizes, "default", image_newline=image_newline)
                f = torch.cat(list(f), 0) if isinstance(f, (list, tuple)) else f
                return f, (l if torch.is_tensor(l) else torch.tensor(l))
            m.pack_image_features = pack
            pv = torch.randn(1, 5, 3, 56, 56)
            extra = {"pixel_values": pv, "image_sizes": torch.tensor([[100, 100]])}
        m.get_input_embeddings = lambda: m.language_model.get_input_embeddings()
        m.eval()
        w = load_lm(m.language_model, lc, wseed, 0.05)
        # image features far smaller than text embeddings make every visual token look alike; scale them up
        for p_ in m.multi_modal_projector.parameters():
            p_.mul_(6.0)
        settings["voting_numbers"] = list(probs)
        settings["use_random"] = [bool(use_random)]
        calls = []
        spy_lm(m.language_model, calls)
        ids = torch.tensor([[1, 17, 511, 45, 6, 7, 99]])
        torch.manual_seed(rseed)
        out = m.forward(input_ids=ids, attention_mask=torch.ones_like(ids), past_key_values=LegacyCache(config=tc),
                        use_cache=True, return_dict=True, **extra)
        embeds = calls[0]["embeds"][0].clone()
        start, end = m.start_image_pos[0], m.end_image_pos[0]
        L = end - start + 1
        rec = {"embeds": embeds.numpy(), "span_start": np.int64(start), "span_len": np.int64(L),
               "prefill_logits_last": out.logits[0, -1].numpy(), "prefill_image_logits": out.logits[0, start:end + 1].numpy(),
               "topk_ids": m.image_features[1][0].numpy(), "probs": np.array(probs, dtype=np.float64),
               "wseed": np.int64(wseed), "rseed": np.int64(rseed), "std": np.float64(0.05),
               "cfg": np.array([lc.vocab_size, lc.hidden_size, lc.intermediate_size, lc.num_layers, lc.num_heads,
                                lc.num_kv_heads, lc.head_dim], dtype=np.int64),
               "rms_eps": np.float64(lc.rms_eps), "rope_theta": np.float64(lc.rope_theta),
               "use_random": np.int64(use_random)}
        for key, t in m.vision_uncert_dict.items():
            rec[key] = t.numpy()
        pkv, nxt = out.past_key_values, out.logits[:, -1].argmax(-1, keepdim=True)
        toks = [nxt.item()]
        am = torch.ones(1, ids.shape[1], dtype=torch.long)
        K = len(probs)
        min_logit_margin, min_rp_margin = margins(out.logits[0, -1]), 1.0
        steps = []
        for s in range(n_new - 1):
            am = torch.cat([am, am.new_ones(1, 1)], -1)
            n0 = len(calls)
            out = m.forward(input_ids=nxt, attention_mask=am, past_key_values=pkv, use_cache=True, return_dict=True, **extra)
            cs = calls[n0:]
            assert len(cs) == 1 + K
            base = cs[0]["out0"][0, -1]
            mem = [c["out0"][0, -1] for c in cs[1:]]
            win = [i for i, lg in enumerate(mem) if torch.equal(lg, out.logits[0, -1])][0]
            drop = torch.stack([(c["mask"][0, start:end + 1] == 0) for c in cs[1:]])
            for c in cs[1:]:
                mk = c["mask"][0]
                assert int((mk == 0).sum()) == int((mk[start:end + 1] == 0).sum())
            min_logit_margin = min([min_logit_margin, margins(base)] + [margins(x) for x in mem])
            steps.append({"base_logits": base.numpy(), "base_argmax": int(base.argmax()), "drop": drop.numpy(),
                          "member_argmax": [int(x.argmax()) for x in mem], "winner": win,
                          "logits": out.logits[0, -1].numpy(),
                          "masked_numbers": list(getattr(m, "masked_numbers", []))})
            pkv, nxt = out.past_key_values, out.logits[:, -1].argmax(-1, keepdim=True)
            toks.append(nxt.item())
        # uniforms the run consumed (same seed, same stream)
        torch.manual_seed(rseed)
        uni = torch.stack([torch.stack([torch.rand(L) for _ in range(K)]) for _ in range(n_new - 1)])
        epi = m.vision_uncert_dict["epis_uncert_per_token"][0]
        for s in range(n_new - 1):
            for k_, p in enumerate(probs):
                pr = 0.1 + (p - 0.1) * (epi - epi.min()) / (epi.max() - epi.min())
                min_rp_margin = min(min_rp_margin, float((uni[s, k_] - pr).abs().min()))
        rec["tokens"] = np.array(toks, dtype=np.int64)
        rec["uniforms"] = uni.numpy()
        rec["step_base_logits"] = np.stack([s["base_logits"] for s in steps])
        rec["step_base_argmax"] = np.array([s["base_argmax"] for s in steps], dtype=np.int64)
        rec["step_drop"] = np.stack([s["drop"] for s in steps]).astype(np.uint8)
        rec["step_member_argmax"] = np.array([s["member_argmax"] for s in steps], dtype=np.int64)
        rec["step_winner"] = np.array([s["winner"] for s in steps], dtype=np.int64)
        rec["step_logits"] = np.stack([s["logits"] for s in steps])
        if kind == "llava":
            rec["step_masked_numbers"] = np.array([s["masked_numbers"] for s in steps], dtype=np.int64)
        rec["kv_k_sum"] = np.array([float(pkv.layers[i].keys.double().sum()) for i in range(lc.num_layers)])
        rec["kv_v_sum"] = np.array([float(pkv.layers[i].values.double().sum()) for i in range(lc.num_layers)])
        rec["kv_len"] = np.int64(pkv.layers[0].keys.shape[2])
        rec["min_logit_margin"], rec["min_rp_margin"] = np.float64(min_logit_margin), np.float64(min_rp_margin)
        return rec

    def pick(fn, name, seeds, **kw):
        best = None
        for ws in seeds:
            rec = fn(wseed=ws, **kw)
            score = min(float(rec["min_logit_margin"]) / 1e-2, float(rec.get("min_rp_margin", 1.0)) / 1e-3)
            if best is None or score > best[0]:
                best = (score, rec)
            if score >= 1.0:
                break
        rec = best[1]
        rec["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(OUT, name), **rec)
        print(name, "tokens", rec["tokens"].tolist(), "logit margin %.3g" % rec["min_logit_margin"],
              "r-p margin %.3g" % rec.get("min_rp_margin", np.float64(1.0)))

    pick(lambda wseed, **kw: run_llava_like("llava", wseed, **kw), "g5_llava_k3.npz", range(1, 30),
         probs=[0.3, 0.5, 0.7], n_new=7, rseed=5217)
    pick(lambda wseed, **kw: run_llava_like("llava", wseed, **kw), "g5_llava_k8.npz", range(31, 60),
         probs=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8], n_new=6, rseed=24)
    pick(lambda wseed, **kw: run_llava_like("next", wseed, **kw), "g5_next_k4.npz", range(61, 90),
         probs=[0.1, 0.3, 0.5, 0.7], n_new=6, rseed=506)
    pick(lambda wseed, **kw: run_llava_like("next", wseed, **kw), "g5_next_norestore_k2.npz", range(91, 120),
         probs=[0.5, 0.3], n_new=5, rseed=506, use_random=True)

    # ---- InstructBLIP LM class: module globals + HF generate() with inputs_embeds ---------
    def run_iblip(wseed: int, probs, n_new: int):
        lc = LMConfig(512, 256, 512, 2, 2, 2, 128, 1e-6, 10000.0)
        tc = LlamaConfig(vocab_size=lc.vocab_size, hidden_size=lc.hidden_size, intermediate_size=lc.intermediate_size,
                         num_hidden_layers=lc.num_layers, num_attention_heads=lc.num_heads,
                         num_key_value_heads=lc.num_kv_heads, head_dim=lc.head_dim, max_position_embeddings=512,
                         rms_norm_eps=lc.rms_eps, rope_theta=lc.rope_theta, tie_word_embeddings=False,
                         bos_token_id=1, eos_token_id=None, pad_token_id=0)
        torch.manual_seed(wseed)
        lm = RI.CustomLlamaForCausalLM(tc).eval()
        w = load_lm(lm, lc, wseed, 0.05)
        Q, P = 32, 6
        gen = torch.Generator().manual_seed(wseed)
        emb = torch.cat([torch.randn(1, Q, lc.hidden_size, generator=gen) * 0.7,
                         w["model.embed_tokens.weight"][torch.tensor([1, 17, 45, 6, 7, 99])][None]], dim=1)
        settings["voting_numbers"] = list(probs)
        RI.start_img_pos, RI.end_img_pos, RI.start_generation_pos, RI.first_generation = 0, Q - 1, Q + P, True
        calls = []
        spy_lm(lm.model, calls)
        seq = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(1, Q + P, dtype=torch.long),
                          max_new_tokens=n_new, do_sample=False)
        toks = seq[0].tolist()
        K = len(probs)
        assert len(calls) == 1 + (n_new - 1) * (1 + K), len(calls)
        rec = {"embeds": emb[0].numpy(), "span_start": np.int64(0), "span_len": np.int64(Q),
               "probs": np.array(probs, dtype=np.float64), "wseed": np.int64(wseed), "std": np.float64(0.05),
               "cfg": np.array([lc.vocab_size, lc.hidden_size, lc.intermediate_size, lc.num_layers, lc.num_heads,
                                lc.num_kv_heads, lc.head_dim], dtype=np.int64),
               "rms_eps": np.float64(lc.rms_eps), "rope_theta": np.float64(lc.rope_theta),
               "tokens": np.array(toks, dtype=np.int64), "topk_ids": lm.image_features[1][0].numpy()}
        for key, t in lm.vision_uncert_dict.items():
            rec[key] = t.numpy()
        mm = 1e9
        base_masks, drops, margs, winners, pos_used = [], [], [], [], []
        for s in range(n_new - 1):
            cs = calls[1 + s * (1 + K): 1 + (s + 1) * (1 + K)]
            base_masks.append((cs[0]["mask"][0, :Q] == 0).numpy())
            drops.append(np.stack([(c["mask"][0, :Q] == 0).numpy() for c in cs[1:]]))
            hid = [c["out0"][0, -1] for c in cs[1:]]
            margs.append([int(h.argmax()) for h in hid])
            mm = min([mm] + [margins(h) for h in hid])
            cp = cs[0]["cache_position"]
            pos_used.append(-1 if cp is None else int(cp[-1]))
        rec["step_base_drop"] = np.stack(base_masks).astype(np.uint8)
        rec["step_drop"] = np.stack(drops).astype(np.uint8)
        rec["step_member_argmax"] = np.array(margs, dtype=np.int64)
        rec["step_cache_position"] = np.array(pos_used, dtype=np.int64)
        rec["logits_mask_prob"] = np.array(lm.logits_mask_prob, dtype=np.float64)
        rec["token_entropies"] = np.array(lm.token_entropies, dtype=np.float64)
        rec["min_logit_margin"] = np.float64(mm)
        return rec

    pick(run_iblip, "g5_iblip_k3.npz", range(121, 150), probs=[0.3, 0.5, 0.7], n_new=6)
    print("golden fixtures written to", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
