"""The harness adapters (chair_test/chair_test.py:293-372, pope_test/pope_test.py:221-265) driven over
the HIP-backed drop-in wrappers: tiny random HF LLaVA / InstructBLIP models, a stub processor whose tokenizer is a word
table (no checkpoints, no network), the caption jsonl and the POPE rows checked token for token against the oracle."""
import json
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

from oracle.decode_ref import FAMILY_IBLIP, FAMILY_LLAVA, RefDecoder
from oracle.lm_ref import LMConfig as RefCfg, bf16_round

IMAGE_TOKEN, EOS = 511, 2


class _Inputs(dict):
    def to(self, device):
        return _Inputs({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class WordTableProcessor:
    """A processor with the call shapes the reference uses (positional (prompt, image) for the LLaVA families —
    chair_test.py:294 —, keywords for InstructBLIP and POPE — chair_test.py:289-292, pope_test.py:228) and a tokenizer that
    maps whitespace-separated words to ids through a growing table; id n without a word decodes to "t<n>", so the caption
    text carries the generated ids."""

    def __init__(self, qformer: bool = False):
        self.tokenizer = types.SimpleNamespace(eos_token_id=EOS)
        self.words = {"<s>": 1, "</s>": EOS, "<image>": IMAGE_TOKEN}
        self.next_id = 400                              # prompt words live in 400..510
        self.qformer = qformer
        self.calls = []

    def _ids(self, text: str):
        ids = [1]
        for wd in text.replace("\n", " <nl> ").split(" "):        # the newline of the prompts as a visible word
            if wd == "":
                continue
            if wd not in self.words:
                self.words[wd] = self.next_id
                self.next_id += 1
                assert self.next_id < IMAGE_TOKEN
            ids.append(self.words[wd])
        return ids

    def __call__(self, *a, text=None, images=None, return_tensors=None):
        self.calls.append((a, text is not None))
        if a:
            text, images = a
        ids = self._ids(text)
        out = _Inputs(input_ids=torch.tensor([ids]), attention_mask=torch.ones(1, len(ids), dtype=torch.long), pixel_values=images)
        if self.qformer:
            q = [3 + (i % 90) for i in ids[1:5]]
            out["qformer_input_ids"] = torch.tensor([q])
            out["qformer_attention_mask"] = torch.ones(1, len(q), dtype=torch.long)
        return out

    def batch_decode(self, ids, skip_special_tokens=True):
        inv = {v: k for k, v in self.words.items()}
        res = []
        for row in ids.tolist():
            toks = [t for t in row if not (skip_special_tokens and t in (1, EOS, IMAGE_TOKEN))]
            res.append(" ".join(inv.get(t, f"t{t}") for t in toks))
        return res

    def ids_of(self, text: str):
        inv = self.words
        return [inv[w] if w in inv else int(w[1:]) for w in text.split(" ") if w]


def _visible(ids):
    return [t for t in ids if t not in (1, EOS, IMAGE_TOKEN)]       # what batch_decode(skip_special_tokens=True) keeps


def _tiny_llava():
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration
    torch.manual_seed(0)
    vc = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                          image_size=56, patch_size=14, projection_dim=16)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False, eos_token_id=EOS)
    cfg = LlavaConfig(vision_config=vc, text_config=tc, image_token_index=IMAGE_TOKEN, vision_feature_layer=-2,
                      vision_feature_select_strategy="default")
    hf = LlavaForConditionalGeneration(cfg).eval()
    for p_ in hf.parameters():
        p_.copy_(p_.to(torch.bfloat16).float())
    for n, p in hf.named_parameters():
        if "language_model" in n or "lm_head" in n:
            p.mul_(2.5)
    return hf, tc


@pytest.fixture(scope="module")
def built():
    from dropoutdecoding_amd import build
    build.build()
    return True


def _image(i):
    return torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(500 + i))


def test_caption_loop_and_pope_loop_over_the_hip_llava_wrapper(built, tmp_path):
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd import harness as H
    from dropoutdecoding_amd.llava import CustomLlavaForConditionalGeneration
    from dropoutdecoding_amd.vlm import GroupPipeline, lm_state_dict_from_hf
    hf, tc = _tiny_llava()
    sd = {k: bf16_round(v.detach().float().cpu()) for k, v in lm_state_dict_from_hf(hf).items()}
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, tc.rms_norm_eps, 10000.0)
    saved = dict(ddc.settings)
    try:
        H.apply_cli_settings(4)                                   # --voting-numbers 4: [0.1, 0.3, 0.5, 0.7] (chair_test.py:170)
        ddc._module_imported(24)
        m = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=24)
        proc = WordTableProcessor()
        items = [(391895 + i, f"COCO_val2014_{391895 + i:012d}.jpg") for i in range(3)]
        imgs = {path: _image(i) for i, (_, path) in enumerate(items)}
        log = H.CaptionLog(str(tmp_path / "captions" / "dd.json"))
        n = H.caption_images(m, proc, items, "llava-1.5", log, load_image=imgs.__getitem__, max_new_tokens=12)
        # positional (prompt, image), as chair_test.py:294 calls the LLaVA processors
        assert n == 3 and proc.calls[0][0][0] == H.CHAIR_PROMPTS["llava-1.5"] and proc.calls[0][0][1] is imgs[items[0][1]] and proc.calls[0][1] is False
        rows = H.read_caption_log(log.path)
        assert [r["image_id"] for r in rows] == [i for i, _ in items]
        # the oracle: one reference process captioning the three images back to back on one rng stream, stopping at EOS
        ref = RefDecoder(FAMILY_LLAVA, rc, sd, [0.1, 0.3, 0.5, 0.7], seed=24)
        prompt_ids = torch.tensor([proc._ids(H.CHAIR_PROMPTS["llava-1.5"])])
        wants = []
        for (_, path), row in zip(items, rows):
            emb, start = m._merge(prompt_ids.cuda(), m._visual_embeds(pixel_values=imgs[path]))
            want = ref.generate(emb.cpu(), start, 16, 12, eos=EOS)
            wants.append(want)
            text = proc.batch_decode(torch.tensor([prompt_ids[0].tolist() + want]))[0]
            assert row["caption"] == H.filter_unk_sentences(H.strip_prompt_echo("llava-1.5", text))
            assert proc.ids_of(row["caption"]) == _visible(want)         # the row carries the oracle's ids
        # the same images through a lane pipeline: every image decoded as on a lane of its own (a fresh stream per lane)
        m2 = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=24)
        pipe = GroupPipeline(m2, lanes=2)
        log2 = H.CaptionLog(str(tmp_path / "captions" / "lanes.json"))
        assert H.caption_images(m2, proc, items, "llava-1.5", log2, load_image=imgs.__getitem__, max_new_tokens=12,
                                pipeline=pipe, lanes=2) == 3
        rows2 = H.read_caption_log(log2.path)
        assert [r["image_id"] for r in rows2] == [i for i, _ in items]
        for k, ((_, path), row) in enumerate(zip(items, rows2)):          # images 0, 1 = batch 0 on set 0; image 2 = batch 1 on set 1
            emb, start = m._merge(prompt_ids.cuda(), m._visual_embeds(pixel_values=imgs[path]))
            want = RefDecoder(FAMILY_LLAVA, rc, sd, [0.1, 0.3, 0.5, 0.7], seed=24).generate(emb.cpu(), start, 16, 12, eos=EOS)
            assert proc.ids_of(row["caption"]) == _visible(want), f"lane image {k}"
        # POPE: one generated token per question (pope_test.py:228-241), with and without the image prefix kept between questions
        qs = [{"image": items[0][1], "text": "Is there a dog in the image?", "label": "yes"},
              {"image": items[0][1], "text": "Is there a car in the image?", "label": "no"},
              {"image": items[1][1], "text": "Is there a dog in the image?", "label": "no"}]
        root = str(tmp_path)
        load = lambda p: imgs[os.path.basename(p)]
        m3 = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=24)
        ans = H.answer_pope(m3, proc, qs, "llava", root, load_image=load)
        ref3 = RefDecoder(FAMILY_LLAVA, rc, sd, [0.1, 0.3, 0.5, 0.7], seed=24)
        for q, a in zip(qs, ans):
            ids = torch.tensor([proc._ids(H.pope_prompt("llava", q["text"]))])
            emb, start = m._merge(ids.cuda(), m._visual_embeds(pixel_values=load(q["image"])))
            want = ref3.generate(emb.cpu(), start, 16, 1)
            assert a["question"] == q["text"] and proc.ids_of(a["answer"]) == _visible(want)
        m4 = CustomLlavaForConditionalGeneration.from_hf_model(hf, max_new_tokens=24)
        ans4 = H.answer_pope(m4, proc, qs, "llava", root, load_image=load, reuse_image_prefix=True)
        assert ans4 == ans
        out = tmp_path / "pope_answers.json"
        H.write_pope_answers(str(out), ans)
        assert [json.loads(x) for x in out.read_text().splitlines()] == ans
        sc = H.pope_scores([a["answer"] for a in ans], [q["label"] for q in qs])          # the scorer runs on what came out
        assert sc.TP + sc.FP + sc.TN + sc.FN == 3
    finally:
        ddc.settings.clear()
        ddc.settings.update(saved)


def test_caption_loop_over_the_hip_instructblip_wrapper(built, tmp_path):
    from transformers import (InstructBlipConfig, InstructBlipForConditionalGeneration, InstructBlipQFormerConfig,
                              InstructBlipVisionConfig, LlamaConfig)
    from dropoutdecoding_amd import config as ddc
    from dropoutdecoding_amd import harness as H
    from dropoutdecoding_amd.instructblip import CustomInstructBlipForConditionalGeneration
    from dropoutdecoding_amd.vlm import lm_state_dict_from_hf
    torch.manual_seed(2)
    vc = InstructBlipVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                                  image_size=28, patch_size=14)
    qc = InstructBlipQFormerConfig(vocab_size=100, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                   intermediate_size=64, encoder_hidden_size=32, cross_attention_frequency=1)
    tc = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=2, head_dim=128, max_position_embeddings=512, tie_word_embeddings=False, eos_token_id=EOS)
    cfg = InstructBlipConfig(vision_config=vc.to_dict(), qformer_config=qc.to_dict(), text_config=tc.to_dict(), num_query_tokens=32)
    hf = InstructBlipForConditionalGeneration(cfg).eval()
    for n, p in hf.named_parameters():
        if "language_model" in n:
            p.mul_(2.5)
        if "language_projection" in n:
            p.mul_(8.0)
    _in = getattr(hf, "model", hf)
    (_in if hasattr(_in, "query_tokens") else hf).query_tokens.normal_(0, 1.0, generator=torch.Generator().manual_seed(9))
    sd = {k: bf16_round(v.detach().float().cpu()) for k, v in lm_state_dict_from_hf(hf).items()}
    rc = RefCfg(512, 256, 512, 2, 2, 2, 128, tc.rms_norm_eps, 10000.0)
    saved = dict(ddc.settings)
    try:
        ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
        ddc._module_imported(5217)
        m = CustomInstructBlipForConditionalGeneration.from_hf_model(hf, max_new_tokens=16)
        proc = WordTableProcessor(qformer=True)
        imgs = {f"img{i}.jpg": torch.randn(1, 3, 28, 28, generator=torch.Generator().manual_seed(700 + i)) for i in range(2)}
        items = [(100 + i, f"img{i}.jpg") for i in range(2)]
        log = H.CaptionLog(str(tmp_path / "iblip.json"))
        assert H.caption_images(m, proc, items, "instructblip", log, load_image=imgs.__getitem__, max_new_tokens=9) == 2
        assert proc.calls[0][0] == () and proc.calls[0][1] is True   # keywords: processor(images=..., text=...) (chair_test.py:289-292)
        rows = H.read_caption_log(log.path)
        ref = RefDecoder(FAMILY_IBLIP, rc, sd, [0.3, 0.5, 0.7])
        for (iid, path), row in zip(items, rows):
            inp = proc(images=imgs[path], text=H.CHAIR_PROMPTS["instructblip"]).to("cuda")
            vis = m._visual_embeds(pixel_values=inp["pixel_values"], qformer_input_ids=inp["qformer_input_ids"],
                                   qformer_attention_mask=inp["qformer_attention_mask"])
            emb, _ = m._merge(inp["input_ids"], vis)
            want = ref.generate(emb.cpu(), 0, 32, 9, eos=EOS)
            assert row["image_id"] == iid and proc.ids_of(row["caption"]) == _visible(want)
    finally:
        ddc.settings.clear()
        ddc.settings.update(saved)
