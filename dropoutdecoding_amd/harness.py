"""Harness adapters: the callers and on-disk formats either side of `generate()`.

The reference drives the models from two scripts, `chair_test/chair_test.py` (captions -> CHAIR) and
`pope_test/pope_test.py` (yes/no probing).  This module restates what those scripts do AROUND the model call — CLI ->
`settings` mapping, prompts, prompt-echo stripping, the caption post-filter, the jsonl records, POPE's answer
normalisation and confusion counts — so the same runs can be made on the MI355X box without the reference tree, and so
each format has a test.  The metric code itself (CHAIR's synonym tables, pycocoevalcap; needs nltk/Java) stays external
CPU tooling: it consumes the files written here.

Nothing in here touches the GPU except through `model.generate`.
"""
from __future__ import annotations

import argparse
import json
import os
import random
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence

from .config import VOTING_NUMBERS_K4, VOTING_NUMBERS_K8, settings

# ---- CLI -> settings (chair_test/chair_test.py:161-175) --------------------------------------------------------------
_VOTING = {1: [0.3], 2: [0.5, 0.3], 4: list(VOTING_NUMBERS_K4)}


def voting_numbers_for(n: int) -> Optional[List[float]]:
    """The list `--voting-numbers n` selects, or None where the reference prints its "unsupport voting number" notice and
    leaves the default [0.3, 0.5, 0.7] in force (any n other than 1, 2, 4 — including 3 and 5).  8 is this build's
    addition: the K = 8 list BASELINE configs 3-5 are quoted on has no entry in the reference CLI."""
    if n == 8:
        return list(VOTING_NUMBERS_K8)
    return list(_VOTING[n]) if n in _VOTING else None


def apply_cli_settings(voting_numbers: int = 3, use_random: bool = False, avg: bool = False, out=print) -> Dict:
    if use_random is True:
        settings["use_random"] = [True]                       # chair_test.py:163-164
    v = voting_numbers_for(voting_numbers)
    if v is not None:
        settings["voting_numbers"] = v
    else:
        out("unsupport voting number, this should be from 1 to 5 and will be set to 3 by default")   # chair_test.py:171-174
    settings["use_avg"] = avg                                 # chair_test.py:175
    return settings


# ---- prompts (chair_test.py:30-33, pope_test.py:221-226) -------------------------------------------------------------
CHAIR_PROMPTS = {
    "llava-1.5": "USER: <image>\nDescribe the image. ASSISTANT:",
    "instructblip": "Describe the image.",
    "llava-next": "[INST] <image>\nDescribe the image. [/INST]",
}
_POPE_ALIASES = {"llava": "llava-1.5", "llava-1.5": "llava-1.5", "llava-next": "llava-next", "instructblip": "instructblip"}


def pope_prompt(model: str, text: str) -> str:
    m = _POPE_ALIASES[model]
    if m == "llava-next":
        return f"[INST] <image>\n{text}[/INST]"               # no space before [/INST]: pope_test.py:222
    if m == "llava-1.5":
        return f"USER: <image>\n{text} ASSISTANT:"
    return text


def strip_prompt_echo(model: str, decoded: str) -> str:
    """LLaVA / NeXT return prompt + answer, so the scripts cut at the role marker (chair_test.py:344-349,
    pope_test.py:235-240); InstructBLIP returns the answer only."""
    m = _POPE_ALIASES.get(model, model)
    if m == "llava-1.5":
        return decoded.split("ASSISTANT:", 1)[-1].strip()
    if m == "llava-next":
        return decoded.split("[/INST]", 1)[-1].strip()
    return decoded.strip()


def filter_unk_sentences(text: str) -> str:
    """chair_test.py:351-356: drop every '.'-separated sentence containing "unk", re-join with '.'."""
    return ".".join(s for s in text.split(".") if "unk" not in s)


def image_id_from_coco_filename(name: str) -> int:
    return int(name.split(".jpg")[0][-6:])                    # chair_test.py:275


# ---- caption records: one {"image_id": int, "caption": str} per line (chair_test.py:282,358,368-372) -----------------
class CaptionLog:
    def __init__(self, path: str):
        self.path = path
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)

    def append(self, image_id: int, caption: str) -> None:
        with open(self.path, "a") as f:                       # appended per image, so an interrupted run keeps its rows
            json.dump({"image_id": int(image_id), "caption": caption}, f)
            f.write("\n")


def read_caption_log(path: str, dedupe: bool = True) -> List[Dict]:
    rows = [json.loads(line) for line in open(path) if line.strip()]
    if dedupe:
        # chair_test.py:383-388: for each row, remove the FIRST later row with the same image_id (one per outer row)
        i = 0
        while i < len(rows):
            for j in range(i + 1, len(rows)):
                if rows[i]["image_id"] == rows[j]["image_id"]:
                    rows.pop(j)
                    break
            i += 1
    return rows


def sample_image_ids(image_ids: Sequence[int], n: int, seed: Optional[int], save_to: Optional[str] = None) -> List[int]:
    """chair_test.py:228-243: `random.seed(seed); random.sample(img_ids, n)` over COCO's ids in annotation-file order,
    written one per line to the sample log."""
    if seed is not None:
        random.seed(seed)
    ids = random.sample(list(image_ids), n)
    if save_to:
        with open(save_to, "w") as f:
            for i in ids:
                f.write(f"{i}\n")
    return ids


def load_sampled_ids(path: str) -> List[int]:
    return [int(line.strip()) for line in open(path) if line.strip()]          # chair_test.py:222-226


# ---- the caption loop (chair_test.py:270-372) ---------------------------------------------------------------------------
def _caption_inputs(processor, model_name: str, image, device):
    prompt = CHAIR_PROMPTS[model_name]
    if model_name == "instructblip":
        inputs = processor(images=image, text=prompt, return_tensors="pt")     # chair_test.py:289-292
    else:
        inputs = processor(prompt, image, return_tensors="pt")                  # chair_test.py:294
    return inputs.to(device) if hasattr(inputs, "to") else inputs


def caption_images(model, processor, items: Iterable, model_name: str, log: CaptionLog, load_image: Callable,
                   max_new_tokens: int = 512, num_beams: int = 1, device="cuda", on_caption: Optional[Callable] = None,
                   pipeline=None, lanes: int = 8) -> int:
    """items: (image_id, image_path) pairs.  Builds the inputs the way the reference does for each family, calls
    `model.generate(**inputs, max_new_tokens, num_beams, pad_token_id=eos)`, decodes, strips, filters, appends.

    pipeline: a `dropoutdecoding_amd.vlm.GroupPipeline` over `model` — the images are then captioned `lanes` at a time
    (same captions per image as its own lane's generate(); rows are appended in input order)."""
    n = 0

    def finish(image_id, output_ids):
        nonlocal n
        text = processor.batch_decode(output_ids, skip_special_tokens=True)[0]
        caption = filter_unk_sentences(strip_prompt_echo(model_name, text))
        log.append(image_id, caption)
        if on_caption:
            on_caption(image_id, caption)
        n += 1

    if pipeline is None:
        for image_id, path in items:
            inputs = _caption_inputs(processor, model_name, load_image(path), device)
            finish(image_id, model.generate(**inputs, max_new_tokens=max_new_tokens, num_beams=num_beams,
                                            pad_token_id=processor.tokenizer.eos_token_id))  # chair_test.py:337-342
        return n
    if num_beams != 1:
        raise ValueError("the grouped path is greedy (num_beams=1), like the dropout-decoding path itself")
    items = list(items)
    ids_of = [[i for i, _ in items[b:b + lanes]] for b in range(0, len(items), lanes)]

    def batches():
        for b in range(0, len(items), lanes):
            yield [dict(_caption_inputs(processor, model_name, load_image(path), device)) for _, path in items[b:b + lanes]]
    for ids, outs in zip(ids_of, pipeline.run(batches(), max_new_tokens=max_new_tokens,
                                              eos_token_id=getattr(model, "eos_token_ids", None) or None)):
        for image_id, out in zip(ids, outs):
            finish(image_id, out)
    return n


# ---- POPE (pope_test/pope_test.py) ----------------------------------------------------------------------------------------
def parse_pope_file(path: str) -> List[Dict]:
    return [{"image": d["image"], "text": d["text"], **({"label": d["label"]} if "label" in d else {})}
            for d in (json.loads(line) for line in open(path) if line.strip())]      # pope_test.py:52-64


def pope_answer_to_label(text: str) -> str:
    """pope_test.py:83-94: first sentence, commas removed, split on single spaces; 'No' / 'not' / 'no' => no, else yes."""
    if text.find(".") != -1:
        text = text.split(".")[0]
    words = text.replace(",", "").split(" ")
    return "no" if ("No" in words or "not" in words or "no" in words) else "yes"


@dataclass
class PopeScores:
    TP: int
    FP: int
    TN: int
    FN: int
    accuracy: float
    precision: float
    recall: float
    f1: float
    yes_ratio: float


def pope_scores(answers: Sequence[str], labels: Sequence[str], number: Optional[int] = None) -> PopeScores:
    """pope_test.py:79-145 on raw answer strings and 'yes'/'no' labels (anything but 'no' counts as yes, :96-100)."""
    if number is not None:
        answers, labels = answers[:number], labels[:number]
    pred = [0 if pope_answer_to_label(a) == "no" else 1 for a in answers]
    lab = [0 if l == "no" else 1 for l in labels]
    TP = sum(1 for p, l in zip(pred, lab) if p == 1 and l == 1)
    FP = sum(1 for p, l in zip(pred, lab) if p == 1 and l == 0)
    TN = sum(1 for p, l in zip(pred, lab) if p == 0 and l == 0)
    FN = sum(1 for p, l in zip(pred, lab) if p == 0 and l == 1)
    precision = float(TP) / float(TP + FP)
    recall = float(TP) / float(TP + FN)
    return PopeScores(TP, FP, TN, FN, (TP + TN) / (TP + TN + FP + FN), precision, recall,
                      2 * precision * recall / (precision + recall), pred.count(1) / len(pred))


def write_pope_answers(path: str, rows: Sequence[Dict]) -> None:
    with open(path, "w") as f:                                 # {"question", "answer"} per line, pope_test.py:72-77,243-246
        for r in rows:
            f.write(json.dumps({"question": r["question"], "answer": r["answer"]}) + "\n")


def answer_pope(model, processor, questions: Sequence[Dict], model_name: str, image_root: str, load_image: Callable,
                number: Optional[int] = None, device="cuda", reuse_image_prefix: bool = False) -> List[Dict]:
    """pope_test.py:215-241: one generated token per question.  reuse_image_prefix: POPE asks several questions per image
    back to back; with it the image prefix (vision tower, its prefill, uncertainty) is computed once per image and only
    the question text is prefilled for the others — same answers (settings['reuse_image_prefix'], LLaVA families)."""
    rows = []
    if reuse_image_prefix:
        settings["reuse_image_prefix"] = True
    for q in questions[:number] if number is not None else questions:
        image = load_image(os.path.join(image_root, q["image"]))
        inputs = processor(text=pope_prompt(model_name, q["text"]), images=image, return_tensors="pt")
        inputs = inputs.to(device) if hasattr(inputs, "to") else inputs
        out = model.generate(**inputs, max_new_tokens=1, num_beams=1, pad_token_id=processor.tokenizer.eos_token_id)
        text = processor.batch_decode(out, skip_special_tokens=True)[0]
        rows.append({"question": q["text"], "answer": strip_prompt_echo(model_name, text)})
    return rows


# ---- command line: the reference's flags for the part that runs here --------------------------------------------------------
def _bool(s: str) -> bool:
    return bool(s)           # argparse `type=bool` as in the reference: any non-empty string is True (chair_test.py:466)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="caption COCO images with Dropout Decoding on MI355X; writes the jsonl "
                                            "that chair_test.py feeds to CHAIR")
    p.add_argument("--method", type=str, default="None")
    p.add_argument("--use-prev-sample", type=str, default=None)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--original", type=_bool, default=False)
    p.add_argument("--num-beams", type=int, default=None)
    p.add_argument("--sample-save-name", type=str, default="sample.log")
    p.add_argument("--image-numbers", type=int, default=500)
    p.add_argument("--model", type=str, default="llava-1.5", choices=sorted(CHAIR_PROMPTS))
    p.add_argument("--model-path", type=str, required=True)
    p.add_argument("--coco-data-dir", type=str, required=True)
    p.add_argument("--output-dir", type=str, default="./generated_captions/")
    p.add_argument("--voting-numbers", type=int, default=3)
    p.add_argument("--use-random", type=_bool, default=False)
    p.add_argument("--avg", type=_bool, default=False)
    p.add_argument("--max-new-tokens", type=int, default=512)
    p.add_argument("--lanes", type=int, default=1,
                   help="images captioned concurrently on the GPU (1 = the reference's one-at-a-time loop; up to 16)")
    return p


def main(argv: Optional[Sequence[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    apply_cli_settings(args.voting_numbers, args.use_random, args.avg)
    from datetime import datetime

    from PIL import Image
    from transformers import AutoProcessor
    import models.instructblip as MI
    import models.llava as ML
    import models.llavanext as MN
    cls = {"llava-1.5": ML.CustomLlavaForConditionalGeneration, "llava-next": MN.CustomLlavaNextForConditionalGeneration,
           "instructblip": MI.CustomInstructBlipForConditionalGeneration}[args.model]
    processor = AutoProcessor.from_pretrained(args.model_path, use_fast=False) if args.model != "llava-next" \
        else AutoProcessor.from_pretrained(args.model_path)
    model = cls.from_pretrained(args.model_path)
    model.original = bool(args.original)
    ann = json.load(open(os.path.join(args.coco_data_dir, "annotations", "captions_val2014.json")))
    files = {im["id"]: im["file_name"] for im in ann["images"]}
    ids = load_sampled_ids(args.sample_save_name) if args.use_prev_sample is not None else \
        sample_image_ids(list(files), args.image_numbers, args.seed, args.sample_save_name)
    name = args.method + datetime.now().strftime("%m%d%H%M") + ".json"
    log = CaptionLog(os.path.join(args.output_dir, name))
    items = [(image_id_from_coco_filename(files[i]), os.path.join(args.coco_data_dir, "val2014", files[i])) for i in ids]
    pipe = None
    if args.lanes > 1:
        from .vlm import GroupPipeline
        pipe = GroupPipeline(model, lanes=args.lanes)
    n = caption_images(model, processor, items, args.model, log, lambda p: Image.open(p).convert("RGB"),
                       max_new_tokens=args.max_new_tokens, num_beams=args.num_beams or 1,
                       on_caption=lambda i, c: print(c), pipeline=pipe, lanes=max(1, args.lanes))
    print("the result is saved into", args.output_dir, name, f"({n} captions)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
