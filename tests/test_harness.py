"""Harness adapters: the reference's CLI -> settings mapping, prompts, caption post-filter, jsonl
records and POPE scoring, restated from chair_test/chair_test.py and pope_test/pope_test.py (line refs in harness.py)."""
import json
import os
import random
import types

import pytest

from dropoutdecoding_amd import config as ddc
from dropoutdecoding_amd import harness as H


@pytest.fixture(autouse=True)
def _restore_settings():
    saved = {k: (list(v) if isinstance(v, list) else v) for k, v in ddc.settings.items()}
    yield
    ddc.settings.clear()
    ddc.settings.update(saved)


def test_cli_to_settings_mapping():
    msgs = []
    assert H.apply_cli_settings(1, out=msgs.append)["voting_numbers"] == [0.3]
    assert H.apply_cli_settings(2, out=msgs.append)["voting_numbers"] == [0.5, 0.3]
    assert H.apply_cli_settings(4, out=msgs.append)["voting_numbers"] == [0.1, 0.3, 0.5, 0.7]
    assert msgs == []
    ddc.settings["voting_numbers"] = [0.3, 0.5, 0.7]
    for n in (3, 5, 0):                                    # chair_test.py:171-174: notice, default list stays
        assert H.apply_cli_settings(n, out=msgs.append)["voting_numbers"] == [0.3, 0.5, 0.7]
    assert len(msgs) == 3 and all("unsupport voting number" in m for m in msgs)
    assert H.apply_cli_settings(8)["voting_numbers"] == [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]      # this build's K=8 entry
    s = H.apply_cli_settings(4, use_random=True, avg=True)
    assert s["use_random"] == [True] and s["use_avg"] is True and s is ddc.settings


def test_bool_flags_parse_like_the_reference():
    a = H.build_parser().parse_args(["--model-path", "m", "--coco-data-dir", "c", "--original", "False", "--use-random", "True"])
    assert a.original is True and a.use_random is True and a.avg is False       # argparse type=bool: any non-empty string
    assert a.image_numbers == 500 and a.voting_numbers == 3 and a.sample_save_name == "sample.log" and a.method == "None"


def test_prompts_and_echo_stripping():
    assert H.CHAIR_PROMPTS["llava-1.5"] == "USER: <image>\nDescribe the image. ASSISTANT:"
    assert H.pope_prompt("llava", "Is there a dog?") == "USER: <image>\nIs there a dog? ASSISTANT:"
    assert H.pope_prompt("llava-next", "Is there a dog?") == "[INST] <image>\nIs there a dog?[/INST]"
    assert H.pope_prompt("instructblip", "Is there a dog?") == "Is there a dog?"
    assert H.strip_prompt_echo("llava-1.5", "USER:  \nDescribe the image. ASSISTANT: A cat. ") == "A cat."
    assert H.strip_prompt_echo("llava-next", "[INST]  \nDescribe [/INST] Two [/INST] dogs") == "Two [/INST] dogs"
    assert H.strip_prompt_echo("instructblip", "  a bench \n") == "a bench"
    assert H.strip_prompt_echo("llava-1.5", "no marker") == "no marker"


def test_unk_sentence_filter_and_image_id():
    assert H.filter_unk_sentences("A cat. An unk thing. A dog.") == "A cat. A dog."
    assert H.filter_unk_sentences("A bunk bed. Fine") == " Fine"           # substring match, as in the reference
    assert H.filter_unk_sentences("no period") == "no period"
    assert H.image_id_from_coco_filename("COCO_val2014_000000391895.jpg") == 391895


def test_caption_log_format_and_dedupe(tmp_path):
    p = str(tmp_path / "out" / "m0101.json")
    log = H.CaptionLog(p)
    for i, c in [(7, "a"), (9, "b"), (7, "c"), (7, "d"), (9, "e")]:
        log.append(i, c)
    lines = open(p).read().splitlines()
    assert json.loads(lines[0]) == {"image_id": 7, "caption": "a"} and len(lines) == 5
    rows = H.read_caption_log(p)
    # chair_test.py:383-388 removes one later duplicate per outer row (not all of them)
    ref = [json.loads(x) for x in lines]
    for i in range(len(ref)):
        for j in range(i + 1, len(ref)):
            if i < len(ref) and j < len(ref) and ref[i]["image_id"] == ref[j]["image_id"]:
                ref.pop(j)
                break
    assert rows == ref and [r["caption"] for r in rows] == ["a", "b", "d"]
    assert len(H.read_caption_log(p, dedupe=False)) == 5


def test_sampling_matches_random_sample(tmp_path):
    ids = list(range(1000, 1400))
    f = str(tmp_path / "sample.log")
    got = H.sample_image_ids(ids, 25, seed=42, save_to=f)
    random.seed(42)
    assert got == random.sample(ids, 25)
    assert H.load_sampled_ids(f) == got


def test_pope_answer_rule_and_scores():
    assert H.pope_answer_to_label("No, there is no dog.") == "no"
    assert H.pope_answer_to_label("Yes") == "yes"
    assert H.pope_answer_to_label("There is not a dog. Yes it is") == "no"
    assert H.pope_answer_to_label("Nothing here") == "yes"                # 'Nothing' is not one of the three words
    assert H.pope_answer_to_label("no.") == "no"
    s = H.pope_scores(["Yes", "No", "yes", "no", "Yes", "No"], ["yes", "yes", "no", "no", "yes", "no"])
    assert (s.TP, s.FP, s.TN, s.FN) == (2, 1, 2, 1)
    assert s.accuracy == pytest.approx(4 / 6) and s.precision == pytest.approx(2 / 3) and s.recall == pytest.approx(2 / 3)
    assert s.f1 == pytest.approx(2 / 3) and s.yes_ratio == pytest.approx(0.5)
    assert H.pope_scores(["Yes", "No", "No"], ["yes", "no", "yes"], number=2).accuracy == 1.0


def test_pope_files(tmp_path):
    f = tmp_path / "coco_pope_random.json"
    f.write_text("\n".join(json.dumps(d) for d in [
        {"question_id": 1, "image": "COCO_val2014_000000000042.jpg", "text": "Is there a cat?", "label": "yes"},
        {"question_id": 2, "image": "COCO_val2014_000000000043.jpg", "text": "Is there a car?", "label": "no"}]) + "\n")
    qs = H.parse_pope_file(str(f))
    assert qs[0] == {"image": "COCO_val2014_000000000042.jpg", "text": "Is there a cat?", "label": "yes"}
    out = tmp_path / "ans.json"
    H.write_pope_answers(str(out), [{"question": "Is there a cat?", "answer": "Yes"}])
    assert json.loads(out.read_text().strip()) == {"question": "Is there a cat?", "answer": "Yes"}


class _FakeInputs(dict):
    def to(self, device):
        return self


class _FakeProcessor:
    """Records how it was called (positional (prompt, image) for the LLaVA families, keywords for InstructBLIP)."""
    def __init__(self):
        self.tokenizer = types.SimpleNamespace(eos_token_id=2)
        self.calls = []

    def __call__(self, *a, **k):
        self.calls.append((a, {x: y for x, y in k.items() if x != "return_tensors"}))
        return _FakeInputs(input_ids=[[1, 5]], pixel_values="px")

    def batch_decode(self, ids, skip_special_tokens=True):
        return [ids]


class _FakeModel:
    def __init__(self, texts):
        self.texts, self.kw = list(texts), []

    def generate(self, **kw):
        self.kw.append(kw)
        return self.texts.pop(0)


def test_caption_loop_calls_generate_like_the_reference(tmp_path):
    proc = _FakeProcessor()
    model = _FakeModel(["USER:  \nDescribe the image. ASSISTANT: A cat sits. An unk. On a mat", "x ASSISTANT: Two dogs."])
    log = H.CaptionLog(str(tmp_path / "c.json"))
    n = H.caption_images(model, proc, [(11, "/img/a.jpg"), (12, "/img/b.jpg")], "llava-1.5", log, load_image=lambda p: p,
                         device="cpu")
    assert n == 2
    assert proc.calls[0] == ((H.CHAIR_PROMPTS["llava-1.5"], "/img/a.jpg"), {})
    kw = model.kw[0]
    assert kw["max_new_tokens"] == 512 and kw["num_beams"] == 1 and kw["pad_token_id"] == 2 and kw["pixel_values"] == "px"
    rows = H.read_caption_log(log.path)
    assert rows == [{"image_id": 11, "caption": "A cat sits. On a mat"}, {"image_id": 12, "caption": "Two dogs."}]
    proc2 = _FakeProcessor()
    H.caption_images(_FakeModel(["a bench"]), proc2, [(1, "p")], "instructblip", H.CaptionLog(str(tmp_path / "d.json")),
                     load_image=lambda p: p, device="cpu")
    assert proc2.calls[0] == ((), {"images": "p", "text": "Describe the image."})


class _FakePipeline:
    def __init__(self, texts):
        self.texts, self.batches = list(texts), []

    def run(self, batches, max_new_tokens, eos_token_id=None):
        for b in batches:
            self.batches.append((len(b), max_new_tokens))
            yield [self.texts.pop(0) for _ in b]


def test_caption_loop_through_a_lane_pipeline(tmp_path):
    proc = _FakeProcessor()
    pipe = _FakePipeline([f"USER: x ASSISTANT: caption {i}." for i in range(5)])
    log = H.CaptionLog(str(tmp_path / "g.json"))
    n = H.caption_images(_FakeModel([]), proc, [(100 + i, f"/img/{i}.jpg") for i in range(5)], "llava-1.5", log,
                         load_image=lambda p: p, device="cpu", pipeline=pipe, lanes=2, max_new_tokens=64)
    assert n == 5 and pipe.batches == [(2, 64), (2, 64), (1, 64)]
    assert H.read_caption_log(log.path) == [{"image_id": 100 + i, "caption": f"caption {i}."} for i in range(5)]
    with pytest.raises(ValueError):
        H.caption_images(_FakeModel([]), proc, [(1, "p")], "llava-1.5", log, load_image=lambda p: p, pipeline=pipe, num_beams=3)


def test_pope_loop(tmp_path):
    proc = _FakeProcessor()
    model = _FakeModel(["USER:  \nIs there a cat? ASSISTANT: Yes", "q ASSISTANT: No"])
    rows = H.answer_pope(model, proc, [{"image": "a.jpg", "text": "Is there a cat?"}, {"image": "b.jpg", "text": "Is there a car?"}],
                         "llava", "/coco/val2014", load_image=lambda p: p, device="cpu")
    assert rows == [{"question": "Is there a cat?", "answer": "Yes"}, {"question": "Is there a car?", "answer": "No"}]
    assert model.kw[0]["max_new_tokens"] == 1 and model.kw[0]["num_beams"] == 1
    assert proc.calls[0][1] == {"text": "USER: <image>\nIs there a cat? ASSISTANT:", "images": os.path.join("/coco/val2014", "a.jpg")}
