"""generate() surface shared by the three drop-in model wrappers (reference models/{llava,llavanext,instructblip}.py).

What the reference keeps on the model object between forwards is kept here under the same
attribute names: image_features, vision_uncert_dict, masked_numbers, start_image_pos, end_image_pos,
start_generation_pos, logits_mask_prob.  The vision front-end (CLIP / Q-Former, projector) and the token
embedding lookup run as stock PyTorch-ROCm modules — SURVEY.md 8(f) rank 1 lists them as the next row to
move onto own kernels; everything from the merged input embeddings onwards runs in libdropdec.so.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .config import settings
from .lm import DropoutEngine, LMConfig


def lm_state_dict_from_hf(hf_model) -> Dict[str, torch.Tensor]:
    """Collect the language model's tensors under LlamaForCausalLM names from either the transformers 4.44 layout
    (`model.language_model` is a *ForCausalLM) or the 5.x layout (`model.model.language_model` + `model.lm_head`)."""
    sd = hf_model.state_dict()
    out = {}
    for k, v in sd.items():
        kk = None
        if k.startswith("language_model.model."):                 # 4.44 wrapper
            kk = "model." + k[len("language_model.model."):]
        elif k.startswith("language_model.lm_head."):
            kk = "lm_head." + k[len("language_model.lm_head."):]
        elif k.startswith("model.language_model."):               # 5.x wrapper
            kk = "model." + k[len("model.language_model."):]
        elif k.startswith("lm_head."):
            kk = k
        elif k.startswith("model.layers.") or k.startswith("model.embed_tokens.") or k.startswith("model.norm."):
            kk = k                                                # a bare *ForCausalLM
        if kk is not None:
            out[kk] = v
    if "lm_head.weight" not in out and "model.embed_tokens.weight" in out:
        out["lm_head.weight"] = out["model.embed_tokens.weight"]   # tied embeddings
    return out


class DropoutVLM:
    """Base of the drop-in wrappers.  One model object = one in-flight sequence (as in the reference)."""

    family: str = ""
    original: bool = False              # True = stock greedy decode (`--original`), no ensemble
    supports_prefix_reuse: bool = False  # visual tokens depend on the image only (LLaVA families; not InstructBLIP's Q-Former)

    def __init__(self, engine: DropoutEngine, embed_tokens: torch.Tensor, image_token_index: int,
                 eos_token_id=None, config=None):
        self.engine = engine
        self.embed_tokens = embed_tokens            # [V, d] on the GPU (lookup only)
        self.image_token_index = image_token_index
        eos = eos_token_id
        self.eos_token_ids = [] if eos is None else (list(eos) if isinstance(eos, (list, tuple)) else [int(eos)])
        self.config = config
        self.device = engine.device
        self.dtype = embed_tokens.dtype
        # reference models/llava.py:55-72
        self.start_generation_pos = 0
        self.start_image_pos: List[int] = []
        self.end_image_pos: List[int] = []
        self.is_first_generation = False
        self.image_features = None
        self.logits_mask_prob: List[float] = []
        self.vision_uncert_dict = None
        self.masked_numbers: List[int] = []
        self.collect_diagnostics = False            # per-step host read-backs (masked_numbers etc.) cost a sync each

    # chair_test / pope_test call these on the returned object
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def cuda(self, *a, **k):
        return self

    def half(self):
        return self

    # ---- family hooks -------------------------------------------------------------------------
    def _visual_embeds(self, **inputs) -> torch.Tensor:
        raise NotImplementedError

    def _merge(self, input_ids: torch.Tensor, visual: torch.Tensor):
        """-> (embeds [T0, d] fp32, span_start). LLaVA families: text embeddings with the visual tokens spliced in
        at the <image> placeholder (reference llava.py:74-153); accepts ONE placeholder (transformers 4.44
        processors) or a contiguous run of L placeholders (5.x processors) — SURVEY.md 8(b) version hazard."""
        ids = input_ids[0]
        L = visual.shape[0]
        pos = torch.nonzero(ids == self.image_token_index).flatten()
        n = pos.numel()
        if n == 0:
            raise ValueError("The input provided to the model are wrong. The number of image tokens is 0 while "
                             "the number of image given to the model is 1.")
        start = int(pos[0])
        contiguous = bool((pos == torch.arange(start, start + n, device=pos.device)).all())
        if not ((n == 1) or (n == L and contiguous)):
            raise ValueError(f"The input provided to the model are wrong. The number of image tokens is {n} while "
                             f"the number of image given to the model is 1 ({L} visual tokens). This prevents "
                             "correct indexing and breaks batch generation.")       # llava.py:134-138
        emb = torch.nn.functional.embedding(ids.clamp(min=0), self.embed_tokens).float()
        merged = torch.cat([emb[:start], visual.float(), emb[start + n:]], dim=0)
        return merged, start

    # ---- the boundary -------------------------------------------------------------------------
    def _visual_embeds_batch(self, inputs_list: List[dict]) -> List[torch.Tensor]:
        """_visual_embeds for several images; families whose tower takes a batch override this (one pass over the tower's weights)."""
        return [self._visual_embeds(**inp) for inp in inputs_list]

    def _prepare(self, input_ids, max_new_tokens, max_length, num_beams, eos_token_id, do_sample, inputs, stream=None,
                 defer_prefill: bool = False, visual: Optional[torch.Tensor] = None):
        """Everything of generate() up to and including the prefill; -> (input_ids on device, n_new, eos ids).
        defer_prefill: leave the LM prefill to the caller, who runs it for several lanes at once (`prefill_lanes`); what it needs
        is kept in self._deferred = (embeds, span start, span length), None when this call prefilled by itself."""
        self._deferred = None
        if input_ids is None or input_ids.shape[0] != 1:
            raise ValueError("Dropout Decoding runs batch size 1 with exactly one image per prompt "
                             "(reference models/llava.py:75-76)")
        if num_beams != 1 or do_sample:
            raise ValueError("the dropout-decoding path is greedy (num_beams=1, do_sample=False)")
        if max_new_tokens is None:
            max_new_tokens = 20 if max_length is None else max(1, max_length - input_ids.shape[1])
        input_ids = input_ids.to(self.device)
        # the `# if True:` toggle of llava.py:336-337; never for the stock greedy (`--original`) path
        first = bool(settings.get("first_step_ensemble", False)) and not self.original
        if (self.supports_prefix_reuse and bool(settings.get("reuse_image_prefix", False)) and not first
                and getattr(self.engine, "tp_rank", None) is None and self._try_reuse_prefix(input_ids, inputs, stream)):
            eos = self.eos_token_ids if eos_token_id is None else (
                list(eos_token_id) if isinstance(eos_token_id, (list, tuple)) else [int(eos_token_id)])
            return input_ids, max_new_tokens, eos
        if visual is None:
            visual = self._visual_embeds(**inputs)
        embeds, start = self._merge(input_ids, visual)
        L = visual.shape[0]
        # reference llava.py:218-226, 75-78, 285-286
        self.is_first_generation = True
        self.logits_mask_prob = []
        self.start_image_pos, self.end_image_pos = [start], [start + L - 1]
        self.start_generation_pos = embeds.shape[0]
        self.masked_numbers = []
        tpr = getattr(self.engine, "tp_rank", None)
        if tpr is not None:
            # a tensor-parallel rank (build_engine(tp=...)): every rank of the group makes this same call with the same arguments
            if first or defer_prefill or stream is not None:
                raise ValueError("a tensor-parallel model decodes one image at a time on the engine's stream (no first-step ensemble, no lanes)")
            tpr.prefill(embeds, start, L)
        elif defer_prefill and not first:
            self._deferred = (embeds, start, L)
        else:
            self.engine.prefill(embeds, start, L, first_step_ensemble=first, stream=stream)
        self._prefix = None
        if self.supports_prefix_reuse and not first:
            ids = input_ids[0]
            n_ph = int((ids == self.image_token_index).sum())
            self._prefix = {"ids": ids[:start + n_ph].clone(), "n_ph": n_ph, "start": start, "L": L,
                            "inputs": {k: v.detach().clone() for k, v in inputs.items() if torch.is_tensor(v)}}
        eos = self.eos_token_ids if eos_token_id is None else (
            list(eos_token_id) if isinstance(eos_token_id, (list, tuple)) else [int(eos_token_id)])
        return input_ids, max_new_tokens, eos

    def _try_reuse_prefix(self, input_ids: torch.Tensor, inputs: dict, stream) -> bool:
        """settings['reuse_image_prefix']: the previous prompt on this object had the same image and the same ids up to
        the image — keep its first start+L positions (K/V, uncertainty, top-k ids) and prefill only the new tail.  What
        the reference recomputes for the shared prefix (pope_test.py:228-232 re-runs the whole prompt for each of the 6
        questions about an image) is identical to what is kept."""
        p = getattr(self, "_prefix", None)
        if p is None:
            return False
        ids = input_ids[0]
        k = p["ids"].numel()
        if ids.numel() <= k or not torch.equal(ids[:k], p["ids"].to(ids.device)):
            return False
        for name, v in p["inputs"].items():
            w = inputs.get(name)
            if not torch.is_tensor(w) or w.shape != v.shape or not torch.equal(w.to(v.device), v):
                return False
        tail = ids[k:]
        if (tail == self.image_token_index).any():
            return False
        emb = torch.nn.functional.embedding(tail, self.embed_tokens).float()
        P = p["start"] + p["L"]
        self.engine.truncate(P, stream=stream)
        self.engine.prefill_extend(emb, stream=stream)
        self.is_first_generation = True
        self.logits_mask_prob = []
        self.start_image_pos, self.end_image_pos = [p["start"]], [P - 1]
        self.start_generation_pos = P + tail.numel()
        self.masked_numbers = []
        return True

    def _finalize(self, input_ids: torch.Tensor, toks: List[int]) -> torch.LongTensor:
        self.is_first_generation = False
        self._publish_prefill_diagnostics()
        new = torch.tensor([toks], dtype=torch.long, device=input_ids.device)
        return self._format_output(input_ids, new)

    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                 max_new_tokens: Optional[int] = None, max_length: Optional[int] = None, num_beams: int = 1,
                 pad_token_id: Optional[int] = None, eos_token_id=None, do_sample: bool = False, **inputs) -> torch.LongTensor:
        input_ids, max_new_tokens, eos = self._prepare(input_ids, max_new_tokens, max_length, num_beams, eos_token_id,
                                                       do_sample, inputs)
        toks = self._decode_loop(max_new_tokens, eos)
        return self._finalize(input_ids, toks)

    # ---- several images at once (lanes) ---------------------------------------------------------
    def spawn_lane(self) -> "DropoutVLM":
        """A further sequence over the SAME weights: a shallow copy of this wrapper with its own engine (KV cache, state,
        rng stream seeded like a fresh process) that borrows this engine's weights.  See generate_group()."""
        import copy
        eng = self.engine
        if getattr(eng, "tp_rank", None) is not None:
            raise ValueError("lanes over a tensor-parallel model are not built (DESIGN.md 7a): one image at a time per group of ranks")
        lane = copy.copy(self)
        lane.engine = DropoutEngine(eng.cfg, family=eng.family, max_seq=eng.max_seq, max_visual=eng.max_visual,
                                    seed=eng.seed, use_random=eng.use_random, iblip_positions=eng.iblip_positions,
                                    weight_format=eng.weight_format, mask_method=eng.mask_method, use_avg=eng.use_avg,
                                    share_weights_with=eng, kv_format=eng.kv_format, rng_stream=eng.rng_stream)
        lane.start_image_pos, lane.end_image_pos, lane.masked_numbers, lane.logits_mask_prob = [], [], [], []
        lane._prefix = None
        return lane


    def _decode_loop(self, n_new: int, eos: List[int], chunk: int = 16) -> List[int]:
        eng = self.engine
        dropout = not self.original
        ks = getattr(self, "kshard", None)             # dist.KShardDecoder: members sharded over ranks
        tpr = getattr(eng, "tp_rank", None)            # dist.TensorParallelRank: the weights sharded over ranks
        if tpr is not None:
            # every rank decodes the same tokens and must issue the same exchanges: fixed chunks, the EOS looked for in the tokens every rank
            # has (the device-side stop makes the steps of a chunk past an EOS no-ops, as in the un-sharded loop)
            eng.set_eos(eos)
            toks = eng.tokens()
            while len(toks) < n_new and not (eos and any(t in eos for t in toks)):
                for _ in range(min(chunk, n_new - len(toks))):
                    tpr.decode_step(dropout=dropout)
                toks = eng.tokens()
            hit = [i for i, t in enumerate(toks) if eos and t in eos]
            if hit:
                toks = toks[:hit[0] + 1]
            return toks[:n_new]
        if ks is None and not self.collect_diagnostics:
            # steps are enqueued without host syncs; the pinned token mirror is watched for EOS (look-ahead steps past
            # it are device-side no-ops: no tokens, no rng draws)
            return eng.generate(n_new, eos=eos, dropout=dropout)
        eng.set_eos(eos)            # K-shard chunks run past an EOS on every rank: the stop is device-side there too
        toks = eng.tokens()
        while len(toks) < n_new and not (eos and toks[-1] in eos):
            if ks is not None and dropout:
                # every rank must issue the same collectives: fixed chunks, EOS checked on the synchronised tokens
                for _ in range(min(chunk, n_new - len(toks))):
                    ks.decode_step()
            else:
                eng.decode_step(dropout=dropout)
                if dropout:
                    self.masked_numbers = eng.last_step()["masked_numbers"].tolist()    # llava.py:338,661-662
            toks = eng.tokens()
            hit = [i for i, t in enumerate(toks) if eos and t in eos]
            if hit:
                toks = toks[:hit[0] + 1]
                break
        return toks[:n_new]

    def _publish_prefill_diagnostics(self) -> None:
        eng = self.engine
        dev = self.device
        u = eng.vision_uncert_dict()
        self.vision_uncert_dict = {k: torch.from_numpy(v).to(dev) for k, v in u.items()}
        vals, ids = eng.topk()
        self.image_features = (torch.from_numpy(vals)[None].to(dev), torch.from_numpy(ids).long()[None].to(dev))

    def _format_output(self, input_ids: torch.Tensor, new: torch.Tensor) -> torch.LongTensor:
        return torch.cat([input_ids, new], dim=1)                   # LLaVA/NeXT: prompt ids ‖ new ids


def build_engine(lm_cfg: LMConfig, family: str, max_visual: int, max_new_tokens: int = 1024, prompt_tokens: int = 256,
                 use_random: bool = False, seed: Optional[int] = None, checkpoint_dtype=None, tp=None, tp_group=None) -> DropoutEngine:
    """The engine behind a drop-in class.  tp = (rank, world): THIS process holds rank `rank` of a tensor-parallel model over `world` GPUs
    (dist.TensorParallelRank: the rank's head / d_ff slices, the two all-gathers per layer over `tp_group` — "nccl" = RCCL over xGMI); the
    engine returned is the rank's, with the driver attached as `engine.tp_rank` — load_state_dict on it takes the FULL state dict."""
    max_seq = max_visual + prompt_tokens + max_new_tokens + 8
    wfmt = settings.get("weight_format", "auto")
    if wfmt == "auto":            # keep the checkpoint's own 16-bit type: fp16 checkpoints (all the reference loads) stay exact
        wfmt = "fp16" if checkpoint_dtype == torch.float16 else "bf16"
    kw = dict(use_random=use_random, mask_method=settings.get("mask_method", "epis"), use_avg=bool(settings.get("use_avg", False)),
              weight_format=wfmt, kv_format=settings.get("kv_cache", "fp16"), rng_stream=settings.get("rng_stream", "cpu"))
    if tp is not None:
        from .dist import TensorParallelRank
        rank, world = int(tp[0]), int(tp[1])
        if not 0 <= rank < world:
            raise ValueError(f"tp=(rank, world): got {tp}")
        tpr = TensorParallelRank(lm_cfg, rank, world, group=tp_group, family=family, max_seq=max_seq, max_visual=max_visual, seed=seed, **kw)
        eng = tpr.engine
        eng.tp_rank = tpr
        eng.load_state_dict_full = eng.load_state_dict
        eng.load_state_dict = tpr.load_state_dict             # the wrappers hand over the FULL dict: the rank keeps its slices
        return eng
    eng = DropoutEngine(lm_cfg, family=family, max_seq=max_seq, max_visual=max_visual, seed=seed, **kw)
    eng.tp_rank = None
    return eng


_NON_VISUAL_KEYS = ("input_ids", "attention_mask", "max_new_tokens", "max_length", "eos_token_id")


@torch.no_grad()
def _batched_visuals(model: "DropoutVLM", inputs: List[dict]) -> List[Optional[torch.Tensor]]:
    """The visual tokens of several generate() inputs in one tower call where the family supports it (None entries: let
    _prepare compute them — prefix reuse may make the tower unnecessary)."""
    if not inputs or bool(settings.get("reuse_image_prefix", False)):
        return [None] * len(inputs)
    rest = [{k: v for k, v in kw.items() if k not in _NON_VISUAL_KEYS} for kw in inputs]
    return list(model._visual_embeds_batch(rest))


@torch.no_grad()
def prefill_lanes(models: List["DropoutVLM"], stream=None, chunk: int = 16) -> None:
    """The deferred LM prefills of `models` (see _prepare(defer_prefill=True)), `chunk` sequences per pass over the weights
    (dd_lm_prefill_group: the prompts run through the layers as one matrix; every lane ends up bit for bit as its own prefill
    would leave it; 21 instead of 28 ms per 608-position prompt at LLaVA-1.5-7B shapes)."""
    from .lm import prefill_group
    todo = [m for m in models if getattr(m, "_deferred", None) is not None]
    for i in range(0, len(todo), chunk):
        part = todo[i:i + chunk]
        prefill_group([m.engine for m in part], [m._deferred[0] for m in part], [(m._deferred[1], m._deferred[2]) for m in part],
                      stream=stream)
        for m in part:
            m._deferred = None


@torch.no_grad()
def generate_group(models: List[DropoutVLM], inputs: List[dict], max_new_tokens: Optional[int] = None, eos_token_id=None,
                   num_beams: int = 1, do_sample: bool = False, pad_token_id: Optional[int] = None) -> List[torch.LongTensor]:
    """`generate()` for up to 64 images at once: models[i] (a wrapper and its spawn_lane() copies) decodes inputs[i].

    Each image is decoded exactly as `models[i].generate(**inputs[i])` would (same tokens, masks, logits; its own rng
    stream, like one process of the reference's sharded 500-image runs), but all sequences advance together and their
    un-masked passes share one sweep over the weights (EngineGroup) — the throughput mode for CHAIR-style jobs."""
    from .lm import EngineGroup
    if len(models) != len(inputs) or not 1 <= len(models) <= 64:
        raise ValueError("generate_group: one model lane per input, 1..64 of them")
    prepared = []
    visuals = _batched_visuals(models[0], inputs)
    for m, kw, vis in zip(models, inputs, visuals):
        kw = dict(kw)
        kw.pop("attention_mask", None)
        ids = kw.pop("input_ids", None)
        mnt = kw.pop("max_new_tokens", max_new_tokens)
        prepared.append(m._prepare(ids, mnt, kw.pop("max_length", None), num_beams, kw.pop("eos_token_id", eos_token_id),
                                   do_sample, kw, defer_prefill=True, visual=vis))
    prefill_lanes(models)
    n_new = {p[1] for p in prepared}
    eos = prepared[0][2]
    if len(n_new) != 1 or any(p[2] != eos for p in prepared):
        raise ValueError("generate_group: all images of a group share max_new_tokens and eos_token_id")
    dropout = {not m.original for m in models}
    if len(dropout) != 1:
        raise ValueError("generate_group: `original` must be the same for all lanes")
    toks = EngineGroup([m.engine for m in models]).generate(n_new.pop(), eos=eos, dropout=dropout.pop())
    return [m._finalize(p[0], t) for m, p, t in zip(models, prepared, toks)]


class GroupPipeline:
    """Caption a long list of images in batches of up to 64: while one set of lanes decodes, the vision tower and the
    prefill of the NEXT batch are enqueued on a second stream (the decode step is HBM-bound, the prefill MFMA-bound, so
    they overlap) — the 500-image CHAIR job of the reference's SLURM launchers on one GPU.

    Each image is still decoded exactly as `generate()` would decode it on a lane of its own (see generate_group)."""

    def __init__(self, model: DropoutVLM, lanes: int = 8):
        if not 1 <= lanes <= 64:
            raise ValueError("1..64 lanes per set")
        self.device = model.device
        # one host thread, one decode stream: the library's graph capture is not safe against a second capturing thread
        self.sets = [[model] + [model.spawn_lane() for _ in range(lanes - 1)], [model.spawn_lane() for _ in range(lanes)]]
        self.pre_stream = torch.cuda.Stream(device=model.device)
        self.prefill_chunk = 16          # sequences per LM prefill pass (dd_lm_prefill_group); 1: one prefill per image
        self.tower_chunk = None          # images per vision-tower call (None: as many as a prefill pass takes); LLaVA-NeXT: 6 = two calls of
                                         # three 5-tile images (a tower call takes 16 tiles) while the LM prefill takes 4 prompts per pass

    def _stage(self, lanes, batch, kw):
        """-> per-image closures that each enqueue one image's front-end + prefill on the second stream"""
        state = {"prepared": [], "todo": list(zip(lanes, batch)), "event": None, "lanes": lanes[:len(batch)], "staged": [], "vis": []}

        def unit() -> bool:
            if not state["todo"]:
                return False
            if not state["vis"]:                       # the vision tower for the next chunk of images, one call
                with torch.cuda.stream(self.pre_stream):
                    nxt = [inp for _, inp in state["todo"][:(self.tower_chunk or self.prefill_chunk)]]
                    state["vis"] = _batched_visuals(state["todo"][0][0], nxt)
                return True
            m, inp = state["todo"].pop(0)
            vis = state["vis"].pop(0)
            inp = dict(inp)
            inp.pop("attention_mask", None)
            with torch.cuda.stream(self.pre_stream):
                state["prepared"].append(m._prepare(inp.pop("input_ids", None), inp.pop("max_new_tokens", kw["max_new_tokens"]),
                                                    inp.pop("max_length", None), 1, inp.pop("eos_token_id", kw["eos_token_id"]),
                                                    False, inp, stream=self.pre_stream, defer_prefill=True, visual=vis))
                state["staged"].append(m)
                if not state["todo"] or len(state["staged"]) >= self.prefill_chunk:
                    prefill_lanes(state["staged"], stream=self.pre_stream, chunk=self.prefill_chunk)   # one pass over the weights
                    state["staged"].clear()
                if not state["todo"]:
                    state["event"] = torch.cuda.Event()
                    state["event"].record(self.pre_stream)
            return True
        state["unit"] = unit
        return state

    @torch.no_grad()
    def run(self, batches, max_new_tokens: int, eos_token_id=None):
        """batches: iterable of lists of generate() keyword dicts (up to `lanes` each); yields the list of output id tensors
        of each batch, in order."""
        from .lm import EngineGroup
        kw = {"max_new_tokens": max_new_tokens, "eos_token_id": eos_token_id}
        it = iter(batches)
        first = next(it, None)
        while first is not None and len(first) == 0:           # empty batches (a trailing [] of a batching generator) are skipped
            first = next(it, None)
        if first is not None and len(first) > len(self.sets[0]):
            raise ValueError(f"a batch of {len(first)} images for {len(self.sets[0])} lanes")
        cur = self._stage(self.sets[0], first, kw) if first is not None else None
        k = 0
        while cur is not None:
            while cur["unit"]():                       # whatever the previous decode loop did not get to
                pass
            lanes, prepared = cur["lanes"], cur["prepared"]
            lanes[0].engine.torch_stream.wait_event(cur["event"])
            k += 1
            following = {"state": None, "asked": False}

            def fetch(k=k, following=following):
                # the next batch is pulled from the caller's iterator (image loading, preprocessing: host time) only once this
                # batch's first decode steps are queued on the GPU, not between two batches with the GPU idle
                following["asked"] = True
                nb = next(it, None)
                while nb is not None and len(nb) == 0:
                    nb = next(it, None)
                if nb is not None and len(nb) > len(self.sets[0]):
                    raise ValueError(f"a batch of {len(nb)} images for {len(self.sets[0])} lanes")
                following["state"] = self._stage(self.sets[k % 2], nb, kw) if nb is not None else None

            def idle(following=following, fetch=fetch) -> bool:
                if not following["asked"]:
                    fetch()
                    return True
                return following["state"]["unit"]() if following["state"] is not None else False

            dropout = not lanes[0].original
            toks = EngineGroup([m.engine for m in lanes]).generate(prepared[0][1], eos=prepared[0][2], dropout=dropout, idle=idle)
            if not following["asked"]:
                fetch()
            nxt = following["state"]
            yield [m._finalize(p[0], t) for m, p, t in zip(lanes, prepared, toks)]
            cur = nxt
