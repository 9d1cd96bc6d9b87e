"""The operator API for K and the per-member maximum drop probabilities.

Same dict, same keys, same defaults as the reference's models/config.py:1-4; it is read at every decode step
(reference models/llava.py:340), so late mutation by a CLI is honoured.
"""
settings = {}
settings['voting_numbers'] = [0.3, 0.5, 0.7]
settings['use_avg'] = False          # read by nobody in the reference either (models/llava.py:37-52 is never called)
settings['use_random'] = [False]     # LLaVA-NeXT: True selects "epis_no_overlap" (models/llavanext.py:547-550)
# Switches for code paths the reference carries but never enables.  Absent keys mean "off", so
# a harness that replaces this dict with the reference's three keys keeps working.
#   settings['first_step_ensemble'] = True   -> the `# if True:` toggle at models/llava.py:336-337 (read per generate())
#   settings['mask_method'] = 'epis_no_overlap' -> models/llava.py:663-683 / instructblip.py:486-505 (read at model build)
#   settings['mask_method'] = 'epis_kl'      -> models/instructblip.py:464-485, 559-578 (InstructBLIP; read at model build)
#   settings['use_avg'] = True               -> select_by_average, models/llava.py:37-52 (read at model build)
# Not in the reference at all (its harness re-runs the whole prompt for every question):
#   settings['reuse_image_prefix'] = True    -> consecutive prompts over the SAME image keep the image prefix's K/V,
#                                               uncertainty and top-k ids and prefill only the new text (LLaVA families)

# Which torch generator the dropout draws (models/llava.py:650 torch.rand_like) restate, read when a model is built:
#   settings['rng_stream'] = 'cpu' (default) | 'gpu' -> 'cpu': mt19937, the reference run on CPU (what the golden fixtures hold);
#                                               'gpu': Philox4x32-10 keyed by the module seed, bit-equal to torch.rand(...,
#                                               device="cuda") on ROCm — the reference run on a GPU (tests/test_gpu_dropout_ops.py)

# Storage formats, read when a model is built (from_pretrained / from_hf_model / from_synthetic):
#   settings['kv_cache'] = 'fp16' (default) | 'fp32'  -> KV cache width.  fp16 is what the reference keeps (it loads every model
#                                               with torch_dtype=float16, chair_test.py:189-213); K/V are rounded when they enter
#                                               the cache, the attention arithmetic stays fp32.  Goldens + 24-step oracle runs:
#                                               tokens / masks / votes exact, logits <= 1.7e-4 of max|logit| (tests/test_gpu_kv_fp16.py)
#   settings['weight_format'] = 'auto' (default) | 'fp16' | 'bf16' | 'fp8' -> storage of the LM matrices.  'auto': fp16 when the
#                                               checkpoint's LM tensors are float16 (from_pretrained loads with torch_dtype=float16
#                                               like the reference), else bf16.  'fp16': fp16-native checkpoints stay exact, products
#                                               on the f16 MFMA (vs the fp32 oracle on an fp16-valued checkpoint: logits 2e-6, tokens
#                                               and masks exact); 'bf16' rounds them to bf16 (3 mantissa bits less: logits move by
#                                               5e-3, masks and tokens drift — tests/test_gpu_fp16_weights.py);
#                                               'fp8': OCP e4m3fn + per-row scales (BASELINE config 5), quantised by lm.quantize_fp8

# K = 8 is not reachable from the reference CLI (chair_test.py:163-175); BASELINE configs 3-5 use this list.
VOTING_NUMBERS_K8 = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
VOTING_NUMBERS_K4 = [0.1, 0.3, 0.5, 0.7]   # chair_test.py:170

# The reference seeds torch's GLOBAL generator at import time of each model module (models/llava.py:16-20 -> 24,
# models/llavanext.py:18-21 -> 506, models/instructblip.py:17-21 -> 5217); whichever module is imported last wins
# (under chair_test all three are imported, so 5217 is in force for every model — SURVEY.md A2).  The drop-in
# modules record their seed here at import; engines created afterwards start their mt19937 stream from it.
effective_seed = None


def _module_imported(seed: int) -> None:
    global effective_seed
    effective_seed = seed
