"""ctypes binding of libdropdec.so (include/dropdec.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdropdec.so")

# every symbol include/dropdec.h declares (tests/test_cabi.py checks the library exports all of them)
SYMBOLS = [
    "dd_version", "dd_last_error", "dd_arch",
    "dd_rng_create", "dd_rng_create_philox", "dd_rng_destroy", "dd_rng_seed", "dd_rng_uniform",
    "dd_uncertainty_workspace_bytes", "dd_vision_uncertainty", "dd_overlap_keep", "dd_kl_keep", "dd_sample_masks", "dd_vote",
    "dd_argmax_rows",
    "dd_lm_create", "dd_lm_create_shared", "dd_lm_group_step", "dd_lm_destroy", "dd_lm_device_bytes", "dd_lm_load_tensor", "dd_lm_load_tensor_fp8", "dd_lm_load_synthetic",
    "dd_lm_prefill", "dd_lm_prefill_group", "dd_lm_decode_step_sync", "dd_lm_set_speculation", "dd_lm_spec_stats", "dd_lm_prefill_ensemble", "dd_lm_truncate", "dd_lm_prefill_extend", "dd_lm_decode_step", "dd_lm_step_base", "dd_lm_step_members", "dd_lm_step_commit",
    "dd_lm_xchg_stride", "dd_lm_xchg_export_ids", "dd_lm_xchg_import_ids",
    "dd_lm_xchg_export_winner", "dd_lm_xchg_import_winner", "dd_lm_get", "dd_lm_peek_tokens", "dd_lm_set_next_token", "dd_lm_set_eos", "dd_lm_step_algorithmic_bytes",
    "dd_set_tuning",
    "dd_lm_tp_link", "dd_lm_tp_set_exchange", "dd_lm_tp_prefill", "dd_lm_tp_decode_step",
    "dd_vit_create", "dd_vit_destroy", "dd_vit_load_tensor", "dd_vit_forward",
    "dd_qformer_create", "dd_qformer_destroy", "dd_qformer_load_tensor", "dd_qformer_forward",
]


# include/dropdec_tools.h: libdropdec_tools.so only (bench.py's roofline leg, tools/)
TOOLS_LIB_PATH = os.path.join(_HERE, "libdropdec_tools.so")
TOOLS_SYMBOLS = ["dd_lm_time_sweep", "dd_lm_time_gemv", "dd_tools_last_gemv_kernel", "dd_hbm_read_bench", "dd_tools_set_tuning",
                 "dd_tools_trace_attach", "dd_tools_lds_poison", "dd_tools_scratch_probe", "dd_tools_sample_masks_lanes", "dd_tools_sweep_trace", "dd_tools_attn_trace", "dd_tools_lds_barrier_probe", "dd_tools_lds_overlap_probe", "dd_tools_hold_probe", "dd_tools_pk_probe", "dd_tools_pv_probe", "dd_tools_pkadd_gload_probe", "dd_tools_twist_probe", "dd_tools_barrier_probe", "dd_tools_pk_war_probe",
                 "dd_tools_sampler_dbg_attach", "dd_tools_sampler_dbg_words", "dd_tools_sampler_dbg_launches",
                 "dd_tools_stream_create_cu_mask", "dd_tools_stream_destroy", "dd_tools_cu_probe"]


TP_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)     # int exchange(void* ctx, int rows, void* stream)


class DDError(RuntimeError):
    pass


class VitConfigC(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("patch_size", C.c_int32), ("hidden_size", C.c_int32),
                ("intermediate_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
                ("proj_dim", C.c_int32), ("act", C.c_int32), ("ln_eps", C.c_float), ("flags", C.c_int32), ("reserved", C.c_int32 * 6)]


class QFormerConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("intermediate_size", C.c_int32),
                ("encoder_hidden_size", C.c_int32), ("cross_attention_frequency", C.c_int32), ("num_query_tokens", C.c_int32),
                ("vocab_size", C.c_int32), ("max_position_embeddings", C.c_int32), ("proj_dim", C.c_int32),
                ("max_text_tokens", C.c_int32), ("max_encoder_tokens", C.c_int32), ("ln_eps", C.c_float)]


class LMConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
                ("head_dim", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float),
                ("max_seq", C.c_int32), ("max_visual", C.c_int32), ("k_top", C.c_int32), ("mask_mode", C.c_int32),
                ("vote_on", C.c_int32), ("leak_mask", C.c_int32), ("weight_format", C.c_int32), ("kv_format", C.c_int32),
                ("reserved", C.c_int32 * 3)]


_lib = None
_tools = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raises if it has not been built (python -m dropoutdecoding_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("DD_USE_TOOLS_LIB", "0") not in ("", "0"):
        _lib = load_tools()                # the scripts under tools/: one library for the whole process
        for kv in os.environ.get("DD_TOOLS_TUNE", "").replace(";", ",").split(","):      # A/B runs of bench.py: DD_TOOLS_TUNE="38=1,18=-2"
            if "=" in kv:
                k, v = kv.split("=")
                _lib.dd_tools_set_tuning(int(k), int(v))
        return _lib
    _lib = _open(LIB_PATH)
    return _lib


def load_tools() -> C.CDLL:
    """libdropdec_tools.so: the product objects + the measurement hooks (include/dropdec_tools.h).  A second, independent
    instance of the library: engines created through it must be driven through it (DropoutEngine(lib=...))."""
    global _tools
    if _tools is None:
        _tools = _open(TOOLS_LIB_PATH)
        lib = _tools
        vp = C.c_void_p
        lib.dd_lm_time_sweep.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), vp]
        lib.dd_lm_time_gemv.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double), vp]
        lib.dd_tools_last_gemv_kernel.restype = C.c_char_p
        lib.dd_hbm_read_bench.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float), vp]
        lib.dd_tools_set_tuning.argtypes = [C.c_int, C.c_int]
        lib.dd_tools_trace_attach.argtypes = [vp, vp, C.c_int, vp]
        lib.dd_tools_lds_poison.argtypes = [C.c_int, C.c_int, C.c_int, vp]
        lib.dd_tools_scratch_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_sweep_trace.argtypes = [vp, C.c_int]
        lib.dd_tools_attn_trace.argtypes = [vp, C.c_size_t]
        lib.dd_tools_lds_barrier_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_lds_overlap_probe.argtypes = [C.c_int] * 7 + [vp, vp, vp]
        lib.dd_tools_hold_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_pk_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_pv_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_pkadd_gload_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_twist_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_pk_war_probe.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_barrier_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dd_tools_sampler_dbg_attach.argtypes = [vp]
        lib.dd_tools_stream_create_cu_mask.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(vp)]
        lib.dd_tools_stream_destroy.argtypes = [vp]
        lib.dd_tools_cu_probe.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.dd_tools_sampler_dbg_words.restype = C.c_size_t
        lib.dd_tools_sampler_dbg_words.argtypes = []
        lib.dd_tools_sampler_dbg_launches.restype = C.c_uint
        lib.dd_tools_sampler_dbg_launches.argtypes = []
        lib.dd_tools_sample_masks_lanes.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]
    return _tools


def _open(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise DDError(f"{path} is missing: build it with `python -m dropoutdecoding_amd.build` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # torch must be imported BEFORE the dlopen: torch bundles its own libamdhip64.so.7 / libhsa-runtime64 and the
    # library's DT_NEEDED entries resolve to whichever copy is already in the process (same SONAME).  Loaded first,
    # libdropdec.so would pull in /opt/rocm's runtime and torch would then fail with "no ROCm-capable device"
    # (observed on the MI355X box).  One process, one HIP runtime: torch's.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.cuda.current_device()
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.c_void_p
    lib.dd_version.restype = C.c_int
    lib.dd_last_error.restype = C.c_char_p
    lib.dd_arch.restype = C.c_char_p
    lib.dd_rng_create.argtypes = [C.c_uint32, C.POINTER(vp)]
    lib.dd_rng_create_philox.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(vp)]
    lib.dd_rng_destroy.argtypes = [vp]
    lib.dd_rng_seed.argtypes = [vp, C.c_uint32, vp]
    lib.dd_rng_uniform.argtypes = [vp, vp, C.c_int, vp]
    lib.dd_uncertainty_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.dd_uncertainty_workspace_bytes.restype = C.c_size_t
    lib.dd_vision_uncertainty.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp,
                                          C.c_size_t, vp]
    lib.dd_overlap_keep.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.dd_kl_keep.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.dd_sample_masks.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp,
                                    vp, vp]
    lib.dd_vote.argtypes = [vp, C.c_int, vp, vp]
    lib.dd_argmax_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.dd_lm_create.argtypes = [C.POINTER(LMConfigC), C.POINTER(vp)]
    lib.dd_lm_destroy.argtypes = [vp]
    lib.dd_lm_create_shared.argtypes = [C.POINTER(LMConfigC), vp, C.POINTER(vp)]
    lib.dd_lm_group_step.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(vp), vp]
    lib.dd_lm_device_bytes.argtypes = [vp]
    lib.dd_lm_device_bytes.restype = C.c_size_t
    lib.dd_lm_load_tensor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.dd_lm_load_synthetic.argtypes = [vp, C.c_uint32, C.c_float]
    lib.dd_lm_load_tensor_fp8.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int]
    lib.dd_lm_prefill.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.dd_lm_prefill_group.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    lib.dd_lm_truncate.argtypes = [vp, C.c_int, vp]
    lib.dd_lm_prefill_extend.argtypes = [vp, vp, C.c_int, vp]
    lib.dd_lm_prefill_ensemble.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, vp, vp, vp]
    lib.dd_lm_decode_step.argtypes = [vp, C.POINTER(C.c_double), C.c_int, vp, vp, vp]
    lib.dd_lm_decode_step_sync.argtypes = [vp, C.POINTER(C.c_double), C.c_int, vp, vp, C.POINTER(C.c_int)]
    lib.dd_lm_set_speculation.argtypes = [vp, C.c_int]
    lib.dd_lm_spec_stats.argtypes = [vp, C.POINTER(C.c_int64), C.c_int]
    lib.dd_lm_step_base.argtypes = [vp, C.POINTER(C.c_double), C.c_int, vp, vp, vp]
    lib.dd_lm_step_members.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.dd_lm_step_commit.argtypes = [vp, C.c_int, vp]
    lib.dd_lm_xchg_stride.argtypes = [vp]
    lib.dd_lm_xchg_stride.restype = C.c_size_t
    lib.dd_lm_xchg_export_ids.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.dd_lm_xchg_import_ids.argtypes = [vp, vp, vp]
    lib.dd_lm_xchg_export_winner.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.dd_lm_xchg_import_winner.argtypes = [vp, vp, vp]
    lib.dd_lm_get.argtypes = [vp, C.c_int, vp, C.c_size_t, vp]
    lib.dd_lm_set_next_token.argtypes = [vp, C.c_int32, vp]
    lib.dd_lm_set_eos.argtypes = [vp, C.POINTER(C.c_int32), C.c_int, vp]
    lib.dd_lm_peek_tokens.argtypes = [vp, vp, C.c_int]
    lib.dd_lm_peek_tokens.restype = C.c_int
    lib.dd_lm_step_algorithmic_bytes.argtypes = [vp, C.c_int]
    lib.dd_lm_step_algorithmic_bytes.restype = C.c_double
    lib.dd_set_tuning.argtypes = [C.c_int, C.c_int]
    lib.dd_lm_tp_link.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.dd_lm_tp_set_exchange.argtypes = [vp, vp, C.c_size_t, TP_EXCHANGE_FN, vp]
    lib.dd_lm_tp_prefill.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.dd_lm_tp_decode_step.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(vp), vp]
    lib.dd_vit_create.argtypes = [C.POINTER(VitConfigC), C.POINTER(vp)]
    lib.dd_vit_destroy.argtypes = [vp]
    lib.dd_vit_load_tensor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.dd_vit_forward.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.dd_qformer_create.argtypes = [C.POINTER(QFormerConfigC), C.POINTER(vp)]
    lib.dd_qformer_destroy.argtypes = [vp]
    lib.dd_qformer_load_tensor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.dd_qformer_forward.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    if os.environ.get("DD_NO_GRAPH", "0") not in ("", "0"):
        lib.dd_set_tuning(8, 0)          # launch every decode step kernel by kernel instead of replaying hipGraphs
    return lib


def check(rc: int, what: str = "", lib=None) -> None:
    if rc != 0:
        msg = (lib or load()).dd_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")          # the reference raises ValueError on bad shapes (llava.py:134-138)
        raise DDError(f"{what}: rc={rc}: {msg}")
