"""ctypes binding of libvneti_hip.so (the C ABI declared in include/vneti.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  `load()` never builds implicitly on a machine without hipcc; use
`__graft_entry__.build()` / `python view_neti_amd/csrc/build.py` to compile.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vneti.h")

# One precision per process: "fp16" (libvneti_hip.so) or "bf16" (libvneti_hip_bf16.so, the same sources built with
# -DVN_BF16) — the reference's optim.mixed_precision branches fp16 / bf16 (training/coach.py:792-802).  Chosen before the
# first call into the library (set_precision, or VNETI_PRECISION in the environment); every 16-bit buffer of the engines
# is allocated as act_dtype().
_PRECISIONS = {"fp16": ("libvneti_hip.so", 0), "bf16": ("libvneti_hip_bf16.so", 1)}
_precision = os.environ.get("VNETI_PRECISION", "fp16")
if _precision not in _PRECISIONS:
    raise RuntimeError(f"VNETI_PRECISION={_precision!r}: expected one of {sorted(_PRECISIONS)}")

_lib = None


def so_path(precision: str = None) -> str:
    # VNETI_LIB_PATH: kernel-development aid (A/B a lab build of the same ABI); the product path is the in-tree library
    return os.environ.get("VNETI_LIB_PATH") or os.path.join(_HERE, "csrc", _PRECISIONS[precision or _precision][0])


SO_PATH = so_path()


def precision() -> str:
    return _precision


def set_precision(p: str) -> None:
    """select the library build; only before the first call into it (a process computes in ONE 16-bit format)"""
    global _precision, SO_PATH
    if p not in _PRECISIONS:
        raise ValueError(f"precision {p!r}: expected one of {sorted(_PRECISIONS)}")
    if _lib is not None and p != _precision:
        raise RuntimeError(f"libvneti is already loaded in {_precision}; {p} needs its own process")
    _precision = p
    SO_PATH = so_path()


def act_dtype():
    """torch dtype of the 16-bit activations / packed weights the loaded library computes in"""
    import torch
    return torch.bfloat16 if _precision == "bf16" else torch.float16

c_ll = C.c_longlong
c_vp = C.c_void_p
c_int = C.c_int
c_f = C.c_float


class TransposeDesc(C.Structure):
    """Mirror of `vneti_transpose_desc` (include/vneti.h)."""

    _fields_ = [("inp", c_vp), ("ld_in", c_ll), ("stride_in", c_ll), ("out", c_vp), ("ld_out", c_ll),
                ("stride_out", c_ll), ("rows", c_int), ("cols", c_int), ("batch", c_int), ("_pad", c_int)]


class GemmDesc(C.Structure):
    """Mirror of `vneti_gemm_desc` (include/vneti.h)."""

    _fields_ = [
        ("A", c_vp), ("B", c_vp), ("C", c_vp),
        ("lda", c_ll), ("ldb", c_ll), ("ldc", c_ll),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("batch", c_int),
        ("strideA", c_ll), ("strideB", c_ll), ("strideC", c_ll),
        ("bias", c_vp),
        ("rowadd", c_vp), ("ld_rowadd", c_ll), ("rows_per_group", c_int),
        ("resid", c_vp), ("ldr", c_ll),
        ("alpha", c_f), ("act", c_int), ("out_f32", c_int),
        ("conv_mode", c_int),
        ("Hi", c_int), ("Wi", c_int), ("Ci", c_int), ("Ho", c_int), ("Wo", c_int),
        ("stride", c_int), ("pad_t", c_int), ("pad_l", c_int), ("ups", c_int),
        ("ldx", c_ll),
        ("tile_hint", c_int),
        ("workspace", c_vp), ("workspace_bytes", c_ll), ("split_k", c_int),
        ("gate_src", c_vp), ("ld_gate", c_ll), ("gate_act", c_int),
        ("C2", c_vp), ("ldc2", c_ll), ("act2", c_int),
        ("gn_sums", c_vp), ("gn_hw", c_int), ("gn_cpg", c_int), ("gn_groups", c_int), ("gn_slots", c_int),
        ("geglu", c_int),
        ("conv_korder", c_int),
    ]


def declared_symbols() -> list[str]:
    """Every function name declared in include/vneti.h (used by the CPU test suite)."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vneti_[a-z0-9_]+)\s*\(", src)))


def load():
    """Load the library once; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{os.path.basename(SO_PATH)} not found at {SO_PATH}: build it with "
            "`python view_neti_amd/csrc/build.py` (hipcc, gfx950). There is no fallback path.")
    lib = C.CDLL(SO_PATH)
    lib.vneti_version.restype = c_int
    lib.vneti_precision.restype = c_int
    if lib.vneti_precision() != _PRECISIONS[_precision][1]:
        raise RuntimeError(f"{SO_PATH} computes in precision {lib.vneti_precision()}, the process asked for {_precision}")
    lib.vneti_last_error.argtypes = [C.c_char_p, C.c_size_t]
    lib.vneti_last_error.restype = c_int
    lib.vneti_groupnorm_ws_floats.restype = c_ll
    lib.vneti_groupnorm_ws_floats.argtypes = [c_int] * 4
    lib.vneti_gemm_f16.argtypes = [C.POINTER(GemmDesc), c_vp]
    _lib = lib
    return lib


def last_error() -> str:
    lib = load()
    buf = C.create_string_buffer(512)
    lib.vneti_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"vneti call failed ({what}) rc={rc}: {last_error()}")


def call(name: str, *args):
    """Call `vneti_<name>` with positional args; pointers are ints/None, scalars by python type.

    Argument conversion is explicit per call-site through the typed helpers below; this generic
    entry is used for the many small elementwise entry points whose signatures are registered
    in SIGNATURES.
    """
    lib = load()
    fn = getattr(lib, "vneti_" + name)
    sig = SIGNATURES.get(name)
    if sig is not None and fn.argtypes is None:
        fn.argtypes = sig
        fn.restype = c_int
    rc = fn(*args)
    check(rc, name)


# argtypes for every int-returning entry point (kept in the same order as include/vneti.h)
SIGNATURES = {
    "im2col3x3_small": [c_vp, c_int, c_ll, c_ll, c_ll, c_ll, c_vp] + [c_int] * 9 + [c_vp],
    "conv3x3_in": [c_vp, c_int, c_ll, c_ll, c_ll, c_ll, c_vp, c_vp, c_vp, c_ll] + [c_int] * 5 + [c_vp, c_int, c_int, c_vp],
    "transpose_f16": [c_vp, c_ll, c_ll, c_vp, c_ll, c_ll, c_int, c_int, c_int, c_vp],
    "transpose_f16_multi": [c_vp, c_int, c_vp],
    "img_resample_coeffs": [c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "img_resample_pass": [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp],
    "img_crop": [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp],
    "img_enhance": [c_vp, c_int, c_int, c_int, c_f, c_vp, c_vp],
    "img_hue": [c_vp, c_int, c_int, c_int, c_vp],
    "img_blur5": [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp],
    "img_affine_nearest": [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp],
    "img_to_f32_chw": [c_vp, c_vp, c_int, c_int, c_vp],
    "groupnorm_fwd_sums": [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                           c_f, c_int, c_vp],
    "groupnorm_fwd": [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                      c_f, c_int, c_vp],
    "groupnorm_fwd_2l": [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                         c_f, c_int, c_vp],
    "groupnorm_bwd_2l": [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp, c_int, c_vp,
                         c_int, c_int, c_int, c_int, c_int, c_vp],
    "groupnorm_bwd": [c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp,
                      c_int, c_int, c_int, c_int, c_int, c_vp],
    "layernorm_fwd": [c_vp, c_int, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_f, c_vp],
    "layernorm_bwd": [c_vp, c_int, c_ll, c_vp, c_int, c_ll, c_vp, c_vp, c_vp, c_vp, c_int, c_ll, c_vp,
                      c_ll, c_vp, c_ll, c_int, c_int, c_vp],
    "attn_fwd": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_int, c_int, c_int, c_int, c_int,
                 c_f, c_int, c_vp],
    "attn_bwd_delta": [c_vp, c_ll, c_vp, c_ll, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "attn_bwd_dq": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_vp, c_ll, c_vp, c_ll,
                    c_int, c_int, c_int, c_int, c_int, c_f, c_int, c_vp],
    "attn_bwd_dkv": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp,
                     c_vp, c_ll, c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_f, c_int, c_vp, c_ll, c_vp],
    "attn_bwd_small": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_ll, c_vp, c_ll,
                       c_int, c_int, c_int, c_int, c_f, c_int, c_vp],
    "softmax_rows_f16": [c_vp, c_ll, c_int, c_int, c_vp],
    "add_f16": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_int, c_int, c_vp],
    "geglu_fwd": [c_vp, c_ll, c_vp, c_ll, c_int, c_int, c_vp],
    "geglu_bwd": [c_vp, c_ll, c_vp, c_ll, c_vp, c_ll, c_int, c_int, c_vp],
    "act_fwd_f16": [c_vp, c_vp, c_ll, c_int, c_vp],
    "act_bwd_f16": [c_vp, c_vp, c_vp, c_ll, c_int, c_vp],
    "timestep_embedding": [c_vp, c_vp, c_int, c_int, c_vp],
    "sum2x2_f16": [c_vp, c_ll, c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp],
    "rng_fill_normal": [c_vp, c_ll, c_vp, C.c_uint, c_vp],
    "rng_fill_randint": [c_vp, c_int, c_int, c_vp, C.c_uint, c_vp],
    "rng_advance": [c_vp, c_vp],
    "sample_add_noise": [c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_f, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int,
                         c_vp],
    "latent_sample": [c_vp, c_ll, c_vp, c_f, c_vp, c_int, c_int, c_int, c_vp],
    "add_noise": [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "cfg_sampler_step": [c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_vp],
    "cfg_sampler_step_table": [c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f, c_vp, c_vp, c_int, c_vp],
    "table_fill_i64": [c_vp, c_int, c_vp, c_vp, c_vp],
    "counter_advance": [c_vp, c_vp],
    "conv1x1_nchw_f32": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_f, c_vp],
    "image_postprocess": [c_vp, c_ll, c_vp, c_ll, c_int, c_vp],
    "mse_loss_grad": [c_vp, c_ll, c_vp, c_vp, c_ll, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "adamw_flat": [c_vp, c_vp, c_vp, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_int, c_vp],
    "adamw_segments": [c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp],
    "nested_dropout_mask": [c_vp, c_int, c_int, c_int, c_f, c_vp, C.c_uint, c_vp],
    "mapper_fwd": [c_vp, c_vp, c_ll, c_vp, c_int, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int,
                   c_int, c_vp, c_vp],
    "mapper_bwd": [c_vp, c_vp, c_ll, c_vp, c_f, c_vp, c_vp, c_vp, c_ll, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                   c_int, c_int, c_int, c_int, c_vp, c_vp],
    "mapper_legacy_input_fwd": [c_vp, c_vp, c_ll, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "mapper_legacy_input_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_int, c_int, c_int, c_int, c_vp],
    "text_embed": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "text_final_fwd": [c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_f, c_int, c_vp, c_vp, c_f, c_int, c_vp, c_vp, c_vp,
                       c_int, c_int, c_int, c_int, c_vp],
    "text_final_bwd": [c_vp, c_vp, c_f, c_vp, c_vp, c_f, c_int, c_vp, c_vp, c_vp, c_f, c_int, c_vp, c_vp, c_vp,
                       c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "cast_f32_f16": [c_vp, c_vp, c_ll, c_vp],
    "comm_unique_id": [c_vp],
    "comm_init": [c_vp, c_int, c_int, C.POINTER(c_vp)],
    "allreduce_flat": [c_vp, c_vp, c_ll, c_vp],
    "comm_destroy": [c_vp],
    "mapper_inputs": [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp],
    "stream_create_cu_mask": [C.POINTER(C.c_uint), c_int, C.POINTER(c_vp)],
    "stream_get_cu_mask": [c_vp, C.POINTER(C.c_uint), c_int],
    "stream_destroy": [c_vp],
}

INT_FUNCS = {"gemm_select_tile": [c_int] * 3, "gemm_select_split": [c_int] * 5 + [c_ll],
             "img_resample_ksize": [c_int] * 3}

LL_FUNCS = {
    "mapper_num_params": [c_int] * 4,
    "mapper_save_floats": [c_int] * 3,
    "mapper_rowgrad_floats": [c_int] * 4,
    "mapper_legacy_input_params": [c_int] * 2,
}


def call_ll(name: str, *args) -> int:
    """call a `long long vneti_<name>(...)` size query; negative means unsupported."""
    fn = getattr(load(), "vneti_" + name)
    fn.argtypes = LL_FUNCS[name]
    fn.restype = c_ll
    return int(fn(*args))
