"""ctypes binding of ``libfishvoc_hip.so`` (include/fishvoc.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises
``ImportError`` loudly, and every op requires CUDA(HIP) tensors.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# FV_LIB_PATH points the loader at an experimental build (A/B kernel variants); default = the in-tree library
LIB_PATH = os.environ.get("FV_LIB_PATH") or os.path.join(CSRC, "libfishvoc_hip.so")

FV_ABI_VERSION = 5
FV_MAX_STAGES = 8
FV_MAX_KERNELS = 8
FV_MAX_DILATIONS = 3

FV_MODEL_HIFIGAN, FV_MODEL_BIGVGAN, FV_MODEL_VOCOS, FV_MODEL_FIREFLY, FV_MODEL_CONVNEXT, FV_MODEL_ISTFT_HEAD = 1, 2, 3, 4, 5, 6
FV_MODEL_LOGMEL = 7
FV_MODEL_REFINEGAN = 8
FV_PRECISION_F32, FV_PRECISION_F16X3 = 0, 1
PRECISIONS = {"f32": FV_PRECISION_F32, "f16x3": FV_PRECISION_F16X3}
FV_CONV_ALGO_AUTO, FV_CONV_ALGO_DIRECT, FV_CONV_ALGO_WINOGRAD = 0, 1, 2
CONV_ALGOS = {"auto": FV_CONV_ALGO_AUTO, "direct": FV_CONV_ALGO_DIRECT, "winograd": FV_CONV_ALGO_WINOGRAD}
FV_ISTFT_SAME, FV_ISTFT_CENTER = 0, 1
FV_ACT_NONE, FV_ACT_SILU, FV_ACT_LEAKY_RELU, FV_ACT_GELU, FV_ACT_TANH, FV_ACT_LOG_CLAMP = 0, 1, 2, 3, 4, 5
FV_POST_ACT_DEFAULT, FV_POST_ACT_IDENTITY = 0, -1   # fv_upsampler_config.post_activation (ABI 5): 0 = reference default SiLU, -1 = nn.Identity

EXPORTS = (
    "fv_create", "fv_load_weight", "fv_finalize", "fv_destroy", "fv_output_length", "fv_output_channels",
    "fv_input_channels", "fv_workspace_bytes", "fv_forward", "fv_conv_create", "fv_conv_output_length",
    "fv_conv_forward", "fv_conv_destroy", "fv_last_error", "fv_abi_version", "fv_last_kernel",
    "fv_profile_begin", "fv_profile_end", "fv_conv_pair_forward", "fv_forward_template",
    "fv_set_precision", "fv_conv_set_precision", "fv_refinegan_noise_elems", "fv_forward_refinegan",
    "fv_set_graph_replay", "fv_get_graph_replay", "fv_reload_env", "fv_set_conv_algorithm", "fv_set_batch_invariant", "fv_conv_set_algorithm",
)

_i32 = ctypes.c_int32


class UpsamplerConfig(ctypes.Structure):
    _fields_ = [
        ("hop_length", _i32), ("num_upsamples", _i32),
        ("upsample_rates", _i32 * FV_MAX_STAGES), ("upsample_kernel_sizes", _i32 * FV_MAX_STAGES),
        ("num_kernels", _i32), ("resblock_kernel_sizes", _i32 * FV_MAX_KERNELS),
        ("resblock_dilation_sizes", (_i32 * FV_MAX_DILATIONS) * FV_MAX_KERNELS),
        ("num_mels", _i32), ("upsample_initial_channel", _i32), ("use_template", _i32),
        ("pre_conv_kernel_size", _i32), ("post_conv_kernel_size", _i32),
        ("post_activation", _i32), ("post_activation_slope", ctypes.c_float),
    ]


class ConvNeXtConfig(ctypes.Structure):
    _fields_ = [("input_channels", _i32), ("num_stages", _i32), ("depths", _i32 * FV_MAX_STAGES),
                ("dims", _i32 * FV_MAX_STAGES), ("kernel_size", _i32)]


class IstftHeadConfig(ctypes.Structure):
    _fields_ = [("dim", _i32), ("n_fft", _i32), ("hop_length", _i32), ("win_length", _i32), ("padding", _i32)]


class LogMelConfig(ctypes.Structure):
    _fields_ = [("sample_rate", _i32), ("n_fft", _i32), ("win_length", _i32), ("hop_length", _i32), ("n_mels", _i32),
                ("f_min", ctypes.c_float), ("f_max", ctypes.c_float)]


class RefineGANConfig(ctypes.Structure):
    _fields_ = [("hop_length", _i32), ("num_stages", _i32), ("downsample_rates", _i32 * FV_MAX_STAGES),
                ("upsample_rates", _i32 * FV_MAX_STAGES), ("num_mels", _i32), ("start_channels", _i32),
                ("leaky_relu_slope", ctypes.c_float)]


class Config(ctypes.Structure):
    _fields_ = [("abi_version", _i32), ("model", _i32), ("ups", UpsamplerConfig), ("backbone", ConvNeXtConfig),
                ("head", IstftHeadConfig), ("mel", LogMelConfig), ("refine", RefineGANConfig)]


class ConvDesc(ctypes.Structure):
    _fields_ = [("transposed", _i32), ("c_in", _i32), ("c_out", _i32), ("kernel_size", _i32), ("dilation", _i32),
                ("padding", _i32), ("stride", _i32), ("pre_act", _i32), ("post_act", _i32),
                ("act_slope", ctypes.c_float)]


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, f"-j{jobs}", "-s"])
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C vocoder_amd/csrc`). "
            "vocoder_amd has no CPU fallback.")
    # torch first: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so), and the process must end up with ONE — loaded after torch, this
    # library's libamdhip64.so.7 dependency resolves to the copy torch already mapped; loaded before it, the system runtime comes in as a second one and
    # its first hipMalloc fails with "no ROCm-capable device is detected" once torch has initialised the device (build() followed by smoke() in one process)
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    fp = ctypes.POINTER(ctypes.c_float)
    L.fv_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    L.fv_create.restype = _i32
    L.fv_load_weight.argtypes = [vp, cp, fp, ctypes.POINTER(ctypes.c_int64), _i32]
    L.fv_load_weight.restype = _i32
    L.fv_finalize.argtypes = [vp]
    L.fv_finalize.restype = _i32
    L.fv_destroy.argtypes = [vp]
    L.fv_destroy.restype = None
    L.fv_output_length.argtypes = [vp, _i32]
    L.fv_output_length.restype = ctypes.c_int64
    L.fv_output_channels.argtypes = [vp]
    L.fv_output_channels.restype = _i32
    L.fv_input_channels.argtypes = [vp]
    L.fv_input_channels.restype = _i32
    L.fv_workspace_bytes.argtypes = [vp, _i32, _i32]
    L.fv_workspace_bytes.restype = ctypes.c_size_t
    L.fv_forward.argtypes = [vp, vp, vp, _i32, _i32, vp, ctypes.c_size_t, vp]
    L.fv_forward.restype = _i32
    L.fv_forward_template.argtypes = [vp, vp, vp, vp, _i32, _i32, vp, ctypes.c_size_t, vp]
    L.fv_forward_template.restype = _i32
    L.fv_refinegan_noise_elems.argtypes = [vp, _i32, _i32]
    L.fv_refinegan_noise_elems.restype = ctypes.c_int64
    L.fv_forward_refinegan.argtypes = [vp, vp, vp, vp, vp, _i32, _i32, vp, ctypes.c_size_t, vp]
    L.fv_forward_refinegan.restype = _i32
    L.fv_conv_create.argtypes = [ctypes.POINTER(ConvDesc), fp, fp, ctypes.POINTER(vp)]
    L.fv_conv_create.restype = _i32
    L.fv_conv_output_length.argtypes = [vp, _i32]
    L.fv_conv_output_length.restype = ctypes.c_int64
    L.fv_conv_forward.argtypes = [vp, vp, vp, vp, _i32, _i32, vp]
    L.fv_conv_forward.restype = _i32
    L.fv_conv_destroy.argtypes = [vp]
    L.fv_conv_destroy.restype = None
    L.fv_conv_pair_forward.argtypes = [vp, vp, vp, vp, _i32, _i32, vp]
    L.fv_conv_pair_forward.restype = _i32
    L.fv_set_precision.argtypes = [vp, _i32]
    L.fv_set_precision.restype = _i32
    L.fv_reload_env.argtypes = []
    L.fv_reload_env.restype = None
    L.fv_set_graph_replay.argtypes = [vp, _i32]
    L.fv_set_graph_replay.restype = _i32
    L.fv_get_graph_replay.argtypes = [vp]
    L.fv_get_graph_replay.restype = _i32
    L.fv_set_conv_algorithm.argtypes = [vp, _i32]
    L.fv_set_conv_algorithm.restype = _i32
    L.fv_set_batch_invariant.argtypes = [vp, _i32]
    L.fv_set_batch_invariant.restype = _i32
    L.fv_conv_set_algorithm.argtypes = [vp, _i32]
    L.fv_conv_set_algorithm.restype = _i32
    L.fv_conv_set_precision.argtypes = [vp, _i32]
    L.fv_conv_set_precision.restype = _i32
    L.fv_profile_begin.argtypes = [vp]
    L.fv_profile_begin.restype = _i32
    L.fv_profile_end.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.fv_profile_end.restype = _i32
    L.fv_last_error.restype = cp
    L.fv_abi_version.restype = _i32
    L.fv_last_kernel.restype = cp
    if L.fv_abi_version() != FV_ABI_VERSION:
        raise ImportError(f"libfishvoc_hip.so ABI {L.fv_abi_version()} != binding {FV_ABI_VERSION}; rebuild")
    _lib = L
    return L


class FishVocError(RuntimeError):
    """An fv_* call returned a non-zero status; carries fv_last_error()."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libfishvoc_hip: {message} (status {status})")
        self.status = status


def check(status: int) -> None:
    if status != 0:
        raise FishVocError(status, lib().fv_last_error().decode("utf-8", "replace"))


def reload_env() -> None:
    """Re-read the experiment knobs (FV_PW ...) that the library caches at first use (fv_reload_env)."""
    lib().fv_reload_env()


def last_kernel() -> str:
    return lib().fv_last_kernel().decode()
