"""ctypes binding of libvidmem.so (C ABI: include/vidmem.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible, every entry point
raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvidmem.so")
# the developer build (make -C csrc dev: the same sources with -DVM_DEV_SWITCHES, where VIDMEM_* variables select kernel
# variants).  Never loaded by the package itself: tools/ and the variant tests call use_dev_library() before anything else.
DEV_LIB_PATH = os.path.join(_HERE, "libvidmem_dev.so")

VM_OK = 0
VM_ERR_INVALID, VM_ERR_HIP, VM_ERR_NOMEM, VM_ERR_UNSUPPORTED, VM_ERR_NO_DEVICE = -1, -2, -3, -4, -5
VM_F16, VM_BF16, VM_F32 = 0, 1, 2
VM_ACT_GELU, VM_ACT_QUICK_GELU = 0, 1
VM_LAYOUT_CHW, VM_LAYOUT_PATCHES = 0, 1
VM_SCORE_RAW, VM_SCORE_UNIT_INTERVAL = 0, 1
VM_FLAG_CERTIFIED, VM_FLAG_GAP, VM_FLAG_OVERFLOW = 0, 1, 3       # vm_topk_flag: why a query went to the exhaustive redo
VM_ENC_OPT_SCHEDULE, VM_ENC_OPT_MICRO_BATCH, VM_ENC_OPT_LAST_LAYER = 0, 1, 2
VM_SCHED_AUTO, VM_SCHED_ONE_STREAM, VM_SCHED_TWO_STREAMS = 0, 1, 2
SCHEDULES = {"auto": VM_SCHED_AUTO, "one_stream": VM_SCHED_ONE_STREAM, "two_streams": VM_SCHED_TWO_STREAMS}
DTYPES = {"f16": VM_F16, "bf16": VM_BF16}

# every symbol include/vidmem.h declares (tests/test_abi.py checks the export table against the header)
SYMBOLS = [
    "vm_init", "vm_destroy", "vm_last_error", "vm_abi_version", "vm_preprocess",
    "vm_encoder_create", "vm_encoder_destroy", "vm_encoder_set_option", "vm_encoder_get_option",
    "vm_encoder_tokens", "vm_encoder_patch_k", "vm_encoder_out_dim",
    "vm_encode_workspace_bytes", "vm_encode_micro_batch", "vm_encode",
    "vm_memory_create", "vm_memory_destroy", "vm_memory_append", "vm_memory_size", "vm_memory_capacity",
    "vm_memory_dim", "vm_memory_reset", "vm_memory_sync", "vm_memory_rows",
    "vm_topk_workspace_bytes", "vm_topk_cosine", "vm_topk_redo_workspace_bytes", "vm_topk_redo_flagged",
    "vm_topk_exact_workspace_bytes", "vm_topk_cosine_exact",
    "vm_cosine_exact", "vm_topk_select", "vm_topk_merge", "vm_profile_enable", "vm_profile_read", "vm_profile_mask", "vm_probe_mfma",
]
PROF_CATS = ["preprocess", "gemm_patch", "gemm_qkv", "gemm_act", "gemm_resid", "attention", "layernorm", "pool",
             "append", "topk_scan", "topk_finalize", "topk_exact", "topk_merge", "gemm_cls"]


class VidmemError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libvidmem error {code}: {message}")
        self.code = code


class EncoderDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("image", "patch", "hidden", "layers", "heads", "mlp", "act", "pre_ln", "patch_bias", "proj_dim",
                 "dtype")] + [("ln_eps", C.c_float)]


_lib: Optional[C.CDLL] = None


def use_dev_library() -> None:
    """Developer tools only: load libvidmem_dev.so instead of the release library.  Must run before the first call into
    the library in this process."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_dev_library() after the library was loaded")
    if not os.path.exists(DEV_LIB_PATH):
        raise FileNotFoundError(f"{DEV_LIB_PATH} is missing: make -C csrc dev")
    LIB_PATH = DEV_LIB_PATH


def lib() -> C.CDLL:
    """Load libvidmem.so; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). vidmem has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
    sig = {
        "vm_init": (i32, [i32, C.POINTER(vp)]),
        "vm_destroy": (None, [vp]),
        "vm_last_error": (C.c_char_p, [vp]),
        "vm_abi_version": (i32, []),
        "vm_preprocess": (i32, [vp, vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, i32, i32,
                                i32, i32, vp, vp]),
        "vm_encoder_create": (i32, [vp, C.POINTER(EncoderDesc), C.POINTER(vp), i32, C.POINTER(vp)]),
        "vm_encoder_destroy": (None, [vp]),
        "vm_encoder_set_option": (i32, [vp, i32, i32]),
        "vm_encoder_get_option": (i32, [vp, i32]),
        "vm_encoder_tokens": (i32, [vp]),
        "vm_encoder_patch_k": (i32, [vp]),
        "vm_encoder_out_dim": (i32, [vp]),
        "vm_encode_workspace_bytes": (sz, [vp, i32]),
        "vm_encode_micro_batch": (i32, [vp, i32]),
        "vm_encode": (i32, [vp, vp, i32, vp, i32, vp, sz, vp]),
        "vm_memory_create": (i32, [vp, i64, i32, i32, i32, C.POINTER(vp)]),
        "vm_memory_destroy": (None, [vp]),
        "vm_memory_append": (i32, [vp, vp, i32, C.POINTER(i64), vp]),
        "vm_memory_size": (i64, [vp]),
        "vm_memory_capacity": (i64, [vp]),
        "vm_memory_dim": (i32, [vp]),
        "vm_memory_reset": (i32, [vp, vp]),
        "vm_memory_sync": (i64, [vp, vp]),
        "vm_memory_rows": (vp, [vp]),
        "vm_topk_workspace_bytes": (sz, [vp, i32, i32]),
        "vm_topk_cosine": (i32, [vp, vp, i32, i32, i32, f64, i32, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
        "vm_topk_redo_workspace_bytes": (sz, [vp, i32, i32]),
        "vm_topk_redo_flagged": (i32, [vp, vp, i32, i32, i32, f64, i32, i64, i64, vp, vp, vp, vp, sz, vp]),
        "vm_topk_exact_workspace_bytes": (sz, [vp, i32, i32]),
        "vm_topk_cosine_exact": (i32, [vp, vp, i32, i32, i32, f64, i32, i64, i64, vp, vp, vp, sz, vp]),
        "vm_cosine_exact": (i32, [vp, vp, i32, vp, i64, i32, i32, vp, vp]),
        "vm_topk_select": (i32, [vp, vp, i32, i64, vp, i32, i64, vp, vp, vp]),
        "vm_topk_merge": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
        "vm_profile_enable": (i32, [vp, i32]),
        "vm_profile_read": (i32, [vp, C.POINTER(f64), C.POINTER(i64)]),
        "vm_profile_mask": (i32, [vp, C.c_uint32]),
        "vm_probe_mfma": (i32, [vp, i32, i32, f64, C.POINTER(f64), vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class Context:
    """One vm_ctx per process and device."""
    _cache = {}

    def __init__(self, device: int = 0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.vm_init(int(device), C.byref(h))
        if rc != VM_OK:
            msg = self.L.vm_last_error(None)
            raise VidmemError(rc, (msg or b"vm_init failed").decode())
        self.handle = h
        self.device = int(device)

    @classmethod
    def get(cls, device: int = 0) -> "Context":
        if device not in cls._cache:
            cls._cache[device] = cls(device)
        return cls._cache[device]

    def profile_enable(self, max_events: int) -> None:
        self.check(self.L.vm_profile_enable(self.handle, int(max_events)))

    def profile_mask(self, categories=None) -> None:
        """Record events only for these category names (None = all)."""
        mask = 0xFFFFFFFF if categories is None else sum(1 << PROF_CATS.index(c) for c in categories)
        self.check(self.L.vm_profile_mask(self.handle, mask))

    def profile_read(self) -> dict:
        """{category: (total_ms, launches)} since the last read (synchronises the device)."""
        n = len(PROF_CATS)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self.check(self.L.vm_profile_read(self.handle, ms, cnt))
        return {PROF_CATS[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}

    def probe_mfma(self, variant: int, seconds: float = 1.0, zero_operands: bool = False) -> float:
        """TFLOP/s this GPU sustains on a synthetic 16-bit MFMA loop (include/vidmem.h vm_probe_mfma)."""
        out = C.c_double()
        self.check(self.L.vm_probe_mfma(self.handle, int(variant), 1 if zero_operands else 0, float(seconds),
                                        C.byref(out), current_stream_ptr()))
        return float(out.value)

    def check(self, rc: int) -> None:
        if rc != VM_OK:
            msg = self.L.vm_last_error(self.handle)
            raise VidmemError(rc, (msg or b"").decode())


def current_stream_ptr() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
