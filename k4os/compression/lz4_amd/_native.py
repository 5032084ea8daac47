"""ctypes binding of libk4lz4.so (include/k4lz4.h).  This is the ONLY compute path of the package:
if the shared object is missing, or no gfx950 device is usable, every operation raises -- there is
no CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported here)."""
from __future__ import annotations

import ctypes as C
import os
import threading

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libk4lz4.so")

K4LZ4_OK = 0
E_HIP, E_ARG, E_NOMEM, E_NO_DEVICE, E_UNSUPPORTED = -1, -2, -3, -4, -5
FLAG_RAW_RETURN = 1
FLAG_PICKLE_WRITER = 2
FLAG_NO_REORDER = 4
FLAG_REORDER = 8
FLAG_NO_SPLIT = 16
FLAG_PARTIAL = 32
FLAG_X32 = 128
FLAG_ALLOW_COPY = 64
FLAG_SEGMENTS = 256

_u8p = C.c_void_p
_BATCH = [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]

# every symbol include/k4lz4.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "k4lz4_version": (C.c_int, []),
    "k4lz4_device_count": (C.c_int, []),
    "k4lz4_recommended_min_batch": (C.c_int64, [C.c_int, C.c_int32, C.c_double]),
    "k4lz4_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "k4lz4_ctx_destroy": (None, [C.c_void_p]),
    "k4lz4_last_error": (C.c_char_p, [C.c_void_p]),
    "k4lz4_ctx_device": (C.c_int, [C.c_void_p]),
    "k4lz4_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "k4lz4_ctx_reserve_hc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int]),
    "k4lz4_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "k4lz4_host_unregister": (C.c_int, [C.c_void_p]),
    "k4lz4_selftest_chains": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint32)]),
    "k4lz4_set_enforce32": (None, [C.c_int]),
    "k4lz4_get_enforce32": (C.c_int, []),
    "k4lz4_compress_bound": (C.c_int, [C.c_int]),
    "k4lz4_last_status": (C.c_int, []),
    "k4lz4_compress_fast": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int]),
    "k4lz4_compress_hc": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, C.c_int]),
    "k4lz4_decompress_safe": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int]),
    "k4lz4_decompress_safe_partial": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int]),
    "k4lz4_encode_batch": (C.c_int, _BATCH + [C.c_int, C.c_int]),
    "k4lz4_decompress_safe_using_dict": (C.c_int, [_u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int]),
    "k4lz4_decode_batch": (C.c_int, _BATCH + [C.c_int]),
    "k4lz4_decode_dict_batch": (C.c_int, _BATCH + [C.c_int, _u8p, C.c_void_p, C.c_void_p]),
    "k4lz4_decode_dict_batch_device": (C.c_int, _BATCH + [C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "k4lz4_encode_batch_device": (C.c_int, _BATCH + [C.c_int, C.c_int, C.c_void_p]),
    "k4lz4_decode_batch_device": (C.c_int, _BATCH + [C.c_int, C.c_void_p]),
    "k4lz4_pickle_bound": (C.c_int, [C.c_int]),
    "k4lz4_unpickle_size": (C.c_int, [_u8p, C.c_int]),
    "k4lz4_pickle_batch": (C.c_int, _BATCH + [C.c_int, C.c_int]),
    "k4lz4_unpickle_batch": (C.c_int, _BATCH + [C.c_int]),
    "k4lz4_pickle_batch_device": (C.c_int, _BATCH + [C.c_int, C.c_int, C.c_void_p]),
    "k4lz4_unpickle_batch_device": (C.c_int, _BATCH + [C.c_int, C.c_void_p]),
    "k4lz4_profile_batch_device": (C.c_int, [C.c_void_p, C.c_int] + _BATCH[1:] + [C.c_void_p, C.c_void_p]),
    "k4lz4_xxh32_batch": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32]),
    "k4lz4_xxh32_batch_device": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]),
    "k4lz4_decode_chain_batch": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "k4lz4_decode_chain_batch_device": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "k4lz4_frame_assemble_device": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, _u8p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_int64, C.c_void_p]),
    "k4lz4_unpickle_sizes_device": (C.c_int, [C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}


class NativeLibraryError(RuntimeError):
    """libk4lz4.so is missing / unusable, or a HIP call failed."""


_lib = None
_lib_lock = threading.Lock()


def _share_hip_runtime():
    """PyTorch-ROCm wheels carry their own libamdhip64.so.  Two HIP runtimes in one process do not both
    see the GPU, and which one a process gets would depend on whether torch or libk4lz4 was imported
    first.  Map torch's copy (if there is one) before libk4lz4.so so that its NEEDED libamdhip64.so.7
    binds to it, and a later `import torch` finds the same runtime.  No torch import, no GPU touch."""
    import importlib.util
    if any("libamdhip64" in line for line in _mapped_objects()):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass        # fall back to the system runtime named by libk4lz4.so's RUNPATH


def _mapped_objects():
    try:
        with open("/proc/self/maps") as f:
            return f.readlines()
    except OSError:
        return []


def load_library(path: str | None = None):
    """dlopen libk4lz4.so and type every declared symbol.  Does not touch the GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise NativeLibraryError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _share_hip_runtime()
        lib = C.CDLL(p)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


class Context:
    """k4lz4_ctx: one per GPU per host thread."""

    def __init__(self, device: int = -1):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.k4lz4_ctx_create(C.byref(h), device)
        if rc != K4LZ4_OK:
            msg = (self.lib.k4lz4_last_error(None) or b"").decode()
            raise NativeLibraryError(f"k4lz4_ctx_create failed ({rc}): {msg}")
        self.handle = h

    @property
    def device(self) -> int:
        return self.lib.k4lz4_ctx_device(self.handle)

    def check(self, rc: int):
        if rc == K4LZ4_OK:
            return
        msg = (self.lib.k4lz4_last_error(self.handle) or b"").decode()
        if rc == E_ARG:
            raise ValueError(msg)
        if rc == E_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == E_NOMEM:
            raise MemoryError(msg)
        raise NativeLibraryError(f"libk4lz4 error {rc}: {msg}")

    def synchronize(self, stream: int = 0):
        self.check(self.lib.k4lz4_synchronize(self.handle, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.k4lz4_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def default_context() -> Context:
    """Per-thread context on the current device (LOCAL_RANK when launched by torch.distributed.run)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        dev = -1
        if "LOCAL_RANK" in os.environ:
            lib = load_library()
            n = lib.k4lz4_device_count()
            if n > 0:
                dev = int(os.environ["LOCAL_RANK"]) % n
        ctx = Context(dev)
        _tls.ctx = ctx
    return ctx


def _check_plain(lib, rc: int):
    if rc == K4LZ4_OK:
        return
    msg = (lib.k4lz4_last_error(None) or b"").decode()
    if rc == E_ARG:
        raise ValueError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise NativeLibraryError(f"libk4lz4 error {rc}: {msg}")


def host_register(arr) -> None:
    """k4lz4_host_register: page-lock a (C-contiguous) numpy buffer that is reused across host-pointer batch calls, so that
    its bytes move between the caller's pages and the GPU without the stop in a staging buffer."""
    lib = load_library()
    _check_plain(lib, lib.k4lz4_host_register(C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)))


def host_unregister(arr) -> None:
    lib = load_library()
    _check_plain(lib, lib.k4lz4_host_unregister(C.c_void_p(arr.ctypes.data)))


def check_last_status(lib):
    """raise if the thread's last LLxx-shaped call failed for an infrastructure reason"""
    rc = lib.k4lz4_last_status()
    if rc == K4LZ4_OK:
        return
    msg = (lib.k4lz4_last_error(None) or b"").decode()
    if rc == E_ARG:
        raise ValueError(msg)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise NativeLibraryError(f"libk4lz4 error {rc}: {msg}")
