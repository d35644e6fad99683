"""Phase ranges for rocprofv3 traces (`--marker-trace`): the regions the reference scopes with
`torch.autograd.profiler.record_function` — "train_step-N" (fairseq_cli/train.py:225-227), "forward" / "backward"
(tasks/fairseq_task.py:439-444), "multiply-grads" / "clip-grads" / "optimizer" (trainer.py:601-627) — emitted as roctx ranges around
the same regions of this build, plus "reduce-grads" (the wait for the overlapped bucket collectives, trainer.py:588-589).

Off unless CST_ROCTX=1 (then `scope()` is a shared no-op context: nothing is loaded, nothing is pushed).  With CST_ROCTX=1 a missing
libroctx64.so is an error, not a silent no-op.  Ranges are host-side markers; the kernels enqueued inside a range are attributed to it
by the tracer through the launch correlation ids."""
import contextlib
import ctypes
import os

_ON = os.environ.get("CST_ROCTX") == "1"
_NULL = contextlib.nullcontext()
_lib = None


def enabled():
    return _ON


def _load():
    global _lib
    if _lib is None:
        last = None
        # rocprofv3 (rocprofiler-sdk) intercepts ITS roctx library; the legacy libroctx64 (roctracer) is the fallback for older tools
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError as e:
                last = e
        if _lib is None:
            raise RuntimeError("CST_ROCTX=1 but no roctx library could be loaded: %s" % last)
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.argtypes = []
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name.encode()

    def __enter__(self):
        _load().roctxRangePushA(self.name)
        return self

    def __exit__(self, *exc):
        _load().roctxRangePop()
        return False


def scope(name):
    """`with scope("forward"): ...` — a roctx range when CST_ROCTX=1, otherwise the shared no-op context."""
    return _Range(name) if _ON else _NULL
