"""Tracing hooks of the training loop (counterpart of ldm/experiment.py:230-232,243: jax.profiler.StepTraceAnnotation
per step and clu.periodic_actions.Profile when config.training.profile is set).

Ranges are roctx markers (librocprofiler-sdk-roctx / libroctx64 through ctypes, else torch.cuda.nvtx, which maps onto
roctx on ROCm builds, else no-ops): `rocprofv3 --marker-trace --kernel-trace -- python -m ldm.main ...` shows the
steps and their phases next to the kernels.  Pushing a range costs ~1 us on the host and nothing on the device.
"""
import contextlib
import ctypes

_push = _pop = None


def _bind():
    global _push, _pop
    if _push is not None:
        return
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _push, _pop = (lambda s: lib.roctxRangePushA(s.encode())), lib.roctxRangePop
            return
        except (OSError, AttributeError):
            continue
    try:
        import torch
        _push, _pop = torch.cuda.nvtx.range_push, torch.cuda.nvtx.range_pop
        _push("mulan")
        _pop()
    except Exception:
        _push, _pop = (lambda s: 0), (lambda: 0)


@contextlib.contextmanager
def trace_range(name):
    _bind()
    _push(name)
    try:
        yield
    finally:
        _pop()


class Profile:
    """clu.periodic_actions.Profile look-alike (num_profile_steps = 5, first profile after `first_profile` steps):
    while active the train step runs eagerly (no HIP-graph replay) and emits one roctx range per phase (forward /
    backward / all-reduce / optimizer): the ranges are what `rocprofv3 --marker-trace` shows.  torch.cuda.profiler
    start / stop are called around the window as well; on ROCm builds of torch they may be no-ops, nothing relies on
    them."""

    def __init__(self, num_profile_steps=5, first_profile=10):
        self.n, self.first = num_profile_steps, first_profile
        self.active = False
        self._left = 0
        self._done = False

    def __call__(self, step):
        import torch
        if self.active:
            self._left -= 1
            if self._left <= 0:
                self.active = False
                self._done = True
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                    torch.cuda.profiler.stop()
        elif not self._done and step >= self.first:
            self.active, self._left = True, self.n
            if torch.cuda.is_available():
                torch.cuda.profiler.start()

    def phase(self, name):
        return trace_range(name) if self.active else contextlib.nullcontext()
