"""Timing events recorded straight through the HIP runtime on the launch stream."""
from __future__ import annotations

import ctypes as C


class HipEvent:
    """A timing event on the launch stream (the roofline sample brackets single kernel launches inside the timed region;
    torch.cuda.Event is the same call with default flags).  LQER_BENCH_EVENT_FLAGS selects the creation flags: default
    0x20000000 = hipEventDisableSystemFence (the event's release stays at device scope - nothing on the host reads what the
    bracketed kernel wrote; a pair then costs 3.4 us instead of 4.6-5.2: NOTEBOOK §8.2), 0 = hipEventDefault.  The bench line
    carries the flags, the calibrated pair overhead and a second, event-free timed region beside the instrumented one."""
    _hip = None

    def __init__(self, flags):
        if HipEvent._hip is None:
            h = C.CDLL("libamdhip64.so")
            h.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            h.hipEventDestroy.argtypes = [C.c_void_p]
            HipEvent._hip = h
        self.h = C.c_void_p()
        rc = HipEvent._hip.hipEventCreateWithFlags(C.byref(self.h), flags)
        assert rc == 0, f"hipEventCreateWithFlags: {rc}"

    def record(self, stream):
        rc = HipEvent._hip.hipEventRecord(self.h, stream)
        if rc:
            raise RuntimeError(f"hipEventRecord: {rc}")

    def elapsed_time(self, other):
        ms = C.c_float()
        rc = HipEvent._hip.hipEventElapsedTime(C.byref(ms), self.h, other.h)
        assert rc == 0, f"hipEventElapsedTime: {rc}"
        return ms.value

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and HipEvent._hip is not None:
            try:
                HipEvent._hip.hipEventDestroy(h)
            except Exception:  # (interpreter shutdown)
                pass
