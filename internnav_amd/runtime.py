"""Runtime plumbing around the HIP library: device check, hipGraph capture of a policy call, per-launch profiling.

The policy engines issue thousands of short kernels per call (16 decoder layers x 11 passes + 2 ViTs); launched one by
one from Python the call is host-bound (~100 us of interpreter + ctypes work per launch). Every engine call is
therefore captured once into a hipGraph (torch.cuda.CUDAGraph is the HIP graph API on ROCm; our kernels are launched on
torch's current stream, which is the capture stream) and replayed: the replay issues the whole launch sequence from the
driver with no Python in the loop. Inputs live in static device buffers that are overwritten before a replay.
"""
from __future__ import annotations

import ctypes as C
import gc
from typing import Callable, Dict

import torch

from . import _lib

PROF_KINDS = ("gemm", "attention", "norm", "elementwise", "gemm_skinny")


class CapacityError(RuntimeError):
    """a request exceeds what an engine was sized for (max_envs / max_seq_len / max_patches): a configuration error that must reach
    the operator - the agent never converts it into a STOP action (capacity overflows used to end episodes silently)."""


def require_gfx950() -> str:
    """Fail loudly unless the HIP library is built AND the current device is an MI355X-class gfx950 GPU (no fallback path)."""
    if not torch.cuda.is_available():
        raise RuntimeError("internnav_amd needs a HIP device (MI355X / gfx950); there is no CPU path")
    buf = C.create_string_buffer(64)
    _lib.check(_lib.lib().ina_device_check(buf, 64), "device_check")
    return buf.value.decode()


class GraphedCall:
    """Capture `fn(**static_inputs)` once and replay it. `fn` must only launch library kernels / torch copies on the current
    stream and write its results into persistent buffers (all engines in this package do)."""

    def __init__(self, fn: Callable, static_inputs: Dict[str, torch.Tensor], warmup: int = 2, workspace_slot: int = 0):
        """workspace_slot: library scratch slot baked into this graph; graphs that may replay concurrently need different slots."""
        self.inputs = static_inputs
        self.fn = fn   # keeps the closure (engine, plan, buffers whose addresses are baked into the graph) alive as long as the graph
        _lib.check(_lib.lib().ina_set_workspace_slot(workspace_slot), "set_workspace_slot")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # also performs the one-time hipFuncSetAttribute calls outside the capture
                fn(**static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # The cyclic garbage collector must not run inside the capture: engines and older GraphedCall objects sit in reference cycles (closures
        # over `self`), and collecting one of them there destroys its hipGraph / events while the stream is capturing - HIP refuses that, the
        # destructor throws and the process aborts ("Fatal Python error: Aborted ... Garbage-collecting", seen once in 430 GPU tests, r06v).
        # torch.cuda.graph() collects BEFORE the capture begins; the capture itself runs with the collector off.
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = fn(**static_inputs)
        finally:
            if was_enabled:
                gc.enable()
        _lib.check(_lib.lib().ina_set_workspace_slot(0), "set_workspace_slot")

    def __call__(self, **new_inputs):
        for k, v in new_inputs.items():
            self.inputs[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.outputs


def prof_enable(on: bool) -> None:
    _lib.check(_lib.lib().ina_prof_enable(1 if on else 0), "prof_enable")


GEMM_SUBS = {18: "gemm_bf16_pp_kernel<256,256,4>", 21: "gemm_bf16_pp_kernel<192,256,4>", 22: "gemm_bf16_glds_kernel<128,128,2,2,1>",
             11: "gemm_bf16_glds_kernel<128,128,2,2,2>", 33: "gemm_bf16_glds_kernel<256,256,4,4,2>", 26: "gemm_bf16_glds_kernel<128,256,2,4,1>", 27: "gemm_bf16_glds_kernel<256,128,4,2,1>", 14: "gemm_bf16_glds_kernel<256,128,4,2,3>", 1: "gemm_bf16_nt_kernel<128,128,64,2,2>",
             2: "gemm_bf16_nt_kernel<64,128,64,2,2>", 3: "gemm_bf16_nt_kernel<64,128,64,1,4>", 39: "gemm_bf16_w4_kernel<256,1>", 40: "gemm_bf16_w4p_kernel<256>", 42: "dit_rowchain_kernel", 43: "gemm_dw_kernel",
             34: "gemm_bf16_rowpanel_kernel<8,4>", 35: "gemm_bf16_rowpanel_kernel<4,4>"}


def prof_read_gemm_kernels() -> Dict[str, dict]:
    """the tiled-GEMM class split by kernel (tile config): summed event time, launches, algorithmic FLOPs of each named kernel."""
    out = {}
    for sub, name in GEMM_SUBS.items():
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().ina_prof_read_sub(0, sub, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "prof_read_sub")
        if n.value:
            out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out


def prof_read() -> Dict[str, dict]:
    """Totals per kernel class since prof_enable(True): summed event-pair time (ms), launches, algorithmic FLOPs and bytes."""
    out = {}
    for k, name in enumerate(PROF_KINDS):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().ina_prof_read(k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "prof_read")
        out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out
