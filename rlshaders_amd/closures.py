"""Host-side mirror of the reference's closure classes over the C ABI.

Each class keeps the reference's name and verbs -- ``GgxSampler.evalSample / evalBrdf / evalPdf``
(src/rlGgx.h:97-127), ``DisneySampler`` + ``setSampleType`` (src/rlDisney.cpp:109-152,194-197),
``NDProfile.getRadius / getPdf / evalProfile`` (src/rlSss.h:49-55), ``SssSampler.getProbeRay``
(src/rlSss.h:487) -- but works on a *batch* of shading points: every argument that was one
``AtVector`` / ``AtColor`` / ``float`` per call in the reference is a planar SoA torch tensor on
the GPU (``[3, n]`` / ``[n]`` float32), or a Python float / 3-tuple for a parameter that is
uniform over the batch.  torch is plumbing only (device memory and streams); every number is
computed by the hand-written HIP kernels behind ``include/rlshaders_amd.h``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Union

import torch

from . import _capi as capi
from ._capi import (RLS_KERNEL_NDF, RLS_KERNEL_VNDF, RLS_RAY_DIFFUSE, RLS_RAY_GLOSSY, RlsError, check)

Scalar = Union[float, torch.Tensor]
Color = Union[Sequence[float], torch.Tensor]


class Context:
    """One rls_context: a device plus the stream launches go to (torch's current stream by default)."""

    def __init__(self, device: Optional[int] = None, use_torch_stream: bool = True):
        self.lib = capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("rlshaders_amd.Context: no HIP device visible to torch; the closures run on the GPU only")
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = C.c_void_p()
        check(self.lib.rls_context_create(self.device, C.byref(h)))
        self.handle = h
        self.torch_device = torch.device("cuda", self.device)
        if use_torch_stream:
            self.use_stream(torch.cuda.current_stream(self.torch_device))

    def use_stream(self, stream: Optional["torch.cuda.Stream"]) -> None:
        """stream = a torch.cuda.Stream (its handle may be 0 = the null stream), or None for the
        context's private stream.

        On the private stream (``use_stream(None)`` / ``use_torch_stream=False``) the launches are NOT ordered with
        torch's own work: torch fills and frees tensors on its current stream, the kernels read and write them on a
        non-blocking stream.  The caller then orders the two -- ``torch.cuda.synchronize()`` (or an event) after the
        inputs are produced and ``ctx.synchronize()`` before the outputs are read or any tensor a launch used (inputs,
        and outputs this class allocated) is dropped: torch's caching allocator recycles a freed block for later work
        on ITS stream without knowing that a kernel on the private stream may still be using it.  (Registering the
        tensors with the allocator -- ``Tensor.record_stream`` on a wrapper of the private stream -- is not an option:
        the allocator would record its events on that stream when the tensors die, possibly after ``close()`` has
        destroyed it.)"""
        if stream is None:
            check(self.lib.rls_context_use_own_stream(self.handle))
        else:
            check(self.lib.rls_context_set_stream(self.handle, C.c_void_p(stream.cuda_stream)))

    def set_math_mode(self, fast: bool) -> None:
        """False: RLS_MATH_EXACT (default, bit-faithful to the CPU closures); True: RLS_MATH_FAST."""
        check(self.lib.rls_context_set_math_mode(self.handle, 1 if fast else 0))

    def synchronize(self) -> None:
        check(self.lib.rls_context_synchronize(self.handle))

    def capture(self) -> "GraphCapture":
        """``with ctx.capture() as g: <closure calls>`` records the calls into a launch graph instead of
        running them; afterwards ``g.launch()`` replays them in one go (rls_graph_*).  The outputs must be
        passed in (``out=...``): tensors allocated while recording would be freed before the replay."""
        return GraphCapture(self)

    def timer_start(self) -> None:
        check(self.lib.rls_timer_start(self.handle))

    def timer_stop(self) -> None:
        check(self.lib.rls_timer_stop(self.handle))

    def timer_elapsed_ms(self) -> float:
        ms = C.c_float()
        check(self.lib.rls_timer_elapsed_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    # ---- in-kernel clock stamps (rls_diag_clock_stamps_*; a measurement aid, not part of the closure surface) ----------
    def clock_stamps_begin(self) -> None:
        """From here to clock_stamps_end() the kernels of BASELINE configs 2-5 run as their stamped instantiation."""
        if not hasattr(self.lib, "rls_diag_clock_stamps_begin"):
            raise RlsError(5, "this library was built with RLS_DIAGNOSTICS=0: it carries no clock stamps")
        check(self.lib.rls_diag_clock_stamps_begin(self.handle))

    def clock_stamps_end(self) -> None:
        check(self.lib.rls_diag_clock_stamps_end(self.handle))

    def clock_stamps_read(self):
        """uint64 [workgroups, 4]: {memtime at entry, at exit, memrealtime at entry, at exit} of the workgroups of the last
        stamped launch (synchronises; slots of workgroups that did not run are dropped)."""
        import numpy as np
        count = C.c_int64()
        check(self.lib.rls_diag_clock_stamps_read(self.handle, 0, None, C.byref(count)))
        raw = np.zeros((count.value, 4), dtype=np.uint64)
        check(self.lib.rls_diag_clock_stamps_read(self.handle, count.value, raw.ctypes.data_as(C.c_void_p), C.byref(count)))
        return raw[raw[:, 3] != 0]

    @staticmethod
    def clock_from_stamps(stamps, realtime_hz: float = 100e6) -> dict:
        """Effective shader clock of the stamped launch: per workgroup d(memtime) / d(memrealtime) x 100 MHz
        (MI355X_MICROARCH.md, DVFS item 6); the median over the workgroups is the figure, p05 / p95 its spread.
        `span_ms` = last exit - first entry on the 100 MHz counter: the launch's duration as the stamps see it (compare
        with the HIP-event time of the same launch to check the counter's rate)."""
        import numpy as np
        if len(stamps) == 0:
            raise RuntimeError("no stamps: no stamped kernel ran since clock_stamps_begin()")
        dt = (stamps[:, 1] - stamps[:, 0]).astype(np.float64)
        dr = (stamps[:, 3] - stamps[:, 2]).astype(np.float64)
        ok = dr > 0
        ghz = dt[ok] / dr[ok] * realtime_hz / 1e9
        # the realtime counter ticks every 10 ns: a workgroup that lives 15 us is resolved to 0.07 %
        return {"effective_clock_ghz": round(float(np.median(ghz)), 4),
                "p05_ghz": round(float(np.percentile(ghz, 5)), 4), "p95_ghz": round(float(np.percentile(ghz, 95)), 4),
                "workgroups": int(ok.sum()),
                "median_workgroup_us": round(float(np.median(dr[ok])) / realtime_hz * 1e6, 3),
                "median_workgroup_cycles": int(np.median(dt[ok])),
                "span_ms": round(float(stamps[:, 3].max() - stamps[:, 2].min()) / realtime_hz * 1e3, 5)}

    def device_info(self) -> dict:
        cus = C.c_int()
        total = C.c_size_t()
        free = C.c_size_t()
        name = C.create_string_buffer(64)
        check(self.lib.rls_device_info(self.handle, C.byref(cus), C.byref(total), C.byref(free), name, 64))
        return {"compute_units": cus.value, "hbm_total": total.value, "hbm_free": free.value,
                "arch": name.value.decode()}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.rls_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers --------------------------------------------------------------------------------
    def empty(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.float32, device=self.torch_device)


class Arena:
    """All the planes of a batch in ONE device allocation -- and, with ``candidates`` > 1, in the fastest of that
    many equally sized blocks (each is timed with an arithmetic-free copy of the kernels' plane pattern,
    rls_probe_block; see "Placement" in DESIGN.md).  The memory is a torch tensor; ``plane()`` / ``planes(rows)``
    hand out consecutive [n] / [rows, n] views of it."""

    def __init__(self, ctx: Context, n: int, planes: int, candidates: int = 1):
        self.ctx, self.n, self.count = ctx, int(n), int(planes)
        self.stride = ((self.n * 4 + 255) // 256 * 256) // 4          # planes start on 256-byte boundaries
        self.bytes = self.count * self.stride * 4
        self.used = 0
        self.probe_gbs, self.chosen_gbs = [], None
        self.block = None
        if candidates <= 0:        # comparison mode: every request is its own allocation, as without an arena
            return
        free, _ = torch.cuda.mem_get_info(ctx.torch_device)
        candidates = max(1, min(int(candidates), int(0.8 * free // max(self.bytes, 1))))
        blocks = []
        for _ in range(candidates):
            try:
                b = torch.empty(self.count * self.stride, dtype=torch.float32, device=ctx.torch_device)
            except torch.OutOfMemoryError:
                break
            blocks.append(b)
            if candidates > 1:
                torch.cuda.synchronize(ctx.torch_device)
                g = C.c_float()
                check(ctx.lib.rls_probe_block(ctx.handle, C.c_void_p(b.data_ptr()), self.bytes, C.byref(g)))
                self.probe_gbs.append(float(g.value))
        if not blocks:
            raise torch.OutOfMemoryError(f"Arena: cannot allocate {self.bytes} bytes")
        best = max(range(len(blocks)), key=lambda k: self.probe_gbs[k]) if self.probe_gbs else 0
        self.block = blocks[best]
        self.chosen_gbs = self.probe_gbs[best] if self.probe_gbs else None
        del blocks
        torch.cuda.empty_cache()
        self.block.zero_()

    def planes(self, rows: int) -> torch.Tensor:
        if self.block is None:
            self.used += rows
            return torch.empty(rows, self.n, dtype=torch.float32, device=self.ctx.torch_device)
        if self.used + rows > self.count:
            raise RuntimeError(f"Arena: {self.count} planes, {self.used} handed out, {rows} more requested")
        t = self.block[self.used * self.stride:].as_strided((rows, self.n), (self.stride, 1))
        self.used += rows
        return t

    def plane(self) -> torch.Tensor:
        return self.planes(1)[0]

    def info(self) -> dict:
        if self.block is None:
            return {"arena": False, "planes": self.used}
        return {"arena": True, "bytes": self.bytes, "planes": self.count, "candidates_probed": len(self.probe_gbs),
                "probe_gb_per_s": [round(g, 1) for g in self.probe_gbs], "chosen_gb_per_s": self.chosen_gbs}


class _DevicePlane:
    """device planes the library owns, as torch sees them (zero-copy, through __cuda_array_interface__)"""

    def __init__(self, ptr: int, count: int, rows: int = 0, row_stride_bytes: int = 0):
        if rows:
            self.__cuda_array_interface__ = {"shape": (rows, count), "strides": (row_stride_bytes, 4), "typestr": "<f4",
                                             "data": (ptr, False), "version": 2}
        else:
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class ChunkPlanes:
    """The device planes of one pipeline chunk: ``p[k]`` is plane k as a [count] tensor (None where the host plane is not
    streamed), ``p.rows(k, 3)`` planes k .. k+2 as one [3, count] tensor -- what the closure classes take for a vec3 /
    colour.  Views of the slot's memory: fill and read them with closure calls on the slot's Context only (torch's own
    operations run on torch's stream, which is not ordered with the slot's)."""

    def __init__(self, ptrs, live, count: int, device):
        self._ptrs, self._live, self._count, self._device = ptrs, live, int(count), device

    def __len__(self):
        return len(self._live)

    def ptr(self, k: int) -> Optional[int]:
        """device address of plane k (None where the host plane is not streamed): for callers that talk to the C ABI
        directly and do not want a tensor object per plane and chunk"""
        return int(self._ptrs[k]) if self._live[k] else None

    def device_ptr(self, k: int) -> int:
        """device address of the slot's plane k whether or not a host plane is streamed to / from it (an output the kernel
        must write but the host does not want stays in the slot)"""
        return int(self._ptrs[k])

    def device_rows(self, k: int, rows: int = 1, dtype=None):
        """tensor view of the slot's planes k .. k + rows - 1 whether or not host planes are streamed to / from them: [count]
        for one plane, [rows, count] for equally spaced ones; ``dtype=torch.int32`` reinterprets the bits (a material-id plane)"""
        step = self._ptrs[k + 1] - self._ptrs[k] if rows > 1 else 0
        if any(self._ptrs[k + j] - self._ptrs[k] != j * step for j in range(rows)):
            raise ValueError("ChunkPlanes.device_rows: planes are not equally spaced")
        t = torch.as_tensor(_DevicePlane(self._ptrs[k], self._count, rows if rows > 1 else 0, step), device=self._device)
        return t if dtype is None else t.view(dtype)

    def __getitem__(self, k: int):
        if not self._live[k]:
            return None
        return torch.as_tensor(_DevicePlane(self._ptrs[k], self._count), device=self._device)

    def rows(self, k: int, rows: int = 3):
        if not all(self._live[k:k + rows]):
            return None
        step = self._ptrs[k + 1] - self._ptrs[k] if rows > 1 else 4 * self._count
        if any(self._ptrs[k + j] - self._ptrs[k] != j * step for j in range(rows)):
            raise ValueError("ChunkPlanes.rows: planes are not equally spaced")
        return torch.as_tensor(_DevicePlane(self._ptrs[k], self._count, rows, step), device=self._device)


class Pipeline:
    """Host-resident batches (rls_pipeline_*): a batch whose planes live in page-locked HOST memory goes through the GPU in
    chunks -- upload, closure kernels, download -- on ``depth`` streams, so that the copies of neighbouring chunks overlap
    the kernels and each other.  What an Arnold-side stub does with the shading points its CPU render threads gathered
    (the reference evaluates per hit on those threads, src/rlGgx.cpp:248-261).

    ``run(n, host_in, host_out, launch)``: ``host_in`` / ``host_out`` are lists of pinned float32 CPU tensors of n
    elements (``torch.empty(n, pin_memory=True)``) or None for a plane that is not streamed; ``launch(slot, first, count,
    dev_in, dev_out)`` is called once per chunk with a Context for the chunk's stream and the chunk's device planes
    (ChunkPlanes) and makes the closure calls -- on ``slot``, writing into ``dev_out`` through the verbs' ``out=``."""

    def __init__(self, ctx: Context, chunk_points: int, in_planes: int, out_planes: int, depth: int = 3):
        self.ctx, self.chunk_points = ctx, int(chunk_points)
        self.in_planes, self.out_planes, self.depth = int(in_planes), int(out_planes), int(depth)
        h = C.c_void_p()
        check(ctx.lib.rls_pipeline_create(ctx.handle, self.chunk_points, self.in_planes, self.out_planes, self.depth, C.byref(h)))
        self.handle = h
        self._slots = {}

    def _slot(self, handle: int) -> Context:
        s = self._slots.get(handle)
        if s is None:
            s = Context.__new__(Context)            # a view of the slot's rls_context: borrowed, never destroyed from here
            s.lib, s.device, s.torch_device = self.ctx.lib, self.ctx.device, self.ctx.torch_device
            s.handle = C.c_void_p(handle)
            s.close = lambda: None
            self._slots[handle] = s
        return s

    def run(self, n: int, host_in, host_out, launch) -> None:
        if len(host_in) != self.in_planes or len(host_out) != self.out_planes:
            raise ValueError(f"Pipeline.run: expected {self.in_planes} input and {self.out_planes} output planes")
        for what, planes in (("host_in", host_in), ("host_out", host_out)):
            for k, t in enumerate(planes):
                if t is None:
                    continue
                if t.dtype != torch.float32 or t.is_cuda or t.dim() != 1 or t.shape[0] != n or not t.is_contiguous():
                    raise TypeError(f"{what}[{k}]: expected a contiguous float32 CPU tensor of {n} elements")
                if not t.is_pinned():
                    raise TypeError(f"{what}[{k}]: host planes must be page-locked (torch.empty(n, pin_memory=True))")
        hin = (C.c_void_p * max(self.in_planes, 1))(*[t.data_ptr() if t is not None else None for t in host_in])
        hout = (C.c_void_p * max(self.out_planes, 1))(*[t.data_ptr() if t is not None else None for t in host_out])
        err = []

        def _cb(_user, slot, first, count, din, dout):
            try:
                launch(self._slot(slot), int(first), int(count),
                       ChunkPlanes([din[k] for k in range(self.in_planes)], [t is not None for t in host_in], count,
                                   self.ctx.torch_device),
                       ChunkPlanes([dout[k] for k in range(self.out_planes)], [t is not None for t in host_out], count,
                                   self.ctx.torch_device))
                return 0
            except RlsError as e:       # never unwind through the C frames
                err.append(e)
                return e.status
            except Exception as e:
                err.append(e)
                return 1

        cb = capi.PipelineLaunchFn(_cb)
        st = self.ctx.lib.rls_pipeline_run(self.handle, int(n), hin, hout, C.cast(cb, C.c_void_p), None)
        if err:
            raise err[0]
        check(st)

    def copy_rates(self, nbytes: int = 1 << 28) -> dict:
        """pinned-memory copy rates of this box in GB/s (rls_measure_copy_rates)"""
        r = (C.c_float * 3)()
        check(self.ctx.lib.rls_measure_copy_rates(self.ctx.handle, int(nbytes), r))
        return {"h2d": float(r[0]), "d2h": float(r[1]), "both": float(r[2])}

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.ctx.lib.rls_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GraphCapture:
    """A recorded sequence of closure launches (rls_graph)."""

    def __init__(self, ctx: Context):
        self.ctx, self.handle = ctx, None

    def __enter__(self):
        check(self.ctx.lib.rls_graph_begin_capture(self.ctx.handle))
        return self

    def __exit__(self, exc_type, exc, tb):
        h = C.c_void_p()
        st = self.ctx.lib.rls_graph_end_capture(self.ctx.handle, C.byref(h))
        if exc_type is None:
            check(st)
            self.handle = h
        elif st == 0:
            self.ctx.lib.rls_graph_destroy(h)
        return False

    def launch(self) -> None:
        if self.handle is None:
            raise RuntimeError("GraphCapture.launch: nothing was recorded")
        check(self.ctx.lib.rls_graph_launch(self.ctx.handle, self.handle))

    def close(self) -> None:
        if self.handle is not None:
            self.ctx.lib.rls_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _chk(t: torch.Tensor, n: Optional[int], rows: Optional[int], what: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a torch tensor, got {type(t).__name__}")
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{what}: expected a float32 CUDA tensor, got {t.dtype} on {t.device}")
    if rows is None:
        if t.dim() != 1 or not t.is_contiguous():
            raise ValueError(f"{what}: expected a contiguous [n] tensor, got shape {tuple(t.shape)}")
    else:
        if t.dim() != 2 or t.shape[0] != rows or t.stride(1) != 1:
            raise ValueError(f"{what}: expected a [{rows}, n] tensor with unit inner stride, got {tuple(t.shape)}")
    if n is not None and t.shape[-1] != n:
        raise ValueError(f"{what}: batch size {t.shape[-1]} != {n}")


def cvec3(t: torch.Tensor, n: int, what: str) -> capi.CVec3:
    _chk(t, n, 3, what)
    return capi.CVec3(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())


def vec3(t: torch.Tensor, n: int, what: str) -> capi.Vec3:
    _chk(t, n, 3, what)
    return capi.Vec3(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())


def rgb(t: torch.Tensor, n: int, what: str) -> capi.Rgb:
    _chk(t, n, 3, what)
    return capi.Rgb(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())


def plane(t: torch.Tensor, n: int, what: str) -> int:
    _chk(t, n, None, what)
    return t.data_ptr()


def material_index(materials, n: int):
    """``materials`` = (ids, count): ids an int32 / uint32 CUDA tensor [n] (the node instance of every shading point), count
    the length of the parameter columns -> (rls_material_index, the length a tensor parameter must then have).  With it,
    tensor parameters are per-MATERIAL columns [count] / [3, count] instead of per-point planes (include/rlshaders_amd.h,
    rls_material_index)."""
    if materials is None:
        return capi.MaterialIndex(None, 0), n
    ids, count = materials
    count = int(count)
    if not isinstance(ids, torch.Tensor) or ids.dtype not in (torch.int32, torch.uint32) or ids.shape != (n,) or \
            not ids.is_cuda or not ids.is_contiguous():
        raise TypeError("materials: expected (ids, count) with ids a contiguous int32 CUDA tensor of shape [n]")
    if count < 1:
        raise ValueError("materials: count must be at least 1")
    return capi.MaterialIndex(ids.data_ptr(), count), count


def param(v: Scalar, n: int, what: str) -> capi.Param:
    if isinstance(v, torch.Tensor):
        return capi.Param(plane(v, n, what), 0.0)
    return capi.Param(None, float(v))


def param_rgb(v: Color, n: int, what: str) -> capi.ParamRgb:
    if isinstance(v, torch.Tensor):
        _chk(v, n, 3, what)
        return capi.ParamRgb(v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), 0.0, 0.0, 0.0)
    r, g, b = (float(x) for x in v)
    return capi.ParamRgb(None, None, None, r, g, b)


# ================================================================================================
class GgxSampler:
    """Batched ``rls::GgxSampler`` (= ``GgxSamplerT<VNDFKernel>``, src/rlGgx.h:92-375).

    Constructor arguments follow src/rlGgx.h:130-131: ``(sg, specColor, ior, roughness,
    anisotropic=0)`` where ``sg`` contributes ``wo = -sg->Rd``, ``N = sg->Nf`` and the tangent ``T``
    of the local frame; ``exiting`` (uint8 [n], optional) marks points where
    ``dot(sg->N, sg->Rd) >= AI_EPSILON`` (src/rlGgx.h:137).
    """

    def __init__(self, ctx: Context, wo, N, T, specColor: Color = (1.0, 1.0, 1.0), ior: Scalar = 1.0,
                 roughness: Scalar = 0.0, anisotropic: Scalar = 0.0, exiting: Optional[torch.Tensor] = None, materials=None):
        self.ctx = ctx
        self.n = int(wo.shape[-1])
        n = self.n
        self._keep = (wo, N, T, specColor, ior, roughness, anisotropic, exiting, materials)
        c = capi.GgxClosure()
        c.wo, c.N, c.T = cvec3(wo, n, "wo"), cvec3(N, n, "N"), cvec3(T, n, "T")
        if exiting is not None:
            if exiting.dtype != torch.uint8 or exiting.shape != (n,) or not exiting.is_cuda:
                raise TypeError("exiting: expected a uint8 CUDA tensor of shape [n]")
            c.exiting = exiting.data_ptr()
        c.materials, pn = material_index(materials, n)      # pn: the length of a tensor parameter (n, or the column length)
        self.pn = pn
        c.KsColor = param_rgb(specColor, pn, "specColor")
        c.ior = param(ior, pn, "ior")
        c.specularRoughness = param(roughness, pn, "roughness")
        c.anisotropic = param(anisotropic, pn, "anisotropic")
        self.c = c

    # -- the callback triple ---------------------------------------------------------------------
    def evalSample(self, rx, ry, out_wi=None, out_fresnel=None):
        """src/rlGgx.h:97-107 -> (wi [3,n], fresnel [n])."""
        n, ctx = self.n, self.ctx
        wi = ctx.empty(3, n) if out_wi is None else out_wi
        F = ctx.empty(n) if out_fresnel is None else out_fresnel
        check(ctx.lib.rls_ggx_sample(ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(ry, n, "ry"),
                                     vec3(wi, n, "wi"), plane(F, n, "fresnel")))
        return wi, F

    def evalBrdf(self, indir, out=None):
        """src/rlGgx.h:110-119 -> f [3,n] (BRDF x signed cosine)."""
        n, ctx = self.n, self.ctx
        f = ctx.empty(3, n) if out is None else out
        check(ctx.lib.rls_ggx_eval(ctx.handle, n, C.byref(self.c), cvec3(indir, n, "indir"), rgb(f, n, "f")))
        return f

    def evalPdf(self, indir, out=None):
        """src/rlGgx.h:121-127 -> pdf [n]."""
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n) if out is None else out
        check(ctx.lib.rls_ggx_pdf(ctx.handle, n, C.byref(self.c), cvec3(indir, n, "indir"), plane(pdf, n, "pdf")))
        return pdf

    def sampleEvalPdf(self, rx, ry, out=None):
        """The triple fused in one pass -> (wi, f, pdf, fresnel)."""
        n, ctx = self.n, self.ctx
        wi, f, pdf, F = out if out is not None else (ctx.empty(3, n), ctx.empty(3, n), ctx.empty(n), ctx.empty(n))
        check(ctx.lib.rls_ggx_sample_eval_pdf(ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(ry, n, "ry"),
                                              vec3(wi, n, "wi"), rgb(f, n, "f"), plane(pdf, n, "pdf"),
                                              plane(F, n, "fresnel")))
        return wi, f, pdf, F

    def refractSample(self, rx, ry, out=None):
        """Per-sample body of integrateRefract (src/rlGgx.h:228-242) -> (wt, weight, refracted);
        ``out`` = (wt, weight) or (wt, weight, refracted)."""
        n, ctx = self.n, self.ctx
        wt, w = (out[0], out[1]) if out is not None else (ctx.empty(3, n), ctx.empty(n))
        flag = out[2] if out is not None and len(out) > 2 else torch.empty(n, dtype=torch.uint8, device=ctx.torch_device)
        check(ctx.lib.rls_ggx_refract_sample(ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(ry, n, "ry"),
                                             vec3(wt, n, "wt"), plane(w, n, "weight"), flag.data_ptr()))
        return wt, w, flag

    def reflectRefract(self, rx, ry, rx2, ry2, out=None):
        """Reflect triple + refract sample in one pass -> (wi, f, pdf, fresnel, wt, weight)."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = (ctx.empty(3, n), ctx.empty(3, n), ctx.empty(n), ctx.empty(n), ctx.empty(3, n), ctx.empty(n))
        wi, f, pdf, F, wt, w = out
        check(ctx.lib.rls_ggx_reflect_refract(
            ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(ry, n, "ry"), plane(rx2, n, "rx2"),
            plane(ry2, n, "ry2"), vec3(wi, n, "wi"), rgb(f, n, "f"), plane(pdf, n, "pdf"), plane(F, n, "fresnel"),
            vec3(wt, n, "wt"), plane(w, n, "weight")))
        return out

    def directLighting(self, P, light, spp_n: int, seed: int, KdColor: Color = (1.0, 1.0, 1.0),
                       Kd: Scalar = 0.5, diffuseRoughness: Scalar = 0.0, Ks: Scalar = 0.5, out=None, first_index: int = 0):
        """The light loop of rlGgx's shader_evaluate (src/rlGgx.cpp:274-299) under one spherical light or a sequence
        of them -> (direct_diffuse [3,n], direct_specular [3,n]); parameter names and defaults of src/rlGgx.cpp:170-175."""
        n, ctx = self.n, self.ctx
        dd, ds = out if out is not None else (ctx.empty(3, n), ctx.empty(3, n))
        pn = self.pn
        sh = capi.GgxShader(param_rgb(KdColor, pn, "KdColor"), param(Kd, pn, "Kd"),
                            param(diffuseRoughness, pn, "diffuseRoughness"), param(Ks, pn, "Ks"),
                            param_rgb((1.0, 1.0, 1.0), pn, "KtColor"), param(0.0, pn, "Kt"))
        lights, nl = light_array(light)
        check(ctx.lib.rls_ggx_direct_lighting(ctx.handle, n, C.byref(self.c), C.byref(sh), cvec3(P, n, "P"),
                                              lights, nl, int(spp_n), int(seed) & 0xFFFFFFFF, int(first_index),
                                              rgb(dd, n, "direct_diffuse"), rgb(ds, n, "direct_specular")))
        return dd, ds

    SHADE_AOVS = ("direct_diffuse", "direct_specular", "refraction", "indirect_diffuse", "indirect_specular")

    def shade(self, P, lights, spp_n: int, seed: int, KdColor: Color = (1.0, 1.0, 1.0), Kd: Scalar = 0.5,
              diffuseRoughness: Scalar = 0.0, Ks: Scalar = 0.5, KtColor: Color = (1.0, 1.0, 1.0), Kt: Scalar = 0.0,
              env=(1.0, 1.0, 1.0), traced: bool = True, out=None, first_index: int = 0) -> dict:
        """shader_evaluate of the rlGgx node for a camera ray, whole (src/rlGgx.cpp:248-327; rls_ggx_shade) -> dict of the
        five AOVs (direct_diffuse, direct_specular, refraction, indirect_diffuse, indirect_specular) and out = sg->out.RGB,
        all [3,n]; parameter names and defaults of src/rlGgx.cpp:170-179; ``lights``: None, one light or a sequence."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = {k: ctx.empty(3, n) for k in self.SHADE_AOVS + ("out",)}
        pn = self.pn
        sh = capi.GgxShader(param_rgb(KdColor, pn, "KdColor"), param(Kd, pn, "Kd"),
                            param(diffuseRoughness, pn, "diffuseRoughness"), param(Ks, pn, "Ks"),
                            param_rgb(KtColor, pn, "KtColor"), param(Kt, pn, "Kt"))
        o = capi.GgxShadeOut()
        for k in self.SHADE_AOVS:
            setattr(o, k, rgb(out[k], n, k))
        if "out" in out:
            o.out = rgb(out["out"], n, "out")
        la, nl = light_array(lights)
        e = (C.c_float * 3)(*[float(v) for v in env])
        check(ctx.lib.rls_ggx_shade(ctx.handle, n, C.byref(self.c), C.byref(sh), cvec3(P, n, "P"), la, nl, e,
                                    1 if traced else 0, int(spp_n), int(seed) & 0xFFFFFFFF, int(first_index), C.byref(o)))
        return out

    def microfacet(self, rx, ry, kernel: int = RLS_KERNEL_VNDF):
        """VNDFKernel::evalSample (src/rlGgx.cpp:63-99) or NDFKernel::evalSample (src/rlGgx.h:33-41)."""
        n, ctx = self.n, self.ctx
        m = ctx.empty(3, n)
        check(ctx.lib.rls_ggx_microfacet(ctx.handle, n, C.byref(self.c), kernel, plane(rx, n, "rx"),
                                         plane(ry, n, "ry"), vec3(m, n, "m")))
        return m

    def ndfPdf(self, indir):
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n)
        check(ctx.lib.rls_ggx_ndf_pdf(ctx.handle, n, C.byref(self.c), cvec3(indir, n, "indir"), plane(pdf, n, "pdf")))
        return pdf

    def integrate(self, spp_n: int, seed: int, out=None, first_index: int = 0):
        """spp_n^2 in-kernel samples -> (sum of f/pdf [3,n], getAvgReflectWeight [n]) (src/rlGgx.h:181-184).
        first_index: global index of point 0 (a shard of a larger batch draws the batch's numbers)."""
        n, ctx = self.n, self.ctx
        s, a = out if out is not None else (ctx.empty(3, n), ctx.empty(n))
        check(ctx.lib.rls_ggx_integrate(ctx.handle, n, C.byref(self.c), int(spp_n), int(seed) & 0xFFFFFFFF,
                                        int(first_index), rgb(s, n, "sum"), plane(a, n, "avg")))
        return s, a


    def integrateRefract(self, spp_n: int, seed: int, traced: bool = True, env=(1.0, 1.0, 1.0), want_tir: bool = False,
                         first_index: int = 0):
        """integrateRefract (src/rlGgx.h:205-245) under a uniform environment of radiance ``env`` -> result [3,n]
        (and the fraction of totally internally reflected samples).  traced=False: the single refraction about the
        shading normal of lines 213-222."""
        n, ctx = self.n, self.ctx
        result = ctx.empty(3, n)
        tir = ctx.empty(n) if want_tir else None
        e = (C.c_float * 3)(*[float(v) for v in env])
        check(ctx.lib.rls_ggx_integrate_refract(ctx.handle, n, C.byref(self.c), 1 if traced else 0, e, int(spp_n),
                                                int(seed) & 0xFFFFFFFF, int(first_index), rgb(result, n, "result"),
                                                plane(tir, n, "tir_fraction") if want_tir else None))
        return (result, tir) if want_tir else result


# ================================================================================================
class DisneySampler:
    """Batched ``DisneySampler`` (src/rlDisney.cpp:105-602); parameter names from
    src/rlDisney.cpp:606-610.  ``setSampleType`` picks the lobe the triple acts on."""

    def __init__(self, ctx: Context, wo, N, T, base_color: Color = (1.0, 1.0, 1.0), materials=None, **scalars: Scalar):
        self.ctx = ctx
        self.n = int(wo.shape[-1])
        n = self.n
        unknown = set(scalars) - set(capi.DISNEY_SCALARS)
        if unknown:
            raise TypeError(f"unknown rlDisney parameters: {sorted(unknown)}")
        self._keep = (wo, N, T, base_color, scalars, materials)
        c = capi.DisneyClosure()
        c.wo, c.N, c.T = cvec3(wo, n, "wo"), cvec3(N, n, "N"), cvec3(T, n, "T")
        c.materials, pn = material_index(materials, n)
        c.base_color = param_rgb(base_color, pn, "base_color")
        for name in capi.DISNEY_SCALARS:
            setattr(c, name, param(scalars.get(name, 0.0), pn, name))
        self.c = c
        self.mSampleType = RLS_RAY_GLOSSY

    def setSampleType(self, ray_type: int) -> None:
        if ray_type not in (RLS_RAY_DIFFUSE, RLS_RAY_GLOSSY):
            raise ValueError("ray_type must be RLS_RAY_DIFFUSE or RLS_RAY_GLOSSY")
        self.mSampleType = ray_type

    def evalSample(self, rx, ry):
        n, ctx = self.n, self.ctx
        wi = ctx.empty(3, n)
        check(ctx.lib.rls_disney_sample(ctx.handle, n, C.byref(self.c), self.mSampleType, plane(rx, n, "rx"),
                                        plane(ry, n, "ry"), vec3(wi, n, "wi")))
        return wi

    def evalBrdf(self, indir):
        n, ctx = self.n, self.ctx
        f = ctx.empty(3, n)
        check(ctx.lib.rls_disney_eval(ctx.handle, n, C.byref(self.c), self.mSampleType, cvec3(indir, n, "indir"),
                                      rgb(f, n, "f")))
        return f

    def evalPdf(self, indir):
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n)
        check(ctx.lib.rls_disney_pdf(ctx.handle, n, C.byref(self.c), self.mSampleType, cvec3(indir, n, "indir"),
                                     plane(pdf, n, "pdf")))
        return pdf

    def sampleEvalPdf(self, rx, ry, out=None):
        n, ctx = self.n, self.ctx
        wi, f, pdf = out if out is not None else (ctx.empty(3, n), ctx.empty(3, n), ctx.empty(n))
        check(ctx.lib.rls_disney_sample_eval_pdf(ctx.handle, n, C.byref(self.c), self.mSampleType,
                                                 plane(rx, n, "rx"), plane(ry, n, "ry"), vec3(wi, n, "wi"),
                                                 rgb(f, n, "f"), plane(pdf, n, "pdf")))
        return wi, f, pdf

    # -- alternates the reference compiles but never selects (mSampleFromVisibleNormal = true) -------
    def altSample(self, kind: int, rx, ry):
        """kind 0: sampleGTR2AnisoDirection (src/rlDisney.cpp:406-414), 1: sampleGTR2Direction (504-512)."""
        n, ctx = self.n, self.ctx
        m = ctx.empty(3, n)
        check(ctx.lib.rls_disney_alt_sample(ctx.handle, n, C.byref(self.c), int(kind), plane(rx, n, "rx"),
                                            plane(ry, n, "ry"), vec3(m, n, "m")))
        return m

    def altPdf(self, indir):
        """evalSpecularPdf with mSampleFromVisibleNormal == false (src/rlDisney.cpp:541-542)."""
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n)
        check(ctx.lib.rls_disney_alt_pdf(ctx.handle, n, C.byref(self.c), cvec3(indir, n, "indir"), plane(pdf, n, "pdf")))
        return pdf

    def dGtr2(self, m):
        """D_GTR2 (src/rlDisney.cpp:553-559)."""
        n, ctx = self.n, self.ctx
        d = ctx.empty(n)
        check(ctx.lib.rls_disney_d_gtr2(ctx.handle, n, C.byref(self.c), cvec3(m, n, "m"), plane(d, n, "d")))
        return d

    def integrate(self, spp_n: int, seed: int, streamed: bool = False, out=None, first_index: int = 0):
        """Both lobes, spp_n^2 samples each -> dict(diffuse_sum, diffuse_count, specular_sum,
        specular_count[, wi, f, pdf as [3, 2*spp*n] / [2*spp*n] sample-major planes])."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = {"diffuse_sum": ctx.empty(3, n), "diffuse_count": ctx.empty(n),
                   "specular_sum": ctx.empty(3, n), "specular_count": ctx.empty(n)}
            if streamed:
                m = 2 * spp_n * spp_n * n
                out.update(wi=ctx.empty(3, m), f=ctx.empty(3, m), pdf=ctx.empty(m))
        so = None
        if streamed:
            m = 2 * spp_n * spp_n * n
            so = capi.DisneyStreamOut(vec3(out["wi"], m, "wi"), rgb(out["f"], m, "f"), plane(out["pdf"], m, "pdf"))
        check(ctx.lib.rls_disney_integrate(
            ctx.handle, n, C.byref(self.c), int(spp_n), int(seed) & 0xFFFFFFFF, int(first_index),
            rgb(out["diffuse_sum"], n, "diffuse_sum"), plane(out["diffuse_count"], n, "diffuse_count"),
            rgb(out["specular_sum"], n, "specular_sum"), plane(out["specular_count"], n, "specular_count"),
            C.byref(so) if so is not None else None))
        return out

    def directLighting(self, P, light, spp_n: int, seed: int, out=None, first_index: int = 0):
        """The light loop of rlDisney's shader_evaluate (src/rlDisney.cpp:695-705: evalDiffuseLightSample +
        evalSpecularLightSample per light) under one spherical light or a sequence of them ->
        (direct_diffuse [3,n], direct_specular [3,n]), the two direct AOVs (714-715)."""
        n, ctx = self.n, self.ctx
        dd, ds = out if out is not None else (ctx.empty(3, n), ctx.empty(3, n))
        lights, nl = light_array(light)
        check(ctx.lib.rls_disney_direct_lighting(ctx.handle, n, C.byref(self.c), cvec3(P, n, "P"), lights, nl, int(spp_n),
                                                 int(seed) & 0xFFFFFFFF, int(first_index),
                                                 rgb(dd, n, "direct_diffuse"), rgb(ds, n, "direct_specular")))
        return dd, ds

    SHADE_AOVS = ("direct_diffuse", "direct_specular", "indirect_diffuse", "indirect_specular")

    def shade(self, P, lights, spp_n: int, seed: int, env=(1.0, 1.0, 1.0), out=None, first_index: int = 0) -> dict:
        """shader_evaluate of the rlDisney node for a camera ray, whole (src/rlDisney.cpp:685-727; rls_disney_shade) ->
        dict of the four AOVs and out = sg->out.RGB, all [3,n]; ``lights``: None, one light or a sequence."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = {k: ctx.empty(3, n) for k in self.SHADE_AOVS + ("out",)}
        o = capi.DisneyShadeOut()
        for k in self.SHADE_AOVS:
            setattr(o, k, rgb(out[k], n, k))
        if "out" in out:
            o.out = rgb(out["out"], n, "out")
        la, nl = light_array(lights)
        e = (C.c_float * 3)(*[float(v) for v in env])
        check(ctx.lib.rls_disney_shade(ctx.handle, n, C.byref(self.c), cvec3(P, n, "P"), la, nl, e, int(spp_n),
                                       int(seed) & 0xFFFFFFFF, int(first_index), C.byref(o)))
        return out

    def integrateChunked(self, spp_n: int, seed: int, chunk_points: int, consume=None, out=None, chunk=None,
                         first_index: int = 0):
        """Streamed mode in chunks of the point range (rls_disney_integrate_chunked): every chunk's samples land in the
        same chunk buffers (``chunk`` = dict(wi [3,m], f [3,m], pdf [m]), m = 2*spp*chunk_points, sample-major within
        the chunk) and ``consume(first_point, count, chunk)`` is called after each chunk has been enqueued -- it
        stands for the per-sample loop body of the reference (src/rlDisney.cpp:299-312).  Returns (sums, chunk)."""
        n, ctx = self.n, self.ctx
        spp = spp_n * spp_n
        chunk_points = max(1, min(int(chunk_points), n))
        m = 2 * spp * chunk_points
        if out is None:
            out = {"diffuse_sum": ctx.empty(3, n), "diffuse_count": ctx.empty(n),
                   "specular_sum": ctx.empty(3, n), "specular_count": ctx.empty(n)}
        if chunk is None:
            chunk = dict(wi=ctx.empty(3, m), f=ctx.empty(3, m), pdf=ctx.empty(m))
        so = capi.DisneyStreamOut(vec3(chunk["wi"], m, "wi"), rgb(chunk["f"], m, "f"), plane(chunk["pdf"], m, "pdf"))
        err = []

        def _cb(_user, first_point, count, _chunk):
            try:
                consume(int(first_point), int(count), chunk)
                return 0
            except Exception as e:      # never unwind through the C frames
                err.append(e)
                return 1

        cb = capi.DisneyChunkFn(_cb) if consume is not None else capi.DisneyChunkFn()
        st = ctx.lib.rls_disney_integrate_chunked(
            ctx.handle, n, C.byref(self.c), int(spp_n), int(seed) & 0xFFFFFFFF, int(first_index),
            rgb(out["diffuse_sum"], n, "diffuse_sum"), plane(out["diffuse_count"], n, "diffuse_count"),
            rgb(out["specular_sum"], n, "specular_sum"), plane(out["specular_count"], n, "specular_count"),
            chunk_points, C.byref(so), cb, None)
        if err:
            raise err[0]
        check(st)
        return out, chunk


# ================================================================================================
def _sss_closure(n, sss_scatter_dist, sss_dist_multiplier, sss_color, N, T, has_dPdu, materials=None) -> capi.SssClosure:
    c = capi.SssClosure()
    c.materials, pn = material_index(materials, n)
    c.sss_color = param_rgb(sss_color, pn, "sss_color")
    c.sss_dist_multiplier = param(sss_dist_multiplier, pn, "sss_dist_multiplier")
    if isinstance(sss_scatter_dist, torch.Tensor):
        _chk(sss_scatter_dist, pn, 3, "sss_scatter_dist")
        for k in range(3):
            c.sss_scatter_dist[k] = capi.Param(sss_scatter_dist[k].data_ptr(), 0.0)
    else:
        for k in range(3):
            c.sss_scatter_dist[k] = capi.Param(None, float(sss_scatter_dist[k]))
    if N is not None:
        c.N = cvec3(N, n, "N")
        c.T = cvec3(T, n, "T")
    c.has_dPdu = 1 if has_dPdu else 0
    return c


class NDProfile:
    """Batched ``rls::NDProfile`` (src/rlSss.h:27-61, src/rlSss.cpp:20-106).  ``setDistance(dist,
    albedo)`` happens in the constructor; n must be given when every parameter is uniform."""

    def __init__(self, ctx: Context, n: int, dist, albedo: Color = (1.0, 1.0, 1.0), multiplier: Scalar = 1.0, materials=None):
        self.ctx, self.n = ctx, int(n)
        self._keep = (dist, albedo, multiplier)
        self._materials = materials
        self.c = _sss_closure(self.n, dist, multiplier, albedo, None, None, False, materials)

    def sample(self, rx, out=None):
        """getRadius(rx), getPdf(r), evalProfile(r) in one pass -> (r, pdf, profile)."""
        n, ctx = self.n, self.ctx
        r, pdf, prof = out if out is not None else (ctx.empty(n), ctx.empty(n), ctx.empty(3, n))
        check(ctx.lib.rls_nd_sample(ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(r, n, "r"),
                                    plane(pdf, n, "pdf"), rgb(prof, n, "profile")))
        return r, pdf, prof

    def getRadius(self, rx):
        return self.sample(rx)[0]

    def getPdf(self, r):
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n)
        check(ctx.lib.rls_nd_pdf(ctx.handle, n, C.byref(self.c), plane(r, n, "r"), plane(pdf, n, "pdf")))
        return pdf

    def evalProfile(self, r):
        n, ctx = self.n, self.ctx
        prof = ctx.empty(3, n)
        check(ctx.lib.rls_nd_eval(ctx.handle, n, C.byref(self.c), plane(r, n, "r"), rgb(prof, n, "profile")))
        return prof


class GaussianProfile:
    """Batched ``rls::GaussianProfile`` (src/rlSss.h:63-97), the profile rlSkin leaves commented out
    (src/rlSkin.cpp:242).  ``dist_x`` = dist.x of setDistance: a float or an [n] plane."""

    def __init__(self, ctx: Context, n: int, dist_x: Scalar):
        self.ctx, self.n = ctx, int(n)
        self._keep = dist_x
        self.p = param(dist_x, self.n, "dist_x")

    def sample(self, rx):
        """getRadius(rx), getPdf(r), evalProfile(r) -> (r, pdf, profile)"""
        n, ctx = self.n, self.ctx
        r, pdf, prof = ctx.empty(n), ctx.empty(n), ctx.empty(n)
        check(ctx.lib.rls_gaussian_sample(ctx.handle, n, self.p, plane(rx, n, "rx"), plane(r, n, "r"),
                                          plane(pdf, n, "pdf"), plane(prof, n, "profile")))
        return r, pdf, prof


class SssSampler:
    """Batched hot parts of ``rls::SssSampler<NDProfile>`` (src/rlSss.h:143-167,246-266,401-413,
    487-545).  ``Ns`` = sg->Ns, ``dPdu`` = sg->dPdu (Gram-Schmidt frame) or, with
    ``has_dPdu=False``, the polar-frame tangent."""

    def __init__(self, ctx: Context, Ns, dPdu, albedo: Color, dist, multiplier: Scalar = 1.0, has_dPdu: bool = True,
                 materials=None):
        self.ctx, self.n = ctx, int(Ns.shape[-1])
        self._keep = (Ns, dPdu, albedo, dist, multiplier, materials)
        self.c = _sss_closure(self.n, dist, multiplier, albedo, Ns, dPdu, has_dPdu, materials)

    def getProbeRay(self, rx, ry, P=None, out=None):
        """src/rlSss.h:487-533 -> dict(r, origin, dir, maxdist, pdf, profile)."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = {"r": ctx.empty(n), "origin": ctx.empty(3, n), "dir": ctx.empty(3, n), "maxdist": ctx.empty(n),
                   "pdf": ctx.empty(n), "profile": ctx.empty(3, n)}
        Pv = cvec3(P, n, "P") if P is not None else capi.CVec3(None, None, None)
        check(ctx.lib.rls_sss_probe_ray(
            ctx.handle, n, C.byref(self.c), plane(rx, n, "rx"), plane(ry, n, "ry"), Pv, plane(out["r"], n, "r"),
            vec3(out["origin"], n, "origin"), vec3(out["dir"], n, "dir"), plane(out["maxdist"], n, "maxdist"),
            plane(out["pdf"], n, "pdf"), rgb(out["profile"], n, "profile")))
        return out

    def misPdf(self, disp, sampleN, literal_matrix: bool = False):
        """3-axis MIS pdf of a probe hit (src/rlSss.h:246-266)."""
        n, ctx = self.n, self.ctx
        pdf = ctx.empty(n)
        check(ctx.lib.rls_sss_mis_pdf(ctx.handle, n, C.byref(self.c), cvec3(disp, n, "disp"),
                                      cvec3(sampleN, n, "sampleN"), 1 if literal_matrix else 0, plane(pdf, n, "pdf")))
        return pdf

    def integrateScatter(self, P, scene: "capi.SssScene", spp_n: int, seed: int, want_depth: bool = False, out=None,
                         first_index: int = 0):
        """src/rlSss.h:167-280 over an analytic scene (see rls_sss_integrate_scatter) -> result [3,n]
        (and the mean number of shaded probe hits per probe ray)."""
        n, ctx = self.n, self.ctx
        result = out if out is not None else ctx.empty(3, n)
        depth = ctx.empty(n) if want_depth else None
        check(ctx.lib.rls_sss_integrate_scatter(
            ctx.handle, n, C.byref(self.c), cvec3(P, n, "P"), C.byref(scene), int(spp_n), int(seed) & 0xFFFFFFFF,
            int(first_index), rgb(result, n, "result"), plane(depth, n, "mean_depth") if want_depth else None))
        return (result, depth) if want_depth else result

    @staticmethod
    def cavityFade(ctx: Context, disp, sampleN, No):
        """src/rlSss.h:401-413."""
        n = int(disp.shape[-1])
        fade = ctx.empty(n)
        check(ctx.lib.rls_sss_cavity_fade(ctx.handle, n, cvec3(disp, n, "disp"), cvec3(sampleN, n, "sampleN"),
                                          cvec3(No, n, "No"), plane(fade, n, "fade")))
        return fade

    @staticmethod
    def sampleDiffuseDirection(ctx: Context, rx, ry, normal, T):
        """src/rlSss.h:536-545."""
        n = int(normal.shape[-1])
        wi = ctx.empty(3, n)
        check(ctx.lib.rls_sss_sample_diffuse_direction(ctx.handle, n, cvec3(normal, n, "normal"), cvec3(T, n, "T"),
                                                       plane(rx, n, "rx"), plane(ry, n, "ry"), vec3(wi, n, "wi")))
        return wi


def light_array(lights):
    """one SphereLight, a sequence of them or None -> (ctypes array or None, count)"""
    if lights is None:
        return None, 0
    if isinstance(lights, capi.SphereLight):
        lights = [lights]
    lights = list(lights)
    if not lights:
        return None, 0
    return (capi.SphereLight * len(lights))(*lights), len(lights)


def make_light(center=(0.0, 0.0, 5.0), radius=1.0, radiance=(1.0, 1.0, 1.0), mis_mode=capi.RLS_MIS_BOTH) -> "capi.SphereLight":
    """The spherical area light of ``GgxSampler.directLighting`` (rls_sphere_light)."""
    lt = capi.SphereLight()
    lt.center[:] = center
    lt.radius = radius
    lt.radiance[:] = radiance
    lt.mis_mode = mis_mode
    return lt


def make_scene(geometry="plane", plane_point=(0.0, 0.0, 0.0), plane_normal=(0.0, 0.0, 1.0),
               sphere_center=(0.0, 0.0, 0.0), sphere_radius=1.0,
               light_dir=(0.0, 0.0, 1.0), light_color=(1.0, 1.0, 1.0),
               gate_point=None, gate_normal=(1.0, 0.0, 0.0),
               use_cavity_fade=False, literal_matrix=False) -> "capi.SssScene":
    """The analytic scene of ``SssSampler.integrateScatter`` (rls_sss_scene)."""
    sc = capi.SssScene()
    sc.geometry = {"plane": capi.RLS_SCENE_PLANE, "sphere": capi.RLS_SCENE_SPHERE}[geometry]
    sc.plane_point[:] = plane_point
    sc.plane_normal[:] = plane_normal
    sc.sphere_center[:] = sphere_center
    sc.sphere_radius = sphere_radius
    sc.light_dir[:] = light_dir
    sc.light_color[:] = light_color
    sc.has_gate = 0 if gate_point is None else 1
    sc.gate_point[:] = gate_point if gate_point is not None else (0.0, 0.0, 0.0)
    sc.gate_normal[:] = gate_normal
    sc.use_cavity_fade = 1 if use_cavity_fade else 0
    sc.literal_matrix = 1 if literal_matrix else 0
    return sc


# ================================================================================================
SKIN_OUT_VEC = ("sheen_wi", "sheen_f", "spec_wi", "spec_f", "profile")
SKIN_OUT_SCALAR = ("sheen_pdf", "sheen_fresnel", "spec_pdf", "spec_fresnel", "r", "r_pdf",
                   "sheenFresnel", "specularFresnel", "sssWeight")


class SkinShader:
    """Batched lobe composition of rlSkin's shader_evaluate (src/rlSkin.cpp:174-246); parameter
    names and defaults from src/rlSkin.cpp:109-128."""

    DEFAULTS = dict(sss_color=(1.0, 1.0, 1.0), sss_weight=1.0, sss_dist_multiplier=1.0,
                    sss_scatter_dist=(1.0, 1.0, 1.0),
                    specular_color=(1.0, 1.0, 1.0), specular_weight=0.6, specular_roughness=0.5, specular_ior=1.44,
                    sheen_color=(1.0, 1.0, 1.0), sheen_weight=0.0, sheen_roughness=0.35, sheen_ior=1.44)

    def __init__(self, ctx: Context, wo, N, T, materials=None, **params):
        self.ctx, self.n = ctx, int(wo.shape[-1])
        n = self.n
        unknown = set(params) - set(self.DEFAULTS)
        if unknown:
            raise TypeError(f"unknown rlSkin parameters: {sorted(unknown)}")
        p = dict(self.DEFAULTS)
        p.update(params)
        self._keep = (wo, N, T, p, materials)
        c = capi.SkinClosure()
        c.wo, c.N, c.T = cvec3(wo, n, "wo"), cvec3(N, n, "N"), cvec3(T, n, "T")
        c.materials, pn = material_index(materials, n)
        for name in ("sss_color", "specular_color", "sheen_color"):
            setattr(c, name, param_rgb(p[name], pn, name))
        for name in ("sss_weight", "sss_dist_multiplier", "specular_weight", "specular_roughness", "specular_ior",
                     "sheen_weight", "sheen_roughness", "sheen_ior"):
            setattr(c, name, param(p[name], pn, name))
        d = p["sss_scatter_dist"]
        for k in range(3):
            if isinstance(d, torch.Tensor):
                _chk(d, pn, 3, "sss_scatter_dist")
                c.sss_scatter_dist[k] = capi.Param(d[k].data_ptr(), 0.0)
            else:
                c.sss_scatter_dist[k] = capi.Param(None, float(d[k]))
        self.c = c

    def alloc_out(self, arena: Optional["Arena"] = None) -> dict:
        n, ctx = self.n, self.ctx
        if arena is not None:
            out = {k: arena.planes(3) for k in SKIN_OUT_VEC}
            out.update({k: arena.plane() for k in SKIN_OUT_SCALAR})
            return out
        out = {k: ctx.empty(3, n) for k in SKIN_OUT_VEC}
        out.update({k: ctx.empty(n) for k in SKIN_OUT_SCALAR})
        return out

    def sampleEvalPdf(self, xi, out=None) -> dict:
        """xi: [6, n] (sheen rx, ry, specular rx, ry, sss rx, ry) -> dict of output planes."""
        n, ctx = self.n, self.ctx
        _chk(xi, n, 6, "xi")
        out = self.alloc_out() if out is None else out
        o = capi.SkinOut()
        for k in SKIN_OUT_VEC:
            setattr(o, k, (rgb if k.endswith("_f") or k == "profile" else vec3)(out[k], n, k))
        for k in SKIN_OUT_SCALAR:
            setattr(o, k, plane(out[k], n, k))
        xs = (C.c_void_p * 6)(*[xi[k].data_ptr() for k in range(6)])
        check(ctx.lib.rls_skin_sample_eval_pdf(ctx.handle, n, C.byref(self.c), xs, C.byref(o)))
        return out


    def integrate(self, P, scene: "capi.SssScene", spp_n: int, seed: int, env=(1.0, 1.0, 1.0), out=None,
                  first_index: int = 0, lights=None) -> dict:
        """shader_evaluate over spp_n^2 samples per layer (src/rlSkin.cpp:174-254; rls_skin_integrate) -> dict(sheen,
        specular, sss, out [3,n]; sheenFresnel, specularFresnel, sssWeight [n]).  ``lights``: the spherical lights of
        the two light loops (193-198, 217-222); None = no lights."""
        n, ctx = self.n, self.ctx
        if out is None:
            out = {k: ctx.empty(3, n) for k in ("sheen", "specular", "sss", "out")}
            out.update({k: ctx.empty(n) for k in ("sheenFresnel", "specularFresnel", "sssWeight")})
        o = capi.SkinIntegrateOut()
        for k in ("sheen", "specular", "sss", "out"):
            if k in out:
                setattr(o, k, rgb(out[k], n, k))
        for k in ("sheenFresnel", "specularFresnel", "sssWeight"):
            if k in out:
                setattr(o, k, plane(out[k], n, k))
        e = (C.c_float * 3)(*[float(v) for v in env])
        la, nl = light_array(lights)
        check(ctx.lib.rls_skin_integrate(ctx.handle, n, C.byref(self.c), cvec3(P, n, "P"), C.byref(scene), e, la, nl,
                                         int(spp_n), int(seed) & 0xFFFFFFFF, int(first_index), C.byref(o)))
        return out


# ================================================================================================
def util_reflect_luminance(ctx: Context, i, nrm, color):
    """reflectDirection(i, nrm) and colorToLuminance(color) (src/rlUtil.h:31-39) -> ([3,n], [n])"""
    n = int(i.shape[-1])
    r, lum = ctx.empty(3, n), ctx.empty(n)
    check(ctx.lib.rls_util_reflect_luminance(ctx.handle, n, cvec3(i, n, "i"), cvec3(nrm, n, "nrm"), cvec3(color, n, "color"),
                                             vec3(r, n, "reflected"), plane(lum, n, "luminance")))
    return r, lum


def gen_frame(ctx: Context, seed: int, first: int, n: int, out=None):
    """Synthetic (wo, N, T) planes (DESIGN.md "Synthetic inputs")."""
    wo, N, T = out if out is not None else (ctx.empty(3, n), ctx.empty(3, n), ctx.empty(3, n))
    check(ctx.lib.rls_gen_frame(ctx.handle, seed, first, n, vec3(wo, n, "wo"), vec3(N, n, "N"), vec3(T, n, "T")))
    return wo, N, T


def gen_uniform(ctx: Context, seed: int, first: int, n: int, stream: int, lo: float = 0.0, hi: float = 1.0,
                out: Optional[torch.Tensor] = None):
    out = ctx.empty(n) if out is None else out
    check(ctx.lib.rls_gen_uniform(ctx.handle, seed, first, n, stream, lo, hi, plane(out, n, "out")))
    return out


def gen_aniso(ctx: Context, seed: int, first: int, n: int, out: Optional[torch.Tensor] = None):
    out = ctx.empty(n) if out is None else out
    check(ctx.lib.rls_gen_aniso(ctx.handle, seed, first, n, plane(out, n, "out")))
    return out


def libm_flavour() -> str:
    """which host libm the EXACT kernels reproduce bit for bit: "glibc-fma" or "glibc-sse2" (rls_libm_flavour)"""
    return capi.load().rls_libm_flavour().decode()


def host_libm_mismatches() -> int:
    """0 when THIS process's libm is the one the library follows (rls_host_libm_matches); needs no device"""
    v = C.c_int(-1)
    check(capi.load().rls_host_libm_matches(C.byref(v)))
    return int(v.value)


def checksum(ctx: Context, t: torch.Tensor) -> int:
    flat = t.reshape(-1)
    v = C.c_uint64()
    check(ctx.lib.rls_checksum(ctx.handle, flat.numel(), plane(flat, flat.numel(), "data"), C.byref(v)))
    return int(v.value)


def util_directions(ctx: Context, a, b):
    n = int(a.shape[0])
    sph, disk = ctx.empty(3, n), ctx.empty(3, n)
    check(ctx.lib.rls_util_directions(ctx.handle, n, plane(a, n, "a"), plane(b, n, "b"), vec3(sph, n, "spherical"),
                                      vec3(disk, n, "disk")))
    return sph, disk
