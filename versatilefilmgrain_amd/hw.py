"""ctypes mirror of include/vfgs_hip.h.

Same names and argument meaning as the reference hardware-layer interface
(/root/reference/src/vfgs_hw.h:51-62) plus the vfgs_hip_* extensions.  The library state is
a process-global singleton exactly like the reference's (vfgs_hw.c:49-68), so ``VfgsHip`` is
a namespace around that singleton, not an object with private state.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .build import LIB

_lib = None


class VfgsHipError(RuntimeError):
    pass


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's) and opens it by path:
    if this library is loaded first it binds /opt/rocm's copy, a later `import torch` brings a second runtime into the process
    and whichever initialises second finds no device ("no ROCm-capable device is detected", seen when build() and smoke() ran in
    one process).  So where torch is installed but not imported yet, its copy is opened first (by the path torch itself will
    use) and libvfgs_hip.so's NEEDED entry resolves to it -- the arrangement every test and bench.py run has anyway (they import
    torch first).  Without torch (a C host) nothing happens and /opt/rocm's runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("VFGS_HIP_NO_TORCH_RUNTIME"):     # (opt out: a Python host that will never import torch)
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    rt = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if rt.exists():
        try:
            C.CDLL(str(rt), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def hip_runtimes_mapped() -> list[str]:
    """Paths of every libamdhip64 mapped into this process (Linux): more than one means two HIP runtimes, and whichever
    initialises second finds no device."""
    out = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in path and path not in out:
                    out.append(path)
    except OSError:
        pass
    return out


def load(path: Path | None = None) -> C.CDLL:
    """dlopen libvfgs_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(path or LIB)
    if not path.exists():
        raise VfgsHipError(f"{path} not built: run `python -m versatilefilmgrain_amd.build` "
                           "(needs hipcc); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(str(path))
    rts = hip_runtimes_mapped()
    if len(rts) > 1:
        import warnings
        warnings.warn(f"libvfgs_hip: {len(rts)} HIP runtimes are mapped into this process ({', '.join(rts)}): device calls of the one "
                      "that initialises second will fail.  Import torch before loading the library, or set VFGS_HIP_NO_TORCH_RUNTIME=1 "
                      "in a process that never imports torch.", RuntimeWarning, stacklevel=2)
    vp, u, i = C.c_void_p, C.c_uint, C.c_int
    lib.vfgs_set_luma_pattern.argtypes = [i, vp]
    lib.vfgs_set_chroma_pattern.argtypes = [i, vp]
    lib.vfgs_set_scale_lut.argtypes = [i, vp]
    lib.vfgs_set_pattern_lut.argtypes = [i, vp]
    lib.vfgs_set_seed.argtypes = [u]
    lib.vfgs_set_scale_shift.argtypes = [i]
    lib.vfgs_set_depth.argtypes = [i]
    lib.vfgs_set_legal_range.argtypes = [i]
    lib.vfgs_set_chroma_subsampling.argtypes = [i, i]
    lib.vfgs_add_grain_line.argtypes = [vp, vp, vp, i, i]
    lib.vfgs_add_grain_stripe.argtypes = [vp, vp, vp, u, u, u, u, u]
    lib.vfgs_hip_init.argtypes = [i]
    lib.vfgs_hip_init_devices.argtypes = [C.POINTER(i), i]
    lib.vfgs_hip_overlap_begin.argtypes = [vp]
    lib.vfgs_hip_get_stream_stats.argtypes = [C.POINTER(C.c_uint64)]
    lib.vfgs_hip_get_stream_stats.restype = None
    lib.vfgs_hip_get_stripe_stream_stats.argtypes = [C.POINTER(C.c_uint64)]
    lib.vfgs_hip_get_stripe_stream_stats.restype = None
    lib.vfgs_hip_lfsr_segments.argtypes = [u, C.c_uint64, C.c_uint64, u, u, C.POINTER(C.c_uint32)]
    lib.vfgs_hip_overlap_end.argtypes = [vp]
    lib.vfgs_hip_add_grain_stripe_dev.argtypes = [vp, vp, vp, u, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frame_dev.argtypes = [vp, vp, vp, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frame_part_dev.argtypes = [vp, vp, vp, u, u, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frames_dev.argtypes = [vp, vp, vp, u, u, u, u, u, C.c_uint64, C.c_uint64, vp]
    lib.vfgs_hip_add_grain_frames_part_dev.argtypes = [vp, vp, vp, u, u, u, u, u, u, u, C.c_uint64, C.c_uint64, vp]
    lib.vfgs_hip_add_grain_copy_dev.argtypes = [vp, vp, vp, vp, vp, vp, u, u, u, u, u, u, u, C.c_uint64, C.c_uint64, vp]
    lib.vfgs_hip_add_grain_copy8_dev.argtypes = [vp, vp, vp, vp, vp, vp, u, u, u, u, u, u, u, u, u,
                                                 C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    fp = C.POINTER(FramePtrs)
    lib.vfgs_hip_add_grain_frame_list_dev.argtypes = [fp, u, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frame_list_part_dev.argtypes = [fp, u, u, u, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frame_list_copy_dev.argtypes = [fp, fp, u, u, u, u, u, vp]
    lib.vfgs_hip_add_grain_frame_list_copy8_dev.argtypes = [fp, fp, u, u, u, u, u, u, u, vp]
    lib.vfgs_hip_get_seed_state.argtypes = [vp]
    lib.vfgs_hip_get_luts.argtypes = [i, vp, vp]
    lib.vfgs_hip_get_params.argtypes = [vp]
    lib.vfgs_hip_get_params.restype = None
    lib.vfgs_hip_last_error_string.restype = C.c_char_p
    lib.vfgs_hip_timer_begin.argtypes = [vp]
    lib.vfgs_hip_timer_end.argtypes = [vp, C.POINTER(C.c_float)]
    lib.vfgs_hip_device_info.argtypes = [C.POINTER(i), C.POINTER(i), C.POINTER(i), C.c_char_p, i]
    lib.vfgs_hip_line_lookahead.argtypes = [i]
    lib.vfgs_hip_line_lookahead.restype = None
    lib.vfgs_hip_declare_frame.argtypes = [vp, vp, vp, u, u, u, u]
    lib.vfgs_hip_add_grain_frames_host.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u, u, u, u, u]
    lib.vfgs_hip_host_alloc.argtypes = [C.c_uint64]
    lib.vfgs_hip_host_alloc.restype = vp
    lib.vfgs_hip_host_free.argtypes = [vp]
    lib.vfgs_hip_host_free.restype = None
    lib.vfgs_hip_last_launch_info.argtypes = [C.POINTER(LaunchInfo)]
    # a library built by a developer tool with tuning / ablation knobs may compute something else by design: only on request
    if lib.vfgs_hip_dev_build() and not os.environ.get("VFGS_ALLOW_DEV_BUILD"):
        raise VfgsHipError(f"{path} is a developer build (tuning / ablation knobs); set VFGS_ALLOW_DEV_BUILD=1 to load it anyway")
    _lib = lib
    return lib


class LaunchInfo(C.Structure):
    """include/vfgs_hip.h: vfgs_hip_launch_info."""
    _fields_ = [("depth", C.c_int), ("csubx", C.c_int), ("csuby", C.c_int), ("out8", C.c_int), ("one_y", C.c_int), ("one_c", C.c_int),
                ("in_place", C.c_int), ("nframes", C.c_int), ("workgroups_per_frame", C.c_int), ("frames_per_front", C.c_int),
                ("rows_per_wave", C.c_int * 2), ("positions_per_row", C.c_int * 2), ("parts_per_row", C.c_int), ("persistent_luma_workgroups", C.c_int),
                ("waves_per_workgroup", C.c_int), ("lds_bytes_per_workgroup", C.c_int), ("launches", C.c_ulonglong), ("kernel", C.c_char * 96), ("listed", C.c_int), ("internal", C.c_int)]


class FramePtrs(C.Structure):
    """include/vfgs_hip.h: vfgs_hip_frame_ptrs (device pointers to line 0 of one frame's planes)."""
    _fields_ = [("Y", C.c_void_p), ("U", C.c_void_p), ("V", C.c_void_p)]


EXPORTS = [
    # drop-in, vfgs_hw.h:51-62
    "vfgs_set_luma_pattern", "vfgs_set_chroma_pattern", "vfgs_set_scale_lut", "vfgs_set_pattern_lut",
    "vfgs_set_seed", "vfgs_set_scale_shift", "vfgs_set_depth", "vfgs_set_legal_range",
    "vfgs_set_chroma_subsampling", "vfgs_add_grain_line",
    # extensions
    "vfgs_add_grain_stripe", "vfgs_hip_init", "vfgs_hip_shutdown", "vfgs_hip_reset_state",
    "vfgs_hip_add_grain_stripe_dev", "vfgs_hip_add_grain_frame_dev", "vfgs_hip_add_grain_frame_part_dev",
    "vfgs_hip_add_grain_frames_dev", "vfgs_hip_add_grain_frames_part_dev", "vfgs_hip_add_grain_copy_dev",
    "vfgs_hip_add_grain_copy8_dev", "vfgs_hip_add_grain_frame_list_dev", "vfgs_hip_add_grain_frame_list_part_dev", "vfgs_hip_add_grain_frame_list_copy_dev",
    "vfgs_hip_add_grain_frame_list_copy8_dev", "vfgs_hip_get_seed_state", "vfgs_hip_get_luts", "vfgs_hip_get_params", "vfgs_hip_last_error",
    "vfgs_hip_last_error_string", "vfgs_hip_timer_begin", "vfgs_hip_timer_end", "vfgs_hip_device_info",
    "vfgs_hip_dev_build", "vfgs_hip_init_devices", "vfgs_hip_overlap_begin", "vfgs_hip_overlap_end", "vfgs_hip_get_stream_stats", "vfgs_hip_line_lookahead", "vfgs_hip_declare_frame",
    "vfgs_hip_get_stripe_stream_stats", "vfgs_hip_lfsr_segments",
    "vfgs_hip_add_grain_frames_host", "vfgs_hip_host_alloc", "vfgs_hip_host_free", "vfgs_hip_last_launch_info",
]


def _b(data):
    return (C.c_char * len(data)).from_buffer_copy(bytes(data))


class VfgsHip:
    """Namespace over the library singleton (same method names as the test-side wrappers)."""

    def __init__(self, device: int | None = None, reset: bool = True):
        self.lib = load()
        if device is not None:
            self._ck(self.lib.vfgs_hip_init(device))
        if reset:
            self.lib.vfgs_hip_reset_state()

    def _ck(self, rc):
        if rc:
            raise VfgsHipError(f"libvfgs_hip error {rc}: {self.lib.vfgs_hip_last_error_string().decode()}")

    # ---- drop-in setters
    def set_luma_pattern(self, i, P):       self.lib.vfgs_set_luma_pattern(i, _b(P))
    def set_chroma_pattern(self, i, P):     self.lib.vfgs_set_chroma_pattern(i, _b(P))
    def set_scale_lut(self, c, lut):        self.lib.vfgs_set_scale_lut(c, _b(lut))
    def set_pattern_lut(self, c, lut):      self.lib.vfgs_set_pattern_lut(c, _b(lut))
    def set_seed(self, s):                  self.lib.vfgs_set_seed(s & 0xFFFFFFFF)
    def set_scale_shift(self, s):           self.lib.vfgs_set_scale_shift(s)
    def set_depth(self, d):                 self.lib.vfgs_set_depth(d)
    def set_legal_range(self, l):           self.lib.vfgs_set_legal_range(l)
    def set_chroma_subsampling(self, x, y): self.lib.vfgs_set_chroma_subsampling(x, y)

    # ---- host-memory processing (pointers are plain integers / ctypes addresses)
    def add_grain_line(self, Y, U, V, y, width):
        self.lib.vfgs_add_grain_line(Y, U, V, y, width)

    def add_grain_stripe(self, Y, U, V, y, width, height, stride, cstride):
        self.lib.vfgs_add_grain_stripe(Y, U, V, y, width, height, stride, cstride)

    # ---- device-resident processing
    def add_grain_stripe_dev(self, dY, dU, dV, y, width, height, stride, cstride, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_stripe_dev(dY, dU, dV, y, width, height, stride, cstride, stream))

    def add_grain_frame_dev(self, dY, dU, dV, width, height, stride, cstride, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_frame_dev(dY, dU, dV, width, height, stride, cstride, stream))

    def add_grain_frame_part_dev(self, dY, dU, dV, width, frame_height, part_y, part_height, stride, cstride, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_frame_part_dev(dY, dU, dV, width, frame_height, part_y, part_height,
                                                            stride, cstride, stream))

    def add_grain_frames_dev(self, dY, dU, dV, width, height, stride, cstride, nframes, ypitch, cpitch, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_frames_dev(dY, dU, dV, width, height, stride, cstride, nframes,
                                                        ypitch, cpitch, stream))

    def add_grain_frames_part_dev(self, dY, dU, dV, width, frame_height, part_y, part_height, stride, cstride,
                                  nframes, ypitch, cpitch, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_frames_part_dev(dY, dU, dV, width, frame_height, part_y, part_height,
                                                             stride, cstride, nframes, ypitch, cpitch, stream))

    def add_grain_copy_dev(self, sY, sU, sV, dY, dU, dV, width, frame_height, part_y, part_height, stride, cstride,
                           nframes, ypitch, cpitch, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_copy_dev(sY, sU, sV, dY, dU, dV, width, frame_height, part_y, part_height,
                                                      stride, cstride, nframes, ypitch, cpitch, stream))

    def add_grain_copy8_dev(self, sY, sU, sV, dY, dU, dV, width, frame_height, part_y, part_height, stride, cstride,
                            dstride, dcstride, nframes, ypitch, cpitch, dypitch, dcpitch, stream=0):
        self._ck(self.lib.vfgs_hip_add_grain_copy8_dev(sY, sU, sV, dY, dU, dV, width, frame_height, part_y, part_height,
                                                       stride, cstride, dstride, dcstride, nframes, ypitch, cpitch,
                                                       dypitch, dcpitch, stream))

    @staticmethod
    def frame_list(frames):
        """frames: sequence of (Y, U, V) device addresses -> ctypes array of vfgs_hip_frame_ptrs (build it once for a pool of
        frames that is handed over again and again; the list calls take either form)."""
        if isinstance(frames, C.Array):
            return frames
        arr = (FramePtrs * len(frames))()
        for k, (y, u_, v) in enumerate(frames):
            arr[k].Y, arr[k].U, arr[k].V = y, u_, v
        return arr

    def add_grain_frame_list_dev(self, frames, width, height, stride, cstride, stream=0):
        """Frames anywhere in device memory, one launch per 32 of them (include/vfgs_hip.h)."""
        self._ck(self.lib.vfgs_hip_add_grain_frame_list_dev(self.frame_list(frames), len(frames), width, height, stride, cstride, stream))

    def add_grain_frame_list_part_dev(self, frames, width, frame_height, part_y, part_height, stride, cstride, stream=0):
        """Lines [part_y, part_y + part_height) of every listed frame; the pointers address line part_y; seeds advance as for whole frames."""
        self._ck(self.lib.vfgs_hip_add_grain_frame_list_part_dev(self.frame_list(frames), len(frames), width, frame_height, part_y, part_height,
                                                                 stride, cstride, stream))

    def add_grain_frame_list_copy_dev(self, src, dst, width, height, stride, cstride, stream=0):
        assert len(src) == len(dst)
        self._ck(self.lib.vfgs_hip_add_grain_frame_list_copy_dev(self.frame_list(src), self.frame_list(dst), len(src), width, height,
                                                                 stride, cstride, stream))

    def add_grain_frame_list_copy8_dev(self, src, dst, width, height, stride, cstride, dstride, dcstride, stream=0):
        assert len(src) == len(dst)
        self._ck(self.lib.vfgs_hip_add_grain_frame_list_copy8_dev(self.frame_list(src), self.frame_list(dst), len(src), width, height,
                                                                  stride, cstride, dstride, dcstride, stream))

    def seed_state(self):
        out = (C.c_uint32 * 4)()
        self.lib.vfgs_hip_get_seed_state(out)
        return tuple(out)

    def luts(self, c):
        """(scale LUT, pattern LUT) of component c as bytes."""
        a, b = C.create_string_buffer(256), C.create_string_buffer(256)
        self._ck(self.lib.vfgs_hip_get_luts(c, a, b))
        return a.raw, b.raw

    def params(self):
        out = (C.c_int * 8)()
        self.lib.vfgs_hip_get_params(out)
        return dict(zip(("scale_shift", "bs", "ymin", "ymax", "cmin", "cmax", "csubx", "csuby"), out))

    def line_lookahead(self, enable):
        self.lib.vfgs_hip_line_lookahead(1 if enable else 0)

    def declare_frame(self, Y, U, V, width, height, stride, cstride):
        self._ck(self.lib.vfgs_hip_declare_frame(Y, U, V, width, height, stride, cstride))

    def add_grain_frames_host(self, Ys, Us, Vs, width, height, stride, cstride):
        """Frames in host memory (lists of plane addresses), pipelined upload / kernel / download."""
        n = len(Ys)
        assert len(Us) == n and len(Vs) == n
        arr = lambda ps: (C.c_void_p * n)(*ps)
        self._ck(self.lib.vfgs_hip_add_grain_frames_host(arr(Ys), arr(Us), arr(Vs), n, width, height, stride, cstride))

    def host_alloc(self, nbytes):
        p = self.lib.vfgs_hip_host_alloc(nbytes)
        if not p:
            raise VfgsHipError(f"vfgs_hip_host_alloc({nbytes}): {self.lib.vfgs_hip_last_error_string().decode()}")
        return p

    def host_free(self, p):
        self.lib.vfgs_hip_host_free(p)

    def stream_stats(self):
        out = (C.c_uint64 * 4)()
        self.lib.vfgs_hip_get_stream_stats(out)
        return {"refills_in_stream": out[0], "windows_built_ahead": out[1], "switches_to_built_ahead": out[2], "window_words": out[3]}

    def stripe_stream_stats(self):
        out = (C.c_uint64 * 4)()
        self.lib.vfgs_hip_get_stripe_stream_stats(out)
        return {"built_in_stream": out[0], "built_ahead": out[1], "switches_to_built_ahead": out[2], "last_launch_used_it": bool(out[3])}

    def overlap_begin(self, stream=0):
        self._ck(self.lib.vfgs_hip_overlap_begin(stream))

    def overlap_end(self, stream=0):
        self._ck(self.lib.vfgs_hip_overlap_end(stream))

    def init_devices(self, devices):
        """Host-memory frames and stripes are split over these devices from now on (devices[0] stays the library's device)."""
        arr = (C.c_int * len(devices))(*devices)
        self._ck(self.lib.vfgs_hip_init_devices(arr, len(devices)))

    def last_launch_info(self):
        """What the most recent grain launch dispatched (dict), or None before the first launch."""
        li = LaunchInfo()
        if self.lib.vfgs_hip_last_launch_info(C.byref(li)):
            return None
        d = {k: getattr(li, k) for k, _ in LaunchInfo._fields_ if k not in ("rows_per_wave", "positions_per_row", "kernel")}
        d["rows_per_wave"] = list(li.rows_per_wave)
        d["positions_per_row"] = list(li.positions_per_row)
        d["kernel"] = li.kernel.decode()
        return d

    def device_info(self):
        cu, lds, clk = C.c_int(), C.c_int(), C.c_int()
        name = C.create_string_buffer(128)
        self._ck(self.lib.vfgs_hip_device_info(C.byref(cu), C.byref(lds), C.byref(clk), name, 128))
        return {"cu_count": cu.value, "lds_per_cu": lds.value, "clock_khz": clk.value, "name": name.value.decode()}
