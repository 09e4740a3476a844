"""ctypes mirror of include/vfgs_hip_fw.h: the firmware interface of the reference
(/root/reference/src/vfgs_fw.h:49-92) with the patterns generated on the GPU.

``FgsSei`` / ``FgsAfgs1`` have the members, order and C layout of ``fgs_sei`` / ``fgs_afgs1``;
``init_sei`` / ``init_afgs1`` are ``vfgs_init_sei`` / ``vfgs_init_afgs1``.
"""
from __future__ import annotations

import ctypes as C

from . import hw

SEI_MAX_MODEL_VALUES = 6   # vfgs_fw.h:49

EXPORTS = ["vfgs_init_sei", "vfgs_init_afgs1", "vfgs_hip_generate_patterns", "vfgs_hip_get_pattern",
           "vfgs_hip_cfg_defaults", "vfgs_hip_cfg_read", "vfgs_hip_cfg_check", "vfgs_hip_cfg_adjust_chroma",
           "vfgs_hip_cfg_apply_gain", "vfgs_hip_cfg_program"]


class FgsSei(C.Structure):   # vfgs_fw.h:51-60
    _fields_ = [
        ("model_id", C.c_uint8),
        ("log2_scale_factor", C.c_uint8),
        ("comp_model_present_flag", C.c_uint8 * 3),
        ("num_intensity_intervals", C.c_uint16 * 3),
        ("num_model_values", C.c_uint8 * 3),
        ("intensity_interval_lower_bound", (C.c_uint8 * 256) * 3),
        ("intensity_interval_upper_bound", (C.c_uint8 * 256) * 3),
        ("comp_model_value", ((C.c_int16 * SEI_MAX_MODEL_VALUES) * 256) * 3),
    ]


class FgsAfgs1(C.Structure):   # vfgs_fw.h:62-89
    _fields_ = [
        ("grain_seed", C.c_uint16),
        ("num_y_points", C.c_uint8),
        ("point_y_values", C.c_uint8 * 14),
        ("point_y_scaling", C.c_uint8 * 14),
        ("chroma_scaling_from_luma", C.c_uint8),
        ("num_cb_points", C.c_uint8),
        ("point_cb_values", C.c_uint8 * 10),
        ("point_cb_scaling", C.c_uint8 * 10),
        ("num_cr_points", C.c_uint8),
        ("point_cr_values", C.c_uint8 * 10),
        ("point_cr_scaling", C.c_uint8 * 10),
        ("grain_scaling", C.c_uint8),
        ("ar_coeff_lag", C.c_uint8),
        ("ar_coeffs_y", C.c_int16 * 24),
        ("ar_coeffs_cb", C.c_int16 * 25),
        ("ar_coeffs_cr", C.c_int16 * 25),
        ("ar_coeff_shift", C.c_uint8),
        ("grain_scale_shift", C.c_uint8),
        ("cb_mult", C.c_uint8),
        ("cb_luma_mult", C.c_uint8),
        ("cb_offset", C.c_uint16),
        ("cr_mult", C.c_uint8),
        ("cr_luma_mult", C.c_uint8),
        ("cr_offset", C.c_uint16),
        ("overlap_flag", C.c_uint8),
        ("clip_to_restricted_range", C.c_uint8),
    ]


class PatternJob(C.Structure):   # vfgs_hip_pattern_job
    _fields_ = [
        ("kind", C.c_int32), ("chroma", C.c_int32), ("index", C.c_int32), ("seed_index", C.c_int32),
        ("fh", C.c_int32), ("fv", C.c_int32), ("scale", C.c_int32), ("shift", C.c_int32),
        ("coef", C.c_int16 * 28),
    ]


class Cfg(C.Structure):   # vfgs_hip_cfg: the CLI's configuration state (vfgs_main.c:69-124)
    _fields_ = [("sei", FgsSei), ("afgs1", FgsAfgs1)]

    @classmethod
    def defaults(cls) -> "Cfg":
        c = cls()
        _lib().vfgs_hip_cfg_defaults(C.byref(c))
        return c

    def read(self, filename) -> int:
        """0, or 1 (message: hw.load().vfgs_hip_last_error_string())."""
        return _lib().vfgs_hip_cfg_read(C.byref(self), str(filename).encode())

    def check(self, fmt: int, depth: int) -> int:
        return _lib().vfgs_hip_cfg_check(C.byref(self), fmt, depth)

    def adjust_chroma(self, fmt: int) -> None:
        _lib().vfgs_hip_cfg_adjust_chroma(C.byref(self), fmt)

    def apply_gain(self, gain: int) -> None:
        _lib().vfgs_hip_cfg_apply_gain(C.byref(self), gain)

    def program(self) -> None:
        _lib().vfgs_hip_cfg_program(C.byref(self))

    @property
    def active(self):
        """The parameter set vfgs_init_* would receive (vfgs_main.c:757-760)."""
        return self.afgs1 if self.afgs1.num_y_points else self.sei


def _lib():
    lib = hw.load()
    if not getattr(lib, "_fw_typed", False):
        lib.vfgs_init_sei.argtypes = [C.POINTER(FgsSei)]
        lib.vfgs_init_sei.restype = None
        lib.vfgs_init_afgs1.argtypes = [C.POINTER(FgsAfgs1)]
        lib.vfgs_init_afgs1.restype = None
        lib.vfgs_hip_generate_patterns.argtypes = [C.POINTER(PatternJob), C.c_int]
        lib.vfgs_hip_get_pattern.argtypes = [C.c_int, C.c_int, C.c_void_p]
        lib.vfgs_hip_cfg_defaults.argtypes = [C.POINTER(Cfg)]
        lib.vfgs_hip_cfg_defaults.restype = None
        lib.vfgs_hip_cfg_read.argtypes = [C.POINTER(Cfg), C.c_char_p]
        lib.vfgs_hip_cfg_check.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int]
        lib.vfgs_hip_cfg_adjust_chroma.argtypes = [C.POINTER(Cfg), C.c_int]
        lib.vfgs_hip_cfg_adjust_chroma.restype = None
        lib.vfgs_hip_cfg_apply_gain.argtypes = [C.POINTER(Cfg), C.c_uint]
        lib.vfgs_hip_cfg_apply_gain.restype = None
        lib.vfgs_hip_cfg_program.argtypes = [C.POINTER(Cfg)]
        lib.vfgs_hip_cfg_program.restype = None
        lib._fw_typed = True
    return lib


def struct_from_bytes(kind: int, raw: bytes):
    """kind 0 -> FgsSei, 1 -> FgsAfgs1, from the C bytes of the structure."""
    cls = FgsAfgs1 if kind else FgsSei
    if len(raw) != C.sizeof(cls):
        raise ValueError(f"{cls.__name__}: {len(raw)} bytes, expected {C.sizeof(cls)}")
    return cls.from_buffer_copy(raw)


def init_sei(cfg: FgsSei) -> None:
    _lib().vfgs_init_sei(C.byref(cfg))


def init_afgs1(cfg: FgsAfgs1) -> None:
    _lib().vfgs_init_afgs1(C.byref(cfg))


def init(cfg) -> None:
    (init_afgs1 if isinstance(cfg, FgsAfgs1) else init_sei)(cfg)


def generate_patterns(jobs) -> None:
    lib = _lib()
    arr = (PatternJob * len(jobs))(*jobs)
    rc = lib.vfgs_hip_generate_patterns(arr, len(jobs))
    if rc:
        raise hw.VfgsHipError(f"libvfgs_hip error {rc}: {lib.vfgs_hip_last_error_string().decode()}")


def get_pattern(chroma: int, index: int) -> bytes:
    lib = _lib()
    buf = C.create_string_buffer(4096)
    rc = lib.vfgs_hip_get_pattern(chroma, index, buf)
    if rc:
        raise hw.VfgsHipError(f"libvfgs_hip error {rc}: {lib.vfgs_hip_last_error_string().decode()}")
    return buf.raw
