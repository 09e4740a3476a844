"""Shared test plumbing (TEST INFRASTRUCTURE).

Three implementations of the VFGS hardware layer are driven through one small
Python interface so that the parity tests read the same for all of them:

* ``OracleHW``   -- this repo's CPU restatement (oracle/vfgs_oracle.c)
* ``ReferenceHW``-- the real reference hardware layer compiled in place from
                    /root/reference/src/vfgs_hw.c into oracle/_ref/libvfgs_ref.so
                    (process-global state, vfgs_hw.c:49-68 -> one instance at a time)
* the product, ``versatilefilmgrain_amd.hw.VfgsHip`` (HIP, GPU only)

All of them are *programmed* by replaying hw-programming traces
(tests/golden/traces/*.npz) that were recorded from the reference firmware
(oracle/trace_shim.c), because the firmware layer is out of scope and
/root/reference does not exist on the GPU box.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
ORACLE_SO = ORACLE_DIR / "_build" / "libvfgs_oracle.so"
REF_DIR = ORACLE_DIR / "_ref"
REF_SO = REF_DIR / "libvfgs_ref.so"
GOLDEN = ROOT / "tests" / "golden"
TRACES = GOLDEN / "traces"

OP_LUMA_PATTERN, OP_CHROMA_PATTERN, OP_SCALE_LUT, OP_PATTERN_LUT = 1, 2, 3, 4
OP_SEED, OP_SCALE_SHIFT, OP_DEPTH, OP_LEGAL_RANGE, OP_CHROMA_SUBSAMPLING = 5, 6, 7, 8, 9


def build_oracle() -> Path:
    """(Re)build the oracle .so (and the reference builds when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR), "oracle", "ref"], check=True)
    return ORACLE_SO


def have_reference() -> bool:
    return REF_SO.exists()


# --------------------------------------------------------------------------- traces

def parse_trace_file(path) -> list[tuple[int, int, int, bytes]]:
    """Raw trace written by oracle/trace_shim.c -> list of (op, a, b, payload)."""
    raw = Path(path).read_bytes()
    assert raw[:4] == b"VFGT", "not a trace file"
    pos, out = 8, []
    while pos < len(raw):
        op, a, b, n = np.frombuffer(raw, dtype="<i4", count=4, offset=pos)
        pos += 16
        out.append((int(op), int(a), int(b), raw[pos:pos + int(n)]))
        pos += int(n)
    return out


def save_trace_npz(records, path):
    ops = np.array([(op, a, b, len(p)) for op, a, b, p in records], dtype=np.int32).reshape(-1, 4)
    blob = np.frombuffer(b"".join(p for *_, p in records), dtype=np.uint8)
    np.savez_compressed(path, ops=ops, blob=blob)


def load_trace(name: str):
    """name like 'fgs_sei_10_420' -> list of (op, a, b, payload bytes)."""
    with np.load(TRACES / f"{name}.npz") as z:
        ops, blob = z["ops"], z["blob"].tobytes()
    out, pos = [], 0
    for op, a, b, n in ops:
        out.append((int(op), int(a), int(b), blob[pos:pos + int(n)]))
        pos += int(n)
    return out


def list_traces() -> list[str]:
    return sorted(p.stem for p in TRACES.glob("*.npz"))


def replay(hw, records):
    """Feed a recorded programming sequence to any implementation."""
    for op, a, b, payload in records:
        if op == OP_LUMA_PATTERN:
            hw.set_luma_pattern(a, payload)
        elif op == OP_CHROMA_PATTERN:
            hw.set_chroma_pattern(a, payload)
        elif op == OP_SCALE_LUT:
            hw.set_scale_lut(a, payload)
        elif op == OP_PATTERN_LUT:
            hw.set_pattern_lut(a, payload)
        elif op == OP_SEED:
            hw.set_seed(a & 0xFFFFFFFF)
        elif op == OP_SCALE_SHIFT:
            hw.set_scale_shift(a)
        elif op == OP_DEPTH:
            hw.set_depth(a)
        elif op == OP_LEGAL_RANGE:
            hw.set_legal_range(a)
        elif op == OP_CHROMA_SUBSAMPLING:
            hw.set_chroma_subsampling(a, b)
        else:
            raise ValueError(f"unknown trace op {op}")


def trace_geometry(records):
    """(depth, subx, suby) a trace leaves the hardware layer in."""
    depth, subx, suby = 8, 2, 2
    for op, a, b, _ in records:
        if op == OP_DEPTH:
            depth = a
        elif op == OP_CHROMA_SUBSAMPLING:
            subx, suby = a, b
    return depth, subx, suby


# --------------------------------------------------------------------------- frames

class Frame:
    """Planar YUV frame with the reference's buffer geometry (yuv.c:54-87):
    stride = width rounded up to 64 samples (unchanged if already a multiple),
    luma height padded to 16 in the allocation only."""

    def __init__(self, width, height, depth, subx=2, suby=2, stride=None, cstride=None):
        self.width, self.height, self.depth, self.subx, self.suby = width, height, depth, subx, suby
        self.cwidth, self.cheight = width // subx, height // suby
        al = lambda w: w if w % 64 == 0 else (w + 64) & ~63
        self.stride = stride or al(width)
        self.cstride = cstride or al(self.cwidth)
        self.dtype = np.uint16 if depth > 8 else np.uint8
        h2 = (height + 15) & ~15
        ch2 = h2 // suby
        self.Y = np.zeros((h2, self.stride), self.dtype)
        self.U = np.zeros((ch2, self.cstride), self.dtype)
        self.V = np.zeros((ch2, self.cstride), self.dtype)

    def copy(self):
        f = Frame(self.width, self.height, self.depth, self.subx, self.suby, self.stride, self.cstride)
        f.Y[...], f.U[...], f.V[...] = self.Y, self.U, self.V
        return f

    def planes(self):
        return self.Y, self.U, self.V

    def picture_bytes(self) -> bytes:
        """The bytes yuv_write emits (yuv.c:188-214): visible area only, Y then U then V."""
        return (self.Y[:self.height, :self.width].tobytes()
                + self.U[:self.cheight, :self.cwidth].tobytes()
                + self.V[:self.cheight, :self.cwidth].tobytes())

    def equal_picture(self, other) -> bool:
        return self.picture_bytes() == other.picture_bytes()

    def equal_all(self, other) -> bool:
        return all(np.array_equal(a, b) for a, b in zip(self.planes(), other.planes()))


_oracle_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        if not ORACLE_SO.exists():
            build_oracle()
        lib = C.CDLL(str(ORACLE_SO))
        lib.vfgs_oracle_create.restype = C.c_void_p
        lib.vfgs_oracle_lcg_fill.restype = C.c_uint32
        lib.vfgs_oracle_lcg_fill.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64, C.c_int]
        lib.vfgs_oracle_lfsr_stream.argtypes = [C.c_uint32, C.c_void_p, C.c_uint64]
        lib.vfgs_oracle_lfsr_step.restype = C.c_uint32
        lib.vfgs_oracle_lfsr_step.argtypes = [C.c_uint32]
        _oracle_lib = lib
    return _oracle_lib


def lcg_frames(width, height, depth, subx, suby, nframes, state=1, garbage_padding=False):
    """Synthetic input of SURVEY.md Appendix B: one LCG run through Y,U,V of every frame."""
    lib = oracle_lib()
    frames = []
    for _ in range(nframes):
        f = Frame(width, height, depth, subx, suby)
        if garbage_padding:
            rng = np.random.default_rng(state)
            for p in f.planes():
                p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape, dtype=np.uint32).astype(f.dtype)
        for plane, (w, h) in zip(f.planes(), ((width, height), (f.cwidth, f.cheight), (f.cwidth, f.cheight))):
            buf = np.empty(w * h, f.dtype)
            state = lib.vfgs_oracle_lcg_fill(state, buf.ctypes.data, buf.size, depth)
            plane[:h, :w] = buf.reshape(h, w)
        frames.append(f)
    return frames, state


def md5_frames(frames) -> str:
    m = hashlib.md5()
    for f in frames:
        m.update(f.picture_bytes())
    return m.hexdigest()


# --------------------------------------------------------------------------- implementations

def _buf(b):
    return (C.c_char * len(b)).from_buffer_copy(bytes(b))


class OracleHW:
    """oracle/vfgs_oracle.c through ctypes (re-entrant: one context per instance)."""

    def __init__(self):
        self.lib = oracle_lib()
        self.ctx = C.c_void_p(self.lib.vfgs_oracle_create())

    def __del__(self):
        try:
            self.lib.vfgs_oracle_destroy(self.ctx)
        except Exception:
            pass

    def set_luma_pattern(self, i, P):       self.lib.vfgs_oracle_set_luma_pattern(self.ctx, i, _buf(P))
    def set_chroma_pattern(self, i, P):     self.lib.vfgs_oracle_set_chroma_pattern(self.ctx, i, _buf(P))
    def set_scale_lut(self, c, lut):        self.lib.vfgs_oracle_set_scale_lut(self.ctx, c, _buf(lut))
    def set_pattern_lut(self, c, lut):      self.lib.vfgs_oracle_set_pattern_lut(self.ctx, c, _buf(lut))
    def set_seed(self, s):                  self.lib.vfgs_oracle_set_seed(self.ctx, C.c_uint32(s))
    def set_scale_shift(self, s):           self.lib.vfgs_oracle_set_scale_shift(self.ctx, s)
    def set_depth(self, d):                 self.lib.vfgs_oracle_set_depth(self.ctx, d)
    def set_legal_range(self, l):           self.lib.vfgs_oracle_set_legal_range(self.ctx, l)
    def set_chroma_subsampling(self, x, y): self.lib.vfgs_oracle_set_chroma_subsampling(self.ctx, x, y)

    def add_grain_line(self, Y, U, V, y, width):
        self.lib.vfgs_oracle_add_grain_line(self.ctx, C.c_void_p(Y), C.c_void_p(U), C.c_void_p(V), y, width)

    def add_grain_frame(self, f: Frame, closed_form=False):
        fn = self.lib.vfgs_oracle_add_grain_frame_closed_form if closed_form else self.lib.vfgs_oracle_add_grain_frame
        fn(self.ctx, C.c_void_p(f.Y.ctypes.data), C.c_void_p(f.U.ctypes.data), C.c_void_p(f.V.ctypes.data),
           f.width, f.height, f.stride, f.cstride)

    def seed_state(self):
        out = (C.c_uint32 * 4)()
        self.lib.vfgs_oracle_get_seed_state(self.ctx, out)
        return tuple(out)


class ReferenceHW:
    """The real reference hardware layer (process-global state: use one at a time).

    A fresh private copy of the .so is loaded per instance so that every instance
    starts from the reference's power-on state (vfgs_hw.c:49-63)."""

    _n = 0

    def __init__(self):
        import shutil
        import tempfile
        assert have_reference(), "oracle/_ref/libvfgs_ref.so missing (run make -C oracle ref where /root/reference exists)"
        ReferenceHW._n += 1
        self._tmp = tempfile.NamedTemporaryFile(suffix=f"_vfgsref{ReferenceHW._n}.so", delete=False)
        self._tmp.close()
        shutil.copyfile(REF_SO, self._tmp.name)
        self.lib = C.CDLL(self._tmp.name)
        os.unlink(self._tmp.name)

    def set_luma_pattern(self, i, P):       self.lib.vfgs_set_luma_pattern(i, _buf(P))
    def set_chroma_pattern(self, i, P):     self.lib.vfgs_set_chroma_pattern(i, _buf(P))
    def set_scale_lut(self, c, lut):        self.lib.vfgs_set_scale_lut(c, _buf(lut))
    def set_pattern_lut(self, c, lut):      self.lib.vfgs_set_pattern_lut(c, _buf(lut))
    def set_seed(self, s):                  self.lib.vfgs_set_seed(C.c_uint32(s))
    def set_scale_shift(self, s):           self.lib.vfgs_set_scale_shift(s)
    def set_depth(self, d):                 self.lib.vfgs_set_depth(d)
    def set_legal_range(self, l):           self.lib.vfgs_set_legal_range(l)
    def set_chroma_subsampling(self, x, y):
        self.lib.vfgs_set_chroma_subsampling(x, y)
        self.suby = y

    suby = 2

    def add_grain_line(self, Y, U, V, y, width):
        self.lib.vfgs_add_grain_line(C.c_void_p(Y), C.c_void_p(U), C.c_void_p(V), y, width)

    def add_grain_frame(self, f: Frame):
        """Frame loop of vfgs_main.c:664-682."""
        sz = f.Y.itemsize
        py, pu, pv = f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data
        fn = self.lib.vfgs_add_grain_line
        for y in range(f.height):
            fn(C.c_void_p(py), C.c_void_p(pu), C.c_void_p(pv), y, f.width)
            py += f.stride * sz
            if (y & 1) or f.suby == 1:
                pu += f.cstride * sz
                pv += f.cstride * sz


# --------------------------------------------------------------------------- firmware fixtures

FWCFG = GOLDEN / "fwcfg"


def load_fwcfg(name: str):
    """-> (seed, [(kind, struct bytes), ...]) the reference CLI handed to its firmware, in call order."""
    with np.load(FWCFG / f"{name}.npz") as z:
        kinds = [int(k) for k in z["kinds"]]
        return int(z["seed"][0]), [(k, z[f"cfg{i}"].tobytes()) for i, k in enumerate(kinds)]


class BankModel:
    """What the setters leave in the pattern banks (vfgs_hw.c:314-325), for comparing a trace with
    the banks of an implementation.  Only the region the hardware layer reads is kept."""

    def __init__(self):
        self.subx = self.suby = 2
        self.luma, self.chroma = {}, {}

    def set_luma_pattern(self, i, P):
        self.luma[i] = np.frombuffer(bytes(P), dtype=np.int8)[:4096].reshape(64, 64).copy()

    def set_chroma_pattern(self, i, P):
        p = np.frombuffer(bytes(P), dtype=np.int8)
        rows, cols, pitch = 64 // self.suby, 64 // self.subx, 64 // self.suby
        self.chroma[i] = np.stack([p[pitch * r: pitch * r + cols] for r in range(rows)]).copy()

    def set_chroma_subsampling(self, x, y):
        self.subx, self.suby = x, y

    def set_scale_lut(self, c, lut): pass
    def set_pattern_lut(self, c, lut): pass
    def set_seed(self, s): pass
    def set_scale_shift(self, s): pass
    def set_depth(self, d): pass
    def set_legal_range(self, l): pass


class StateModel(BankModel):
    """Everything the setters leave behind except the pattern bytes' device copies: banks
    (BankModel), LUTs, shift (vfgs_hw.c:346-362), ranges (:364-380), subsampling, seed."""

    def __init__(self):
        super().__init__()
        self.slut = [bytes(256)] * 3
        self.plut = [bytes(256)] * 3
        self.shift, self.bs = 5 + 6, 0
        self.rng = (0, 255, 0, 255)
        self.seed = 0xdeadbeef

    def set_scale_lut(self, c, lut): self.slut[c] = bytes(lut)[:256]
    def set_pattern_lut(self, c, lut): self.plut[c] = bytes(lut)[:256]
    def set_seed(self, s): self.seed = (s << 1) & 0xFFFFFFFF
    def set_scale_shift(self, s): self.shift = s + 6 - self.bs

    def set_depth(self, d):
        self.shift += self.bs - (d - 8)
        self.bs = d - 8

    def set_legal_range(self, l): self.rng = (16, 235, 16, 240) if l else (0, 255, 0, 255)
