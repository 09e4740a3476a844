"""CPU: the jump-ahead generator behind the stream of stripe batches (vfgs_host.cpp: LfsrMatrix / LfsrJump / StripeStream) against
the reference's register step (vfgs_hw.c:74-79) -- stepped literally for the short distances, and through the oracle's contiguous
stream (oracle/vfgs_oracle.c: the same step, then the word recurrence) for the long ones.

A rank of a stripe split owns a few block rows of every frame of a batch: its windows lie (nbr - 1) x nblk steps apart
(vfgs_hw.c:291-298,309-310).  vfgs_hip_lfsr_segments hands out exactly what the library uploads for such a call."""
import ctypes as C

import numpy as np
import pytest

import vfgs_testlib as T
from versatilefilmgrain_amd import hw


@pytest.fixture(scope="module")
def lib():
    return hw.load()


@pytest.fixture(scope="module")
def ora():
    T.build_oracle()
    o = C.CDLL(str(T.ORACLE_SO))
    o.vfgs_oracle_lfsr_step.restype = C.c_uint32
    o.vfgs_oracle_lfsr_step.argtypes = [C.c_uint32]
    o.vfgs_oracle_lfsr_stream.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64]
    return o


def segments(lib, reg, first_bit, step_bits, nseg, seg_words):
    out = (C.c_uint32 * (nseg * seg_words))()
    assert lib.vfgs_hip_lfsr_segments(reg, first_bit, step_bits, nseg, seg_words, out) == 0
    return np.frombuffer(out, dtype=np.uint32).reshape(nseg, seg_words).copy()


def window(words, bit):
    w, s = bit >> 5, bit & 31
    return int(words[w]) if s == 0 else ((int(words[w]) >> s) | (int(words[w + 1]) << (32 - s))) & 0xFFFFFFFF


def test_segments_equal_the_literal_register_steps(lib, ora):
    """vfgs_hw.c:74-79 stepped one by one: 2,000 steps per case, every word of every segment"""
    rng = np.random.default_rng(1)
    for reg in (0xdeadbeef, 12345 << 1, 1, 0x80000000, int(rng.integers(1, 1 << 32))):
        first, step, nseg, nw = int(rng.integers(0, 300)), int(rng.integers(1, 400)), 3, 9
        got = segments(lib, reg, first, step, nseg, nw)
        r, at = reg, 0
        regs = {}
        need = {first + f * step + 32 * k for f in range(nseg) for k in range(nw)}
        while at <= max(need):
            if at in need:
                regs[at] = r
            r = ora.vfgs_oracle_lfsr_step(r)
            at += 1
        for f in range(nseg):
            for k in range(nw):
                assert got[f, k] == regs[first + f * step + 32 * k], (hex(reg), first, step, f, k)


@pytest.mark.parametrize("width,height,ranks,frames", [(7680, 4320, 8, 64), (7680, 4320, 4, 32), (3840, 2160, 8, 16), (1920, 1080, 2, 8), (200, 150, 3, 5)])
def test_segments_at_the_shapes_of_a_stripe_split(lib, ora, width, height, ranks, frames):
    """the windows rank r of a stripe split reads for `frames` frames, for every rank: against the contiguous stream of the oracle"""
    nblk, nbr = (width + 15) // 16, (height + 15) // 16
    step = (nbr - 1) * nblk
    total_bits = (frames + 2) * step + nbr * nblk + 4096
    words = (C.c_uint32 * (total_bits // 32 + 2))()
    reg = 12345 << 1
    ora.vfgs_oracle_lfsr_stream(reg, words, len(words))
    stream = np.frombuffer(words, dtype=np.uint32)
    rows = -(-nbr // ranks)
    for r in range(ranks):
        row0 = min(r * rows, nbr - 1)
        nrows = max(1, min(rows, nbr - row0))
        cur0 = step + row0 * nblk                 # second frame of a run: the first one begins at bit 0, in front of which there is nothing
        seg0 = cur0 - nblk - 32
        seg_words = (32 + nblk + nrows * nblk + 64 + 31) // 32 + 1
        got = segments(lib, reg, seg0, step, frames, seg_words)
        for f in (0, 1, frames // 2, frames - 1):
            base = seg0 + f * step
            sh = base & 31
            ref = stream[base >> 5: (base >> 5) + seg_words + 1].astype(np.uint64)
            want = ((ref[:-1] >> np.uint64(sh)) | ((ref[1:] << np.uint64(32 - sh)) if sh else np.uint64(0))) & np.uint64(0xFFFFFFFF)
            assert np.array_equal(got[f].astype(np.uint64), want), (r, f)


def test_a_chain_of_calls_continues_the_jumps(lib, ora):
    """consecutive batches continue one chain of jumps (the image of the next call is built from the last segment of this one):
    three calls of 16 frames = one call of 48"""
    reg, step, nw = 0xdeadbeef, 129360, 40
    one = segments(lib, reg, 5000, step, 48, nw)
    out = (C.c_uint32 * (16 * nw))()
    # (vfgs_hip_lfsr_segments reloads its generator per call: the chain inside ONE call is what the library's look-ahead extends;
    # here the three calls start at the positions a chain would reach)
    for c in range(3):
        assert lib.vfgs_hip_lfsr_segments(reg, 5000 + c * 16 * step, step, 16, nw, out) == 0
        assert np.array_equal(np.frombuffer(out, dtype=np.uint32).reshape(16, nw), one[16 * c:16 * c + 16]), c


def test_refusals(lib):
    out = (C.c_uint32 * 8)()
    assert lib.vfgs_hip_lfsr_segments(1, 0, 1, 0, 8, out) != 0
    assert lib.vfgs_hip_lfsr_segments(1, 0, 1, 1, 8, None) != 0


def test_a_call_behind_images_built_ahead_does_not_walk_the_stream_from_bit_0(tmp_path):
    """The stripe stream builds the image of the NEXT calls ahead and tells the contiguous window's cache how far it got.  A call of
    another kind that takes over at the CURRENT registers then asks for a position in front of that newest known point: it must start
    from an older one, not from the seed (a few hundred thousand frames into a stream, bit 0 is seconds away).  Host layer over the
    HIP runtime model of tests/sanitize (no GPU): 3,000 chained 64-frame stripe calls at the 8-rank shape of 4320p, then whole frames."""
    import os
    import shutil
    import subprocess
    import time
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    csrc, inc = root / "versatilefilmgrain_amd" / "csrc", Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "include"
    if shutil.which("g++") is None or not (inc / "hip" / "hip_runtime_api.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    so = tmp_path / "libvfgs_host_model.so"
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", f"-I{inc}", f'-DVFGS_FW_TABLES_PATH="{csrc / "fw_tables.bin"}"',
                        str(csrc / "vfgs_host.cpp"), str(csrc / "vfgs_fw_host.cpp"), str(csrc / "vfgs_cfg_host.cpp"), str(root / "tests" / "sanitize" / "hip_stub.cpp"),
                        "-o", str(so), "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    code = f"""
import ctypes as C, time, sys
lib = C.CDLL({str(so)!r})
vp, u = C.c_void_p, C.c_uint
lib.vfgs_hip_add_grain_frames_part_dev.argtypes = [vp, vp, vp, u, u, u, u, u, u, u, C.c_uint64, C.c_uint64, vp]
lib.vfgs_hip_add_grain_frame_dev.argtypes = [vp, vp, vp, u, u, u, u, vp]
lib.vfgs_hip_get_stripe_stream_stats.argtypes = [C.POINTER(C.c_uint64)]
lib.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
lib.vfgs_set_depth(10); lib.vfgs_set_chroma_subsampling(2, 2); lib.vfgs_set_scale_shift(5); lib.vfgs_set_seed(12345)
W, H, py, ph = 7680, 4320, 544, 544
Y, U, V = vp(), vp(), vp()
lib.hipMalloc(C.byref(Y), W * H * 2); lib.hipMalloc(C.byref(U), W * H // 2); lib.hipMalloc(C.byref(V), W * H // 2)
for i in range(3000):
    assert lib.vfgs_hip_add_grain_frames_part_dev(Y, U, V, W, H, py, ph, W, W // 2, 64, 0, 0, None) == 0
    if i % 64 == 0: lib.hipDeviceSynchronize()
st = (C.c_uint64 * 4)(); lib.vfgs_hip_get_stripe_stream_stats(st)
assert st[3] == 1 and st[1] > 2900, list(st)
t0 = time.perf_counter()
assert lib.vfgs_hip_add_grain_frame_dev(Y, U, V, W, H, W, W // 2, None) == 0      # 192,000 frames in: a contiguous window at the current registers
dt = time.perf_counter() - t0
lib.hipDeviceSynchronize()
print("foreign call behind the chain: %.1f ms" % (dt * 1e3))
sys.exit(0 if dt < 0.1 else 1)
"""
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
