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
