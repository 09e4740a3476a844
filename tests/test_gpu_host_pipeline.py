"""GPU (MI355X): frames in HOST memory through the pipelined entry point vfgs_hip_add_grain_frames_host
(the data path around yuv_read / yuv_write, yuv.c:162-214): upload, kernel and download of
consecutive frames overlap on three streams and a ring of three device frames.  Results, stride padding and seed
registers must equal the oracle's frame-by-frame run; pinned and pageable memory; more frames than ring slots."""
import ctypes as C

import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from versatilefilmgrain_amd import hw
    return hw.VfgsHip(device=0)


def program(hip, name):
    hip.lib.vfgs_hip_reset_state()
    rec = T.load_trace(name)
    T.replay(hip, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    return ora, T.trace_geometry(rec)


def pinned_like(hip, arr, keep):
    """A numpy array of arr's shape / dtype over pinned memory from the library, holding arr's contents."""
    p = hip.host_alloc(arr.nbytes + 64)
    keep.append(p)
    buf = (C.c_char * arr.nbytes).from_address(p)
    out = np.frombuffer(buf, dtype=arr.dtype).reshape(arr.shape)
    out[...] = arr
    return out


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("name,width,height", [("fgs_sei_10_420", 416, 240), ("fgs_afgs1_test1_8_444", 200, 152),
                                               ("fgs_sei_ff_test6_8_422", 264, 136), ("fgs_sei_10_440", 1000, 70)])
def test_frames_in_host_memory_pipelined(hip, name, width, height, pinned):
    ora, (depth, sx, sy) = program(hip, name)
    nframes = 8                                   # the ring has 3 slots: every slot is reused at least twice
    rng = np.random.default_rng(width + nframes)
    frames, want, keep = [], [], []
    for i in range(nframes):
        f = T.Frame(width, height, depth, sx, sy)
        for p in f.planes():
            p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape).astype(f.dtype)   # garbage incl. the stride padding
        w = f.copy()
        ora.add_grain_frame(w)
        want.append(w)
        if pinned:
            f.Y, f.U, f.V = (pinned_like(hip, a, keep) for a in (f.Y, f.U, f.V))
        frames.append(f)
    try:
        hip.add_grain_frames_host([f.Y.ctypes.data for f in frames], [f.U.ctypes.data for f in frames],
                                  [f.V.ctypes.data for f in frames], width, height, frames[0].stride, frames[0].cstride)
        for i, (f, w) in enumerate(zip(frames, want)):
            assert f.equal_all(w), (i, name)
        assert hip.seed_state() == ora.seed_state()
    finally:
        del frames
        for p in keep:
            hip.host_free(p)


def test_host_pipeline_equals_frame_calls_and_mixes_with_them(hip):
    """The pipelined call is interchangeable with the other entry points: frames 0-2 through it, frame 3 through the
    stripe call, frames 4-5 through it again -- one seed sequence."""
    name = "fgs_sei_ar_test1_8_420"
    ora, (depth, sx, sy) = program(hip, name)
    rng = np.random.default_rng(5)
    fr = []
    for i in range(6):
        f = T.Frame(336, 96, depth, sx, sy)
        for p in f.planes():
            p[...] = rng.integers(0, 256, p.shape).astype(f.dtype)
        fr.append(f)
    want = [f.copy() for f in fr]
    for w in want:
        ora.add_grain_frame(w)
    ptrs = lambda fs: ([f.Y.ctypes.data for f in fs], [f.U.ctypes.data for f in fs], [f.V.ctypes.data for f in fs])
    hip.add_grain_frames_host(*ptrs(fr[0:3]), 336, 96, fr[0].stride, fr[0].cstride)
    f = fr[3]
    hip.add_grain_stripe(f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data, 0, 336, 96, f.stride, f.cstride)
    hip.add_grain_frames_host(*ptrs(fr[4:6]), 336, 96, fr[0].stride, fr[0].cstride)
    for f, w in zip(fr, want):
        assert f.equal_all(w)
    assert hip.seed_state() == ora.seed_state()


def test_host_pipeline_refuses_bad_arguments(hip):
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f = T.Frame(416, 64, depth, sx, sy)
    before = hip.seed_state()
    keepY = f.Y.copy()
    arr = lambda ps: (C.c_void_p * len(ps))(*ps)
    Y, U, V = arr([f.Y.ctypes.data]), arr([f.U.ctypes.data]), arr([f.V.ctypes.data])
    lib = hip.lib
    assert lib.vfgs_hip_add_grain_frames_host(Y, U, V, 0, 416, 64, f.stride, f.cstride) == 0           # no frames
    assert lib.vfgs_hip_add_grain_frames_host(None, U, V, 1, 416, 64, f.stride, f.cstride) == 4
    assert lib.vfgs_hip_add_grain_frames_host(arr([None]), U, V, 1, 416, 64, f.stride, f.cstride) == 4
    assert lib.vfgs_hip_add_grain_frames_host(Y, U, V, 1, 100, 64, f.stride, f.cstride) == 5           # vfgs_hw.c:168
    assert lib.vfgs_hip_add_grain_frames_host(Y, U, V, 1, 416, 64, 400, f.cstride) == 6                # stride < whole blocks
    assert hip.seed_state() == before == ora.seed_state()
    assert np.array_equal(f.Y, keepY)


def test_host_pipeline_error_mid_call_drains_what_was_queued(hip):
    """A failure in the middle of the frame loop (frame 5 of 7 has a null plane pointer) returns the error AFTER the three
    streams have been drained: the frames queued before it are complete in host memory when the call returns, and the next
    call finds the ring free."""
    name = "fgs_sei_10_420"
    ora, (depth, sx, sy) = program(hip, name)
    rng = np.random.default_rng(99)
    fr = []
    for i in range(7):
        f = T.Frame(416, 240, depth, sx, sy)
        for p in f.planes():
            p[...] = rng.integers(0, 1024, p.shape).astype(f.dtype)
        fr.append(f)
    want = [f.copy() for f in fr]
    for w in want[:5]:
        ora.add_grain_frame(w)
    arr = lambda ps: (C.c_void_p * len(ps))(*ps)
    Y = arr([f.Y.ctypes.data for f in fr[:5]] + [None] + [fr[6].Y.ctypes.data])
    U = arr([f.U.ctypes.data for f in fr])
    V = arr([f.V.ctypes.data for f in fr])
    rc = hip.lib.vfgs_hip_add_grain_frames_host(Y, U, V, 7, 416, 240, fr[0].stride, fr[0].cstride)
    assert rc == 4
    for i in range(5):                      # everything queued before the bad frame came home
        assert fr[i].equal_all(want[i]), i
    assert fr[6].equal_all(want[6])         # never touched
    assert hip.seed_state() == ora.seed_state()
    # the pipeline is usable again right away
    ora.add_grain_frame(want[6])
    hip.add_grain_frames_host([fr[6].Y.ctypes.data], [fr[6].U.ctypes.data], [fr[6].V.ctypes.data], 416, 240, fr[0].stride, fr[0].cstride)
    assert fr[6].equal_all(want[6])
    assert hip.seed_state() == ora.seed_state()
