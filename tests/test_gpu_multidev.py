"""GPU (MI355X): several devices inside ONE process for frames that live in host memory (vfgs_hip_init_devices; SURVEY 8e x 8f
row f3 -- the reference's frame loop and file I/O are one process, vfgs_main.c:664-682, yuv.c:162-214).  The box has one GPU, so
device 0 is listed two or three times: every "device" has its own replica of the programmed state, its own streams, buffers and
worker thread, and processes its stripe of every frame -- results and seed registers must be those of the oracle's
frame-by-frame run, also when single-device calls are mixed in and when the patterns were generated on the primary's device."""
import ctypes as C
import json

import numpy as np
import pytest

import vfgs_testlib as T

pytestmark = pytest.mark.gpu
MD5 = json.loads((T.GOLDEN / "md5.json").read_text())


@pytest.fixture()
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from versatilefilmgrain_amd import hw
    h = hw.VfgsHip(device=0)
    yield h
    h.init_devices([0])          # back to one device for the rest of the session


def program(hip, name):
    hip.lib.vfgs_hip_reset_state()
    rec = T.load_trace(name)
    T.replay(hip, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    return ora, T.trace_geometry(rec)


def random_frames(n, width, height, depth, sx, sy, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        f = T.Frame(width, height, depth, sx, sy)
        for p in f.planes():
            p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape).astype(f.dtype)   # garbage incl. the stride padding
        out.append(f)
    return out


@pytest.mark.parametrize("ndev", [2, 3])
@pytest.mark.parametrize("name,width,height", [("fgs_sei_10_420", 416, 240), ("fgs_afgs1_test1_8_444", 200, 152),
                                               ("fgs_sei_ff_test6_8_422", 264, 136), ("fgs_sei_10_420", 1920, 1080)])
def test_host_frames_split_over_devices(hip, name, width, height, ndev):
    hip.init_devices([0] * ndev)
    ora, (depth, sx, sy) = program(hip, name)
    frames = random_frames(5, width, height, depth, sx, sy, width + ndev)
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    hip.add_grain_frames_host([f.Y.ctypes.data for f in frames], [f.U.ctypes.data for f in frames],
                              [f.V.ctypes.data for f in frames], width, height, frames[0].stride, frames[0].cstride)
    for i, (f, w) in enumerate(zip(frames, want)):
        assert f.equal_all(w), (i, name)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,width,height,nframes", [("fgs_sei_10_420", 7680, 4320, 2), ("fgs_sei_10_420", 416, 136, 4), ("fgs_afgs1_test1_8_420", 1920, 1080, 3)])
def test_eight_way_split(hip, name, width, height, nframes):
    """The node the scaling bench runs on has eight GPUs; the one-GPU box cannot hold eight rank PROCESSES (its process guard allows
    six), but eight replica states of device 0 in one process run the same decomposition with the real kernels: 4320p -> 270 block
    rows -> stripes of 34,34,34,34,34,34,33,33 rows, 136 lines -> 9 block rows over 8 devices (2,1,1,1,1,1,1,1)."""
    hip.init_devices([0] * 8)
    ora, (depth, sx, sy) = program(hip, name)
    frames = random_frames(nframes, width, height, depth, sx, sy, width + 8)
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    hip.add_grain_frames_host([f.Y.ctypes.data for f in frames], [f.U.ctypes.data for f in frames],
                              [f.V.ctypes.data for f in frames], width, height, frames[0].stride, frames[0].cstride)
    for i, (f, w) in enumerate(zip(frames, want)):
        assert f.equal_all(w), (i, name)
    assert hip.seed_state() == ora.seed_state()
    f = frames[0]
    w = f.copy()
    ora.add_grain_frame(w)
    hip.add_grain_stripe(f.Y.ctypes.data, f.U.ctypes.data, f.V.ctypes.data, 0, width, height, f.stride, f.cstride)
    assert f.equal_all(w)
    assert hip.seed_state() == ora.seed_state()


def test_stripes_split_over_devices_and_mix_with_single_device_calls(hip):
    """stripe (2 devices) -> device-pointer frame (primary only) -> new seed -> pipelined frames (2 devices) -> a stripe that
    starts in the middle of a block row -> line calls: one seed sequence, as the oracle's."""
    from gpu_util import DevFrame, stream_ptr
    hip.init_devices([0, 0])
    name = "fgs_sei_ar_test1_8_420"
    ora, (depth, sx, sy) = program(hip, name)
    W, H = 336, 208
    fr = random_frames(6, W, H, depth, sx, sy, 21)
    want = [f.copy() for f in fr]
    st = lambda f, y, h: hip.add_grain_stripe(f.Y[y:].ctypes.data, f.U[y // sy:].ctypes.data, f.V[y // sy:].ctypes.data, y, W, h, f.stride, f.cstride)
    # 0: one stripe = the whole frame
    st(fr[0], 0, H); ora.add_grain_frame(want[0])
    # 1: device-resident frame through the primary
    d = DevFrame(fr[1])
    hip.add_grain_frame_dev(*d.ptrs(), W, H, fr[1].stride, fr[1].cstride, stream_ptr())
    fr[1] = d.download(); ora.add_grain_frame(want[1])
    hip.set_seed(777); ora.set_seed(777)
    # 2, 3: pipelined
    hip.add_grain_frames_host([f.Y.ctypes.data for f in fr[2:4]], [f.U.ctypes.data for f in fr[2:4]], [f.V.ctypes.data for f in fr[2:4]],
                              W, H, fr[2].stride, fr[2].cstride)
    ora.add_grain_frame(want[2]); ora.add_grain_frame(want[3])
    # 4: two stripes, the second starting in the middle of a block row
    st(fr[4], 0, 72); st(fr[4], 72, H - 72); ora.add_grain_frame(want[4])
    # 5: line by line
    f = fr[5]
    for y in range(H):
        hip.add_grain_line(f.Y[y:].ctypes.data, f.U[y // sy:].ctypes.data, f.V[y // sy:].ctypes.data, y, W)
    ora.add_grain_frame(want[5])
    for i, (f, w) in enumerate(zip(fr, want)):
        assert f.equal_all(w), i
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_420", "fgs_sei_ar_test1_10_420"])
def test_patterns_generated_on_the_primary_reach_the_replicas(hip, name):
    """cfg structures -> our firmware (patterns generated on the primary's device, never on the host) -> frames split over
    three 'devices' == the reference CLI's output md5."""
    from versatilefilmgrain_amd import fw
    hip.init_devices([0, 0, 0])
    depth, sx, sy = T.trace_geometry(T.load_trace(name))
    seed, cfgs = T.load_fwcfg(name)
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(depth)
    hip.set_chroma_subsampling(sx, sy)
    for i, (kind, raw) in enumerate(cfgs):
        fw.init(fw.struct_from_bytes(kind, raw))
        if i == 0:
            hip.set_seed(seed)
    frames, _ = T.lcg_frames(192, 144, depth, sx, sy, 3)
    hip.add_grain_frames_host([f.Y.ctypes.data for f in frames], [f.U.ctypes.data for f in frames], [f.V.ctypes.data for f in frames],
                              192, 144, frames[0].stride, frames[0].cstride)
    assert T.md5_frames(frames) == MD5["small"][name]


def test_init_devices_refuses_nonsense(hip):
    lib = hip.lib
    assert lib.vfgs_hip_init_devices(None, 1) == 26
    arr = (C.c_int * 9)(*([0] * 9))
    assert lib.vfgs_hip_init_devices(arr, 9) == 26
    bad = (C.c_int * 2)(0, 99)
    assert lib.vfgs_hip_init_devices(bad, 2) == 2
    hip.init_devices([0, 0])
    hip.init_devices([0])
