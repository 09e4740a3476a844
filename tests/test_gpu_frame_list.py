"""GPU (MI355X): vfgs_hip_add_grain_frame_list_* -- frames at arbitrary device addresses in ONE launch -- against the oracle,
through the C ABI.

The contract (include/vfgs_hip.h): results and seed registers are those of one vfgs_hip_add_grain_frame_dev call per frame in
list order -- which is the reference's own loop, one frame per call (vfgs_main.c:771-790).  Every frame here is an allocation
of its own (listed in an order that is NOT the address order), lists are mixed with the ordinary entry points, and the shapes
that pick the persistent luma workgroups and the two-frame fronts of large pictures are reached through the list as well."""
import numpy as np
import pytest

import vfgs_testlib as T
from gpu_util import DevFrame, stream_ptr

pytestmark = pytest.mark.gpu

FORMATS = ["fgs_sei_10_420", "fgs_afgs1_test1_8_444", "fgs_sei_8_420", "fgs_afgs1_test1_8_420", "fgs_sei_10_422", "fgs_sei_ff_test6_10_440",
           "fgs_sei_ar_test1_10_420", "fgs_sei_10_444"]


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


def garbage_frame(width, height, depth, sx, sy, seed):
    rng = np.random.default_rng(seed)
    f = T.Frame(width, height, depth, sx, sy)
    for p in f.planes():
        p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape).astype(f.dtype)
    return f


def scattered(frames, seed=0):
    """One device allocation per frame, allocated in a shuffled order so that list order != address order."""
    order = np.random.default_rng(seed).permutation(len(frames))
    dev = [None] * len(frames)
    for i in order:
        dev[i] = DevFrame(frames[i])
    return dev


def run_list(hip, ora, frames, seed=0):
    """frames through the list entry (in place) vs the oracle frame by frame; returns the device frames"""
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    dev = scattered(frames, seed)
    f0 = frames[0]
    hip.add_grain_frame_list_dev([d.ptrs() for d in dev], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    for i, (d, w) in enumerate(zip(dev, want)):
        assert d.download().equal_all(w), i
    assert hip.seed_state() == ora.seed_state()
    return dev


@pytest.mark.parametrize("name", FORMATS)
def test_scattered_frames_equal_one_call_per_frame(hip, name):
    ora, (depth, sx, sy) = program(hip, name)
    frames = [garbage_frame(1032, 90, depth, sx, sy, 10 + i) for i in range(7)]
    run_list(hip, ora, frames)
    info = hip.last_launch_info()
    assert info["listed"] == 1 and info["nframes"] == 7 and info["in_place"] == 1 and info["internal"] == 0, info


def test_lists_mixed_with_the_other_entry_points(hip):
    """frame call, list, contiguous batch, list of one, stripe calls, list: one seed sequence"""
    import torch
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    W, H = 520, 70
    mk = lambda n, s: [garbage_frame(W, H, depth, sx, sy, s + i) for i in range(n)]
    # 1: one ordinary frame
    f = mk(1, 0)[0]
    w = f.copy(); ora.add_grain_frame(w)
    d = DevFrame(f)
    hip.add_grain_frame_dev(*d.ptrs(), W, H, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(w)
    # 2: a list of five
    run_list(hip, ora, mk(5, 100), seed=1)
    # 3: a contiguous batch of three
    fr = mk(3, 200)
    want = [x.copy() for x in fr]
    for x in want:
        ora.add_grain_frame(x)
    Y = torch.from_numpy(np.stack([x.Y for x in fr]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([x.U for x in fr]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([x.V for x in fr]).view(np.uint8)).cuda()
    hip.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), W, H, fr[0].stride, fr[0].cstride, 3, Y[0].numel(), U[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    assert hip.last_launch_info()["listed"] == 0
    for i, x in enumerate(want):
        assert np.array_equal(Y[i].cpu().numpy().view(x.dtype).reshape(x.Y.shape), x.Y) and np.array_equal(V[i].cpu().numpy().view(x.dtype).reshape(x.V.shape), x.V)
    # 4: a list of one, 5: a frame in two stripes, 6: a list again
    run_list(hip, ora, mk(1, 300))
    f = mk(1, 400)[0]
    w = f.copy(); ora.add_grain_frame(w)
    d = DevFrame(f)
    hip.add_grain_stripe_dev(*d.ptrs(0), 0, W, 32, f.stride, f.cstride, stream_ptr())
    hip.add_grain_stripe_dev(*d.ptrs(32), 32, W, H - 32, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(w)
    run_list(hip, ora, mk(4, 500), seed=2)


def test_more_frames_than_one_launch_holds(hip):
    """70 frames = launches of 32 + 32 + 6"""
    ora, (depth, sx, sy) = program(hip, "fgs_afgs1_test1_8_420")
    frames = [garbage_frame(264, 40, depth, sx, sy, 700 + i) for i in range(70)]
    n0 = (hip.last_launch_info() or {"launches": 0})["launches"]
    run_list(hip, ora, frames, seed=3)
    info = hip.last_launch_info()
    assert info["launches"] - n0 == 3 and info["nframes"] == 6 and info["listed"] == 1, info


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_10_444"])
def test_list_with_persistent_luma_workgroups(hip, name):
    """General-form luma, 32 frames of 69 block rows: several rounds of luma tasks -> persistent luma workgroups, whose tasks
    walk the frames of the list (a new plane pointer per task)."""
    ora, (depth, sx, sy) = program(hip, name)
    frames = [garbage_frame(256, 1100, depth, sx, sy, 900 + i) for i in range(32)]
    run_list(hip, ora, frames, seed=4)
    info = hip.last_launch_info()
    assert info["persistent_luma_workgroups"] > 0 and info["listed"] == 1, info


def test_the_benchmarked_shape_1080p_x32_scattered(hip):
    """bench.py's `configs` entry: 32 full-size 1080p frames of the default SEI model, every plane an allocation of its own"""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    frames, _ = T.lcg_frames(1920, 1080, depth, sx, sy, 32)
    run_list(hip, ora, frames, seed=11)
    info = hip.last_launch_info()
    assert info["nframes"] == 32 and info["persistent_luma_workgroups"] > 0 and info["listed"] == 1, info


def test_list_of_large_frames_runs_two_fronts(hip):
    """4320p: frames 2m and 2m + 1 of a launch are swept together (frames_per_front == 2) -- through the list as well; an odd count"""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    frames = [garbage_frame(7680, 4320, depth, sx, sy, 50 + i) for i in range(3)]
    run_list(hip, ora, frames, seed=5)
    info = hip.last_launch_info()
    assert info["frames_per_front"] == 2 and info["listed"] == 1, info


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_420", "fgs_sei_10_444"])
def test_list_of_stripes_equals_the_stripe_split_of_every_frame(hip, name):
    """What one rank of an N-way stripe split runs on a pool of frames: lines [32, 64 + 6) of every listed frame (the last block row
    partial), pointers at line 32, registers advancing as for whole frames.  Three ranks' parts together = the whole frames."""
    ora, (depth, sx, sy) = program(hip, name)
    W, H = 1032, 150
    frames = [garbage_frame(W, H, depth, sx, sy, 40 + i) for i in range(5)]
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    dev = scattered(frames, 12)
    f0 = frames[0]
    st0 = None
    for py, ph in ((0, 32), (32, 38), (70, 80)):          # (the second part ends inside a block row, the third begins inside it: 70 is not a multiple of 16 -> refused)
        if py & 15:
            from versatilefilmgrain_amd.hw import VfgsHipError
            with pytest.raises(VfgsHipError, match="multiple of 16"):
                hip.add_grain_frame_list_part_dev([d.ptrs(py) for d in dev], W, H, py, ph, f0.stride, f0.cstride, stream_ptr())
            continue
    # a proper three-way split: block rows 0-1, 2-4, 5-9 (150 lines = 9 block rows + 6 lines)
    for rank, (py, ph) in enumerate(((0, 32), (32, 48), (80, 70))):
        program_state = program(hip, name)                # every rank starts from the same programmed state and seed
        hip.add_grain_frame_list_part_dev([d.ptrs(py) for d in dev], W, H, py, ph, f0.stride, f0.cstride, stream_ptr())
        st = hip.seed_state()
        assert st0 is None or st == st0                   # all ranks end with the same registers: those of whole frames
        st0 = st
    assert st0 == ora.seed_state()
    for i, (d, w) in enumerate(zip(dev, want)):
        assert d.download().equal_all(w), i


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_444"])
def test_list_out_of_place(hip, name):
    """src[f] -> dst[f]; one pair in place (src == dst); the sources stay as they were"""
    ora, (depth, sx, sy) = program(hip, name)
    frames = [garbage_frame(1032, 90, depth, sx, sy, 30 + i) for i in range(6)]
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    src = scattered(frames, 6)
    blank = T.Frame(1032, 90, depth, sx, sy)
    dst = [DevFrame(blank) for _ in frames]
    dst[2] = src[2]
    f0 = frames[0]
    hip.add_grain_frame_list_copy_dev([d.ptrs() for d in src], [d.ptrs() for d in dst], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    cy = (f0.width + 15) // 16 * 16
    for i, (s, d, w, f) in enumerate(zip(src, dst, want, frames)):
        g = d.download()
        if i == 2:
            assert g.equal_all(w)
            continue
        assert np.array_equal(g.Y[:90, :cy], w.Y[:90, :cy]) and np.array_equal(g.U[:90 // sy, :cy // sx], w.U[:90 // sy, :cy // sx]) and np.array_equal(g.V[:90 // sy, :cy // sx], w.V[:90 // sy, :cy // sx]), i
        assert not g.Y[:, cy:].any() and not g.Y[90:].any()          # the destination's padding is never written
        assert s.download().equal_all(f), i
    assert hip.seed_state() == ora.seed_state()
    assert hip.last_launch_info()["in_place"] == 0


def test_list_with_8bit_output(hip):
    """10-bit sources, 8-bit destinations (yuv.c:216-258 fused into the store), every frame its own allocations"""
    import torch
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    W, H = 1032, 70
    frames = [garbage_frame(W, H, depth, sx, sy, 60 + i) for i in range(5)]
    for f in frames:
        for p in f.planes():
            np.minimum(p, 0xfffd, out=p)
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    src = scattered(frames, 7)
    f8 = T.Frame(W, H, 8, sx, sy)
    dst = []
    for _ in frames:
        dst.append(tuple(torch.full(p.shape, 0x5a, dtype=torch.uint8, device="cuda") for p in f8.planes()))
    f0 = frames[0]
    hip.add_grain_frame_list_copy8_dev([d.ptrs() for d in src], [tuple(t.data_ptr() for t in d) for d in dst], W, H, f0.stride, f0.cstride,
                                       f8.stride, f8.cstride, stream_ptr())
    torch.cuda.synchronize()
    nblk = (W + 15) // 16
    for i, (w, d) in enumerate(zip(want, dst)):
        for got, w16, rows, cols in ((d[0], w.Y, H, nblk * 16), (d[1], w.U, H // sy, nblk * 16 // sx), (d[2], w.V, H // sy, nblk * 16 // sx)):
            g = got.cpu().numpy()
            exp = ((w16[:rows, :cols].astype(np.int32) + 2) >> 2).astype(np.uint8)
            assert np.array_equal(g[:rows, :cols], exp), i
            assert (g[rows:] == 0x5a).all() and (g[:, cols:] == 0x5a).all(), i
    assert hip.seed_state() == ora.seed_state()
    info = hip.last_launch_info()
    assert info["out8"] == 1 and info["listed"] == 1


def test_list_inside_an_overlap_region(hip):
    """Two lists inside one region run on the two internal streams; results as without the region"""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_ar_test1_10_420")
    a = [garbage_frame(520, 70, depth, sx, sy, 80 + i) for i in range(4)]
    b = [garbage_frame(520, 70, depth, sx, sy, 90 + i) for i in range(3)]
    want = [f.copy() for f in a + b]
    for w in want:
        ora.add_grain_frame(w)
    da, db = scattered(a, 8), scattered(b, 9)
    st = stream_ptr()
    hip.overlap_begin(st)
    hip.add_grain_frame_list_dev([d.ptrs() for d in da], 520, 70, a[0].stride, a[0].cstride, st)
    hip.add_grain_frame_list_dev([d.ptrs() for d in db], 520, 70, a[0].stride, a[0].cstride, st)
    hip.overlap_end(st)
    for i, (d, w) in enumerate(zip(da + db, want)):
        assert d.download().equal_all(w), i
    assert hip.seed_state() == ora.seed_state()


def test_refused_lists_change_nothing(hip):
    from versatilefilmgrain_amd.hw import VfgsHipError
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    frames = [garbage_frame(520, 70, depth, sx, sy, i) for i in range(3)]
    dev = scattered(frames)
    f0 = frames[0]
    st0 = hip.seed_state()
    ptrs = [d.ptrs() for d in dev]
    with pytest.raises(VfgsHipError, match="listed twice"):
        hip.add_grain_frame_list_dev([ptrs[0], ptrs[1], ptrs[0]], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    with pytest.raises(VfgsHipError, match="null plane"):
        hip.add_grain_frame_list_dev([ptrs[0], (ptrs[1][0], 0, ptrs[1][2])], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    # destination planes that share rows without being the same pointer (frame 1's luma begins 16 rows into frame 0's) ...
    with pytest.raises(VfgsHipError, match="overlap"):
        hip.add_grain_frame_list_dev([ptrs[0], (dev[0].ptrs(16)[0], ptrs[1][1], ptrs[1][2])], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    # ... and, out of place, a source that is ANOTHER frame's destination: one launch would read it grained or not by chance
    with pytest.raises(VfgsHipError, match="shares bytes"):
        hip.add_grain_frame_list_copy_dev([ptrs[0], ptrs[1]], [ptrs[1], ptrs[2]], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    # a part whose height wraps 32 bits is a part that exceeds the frame (it used to pass the check and select the whole frame)
    with pytest.raises(VfgsHipError, match="exceeds the frame"):
        hip.add_grain_frame_list_part_dev(ptrs, f0.width, f0.height, 16, 0xFFFFFFFF, f0.stride, f0.cstride, stream_ptr())
    with pytest.raises(VfgsHipError, match="16-byte aligned"):
        hip.add_grain_frame_list_dev([ptrs[0], (ptrs[1][0] + 8, ptrs[1][1], ptrs[1][2])], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    with pytest.raises(VfgsHipError, match="stride"):
        hip.add_grain_frame_list_dev(ptrs, f0.width, f0.height, 512, f0.cstride, stream_ptr())
    hip.add_grain_frame_list_dev([], f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())      # an empty list is no call at all
    assert hip.seed_state() == st0
    for d, f in zip(dev, frames):
        assert d.download().equal_all(f)
    # ... and the list is served afterwards
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    hip.add_grain_frame_list_dev(ptrs, f0.width, f0.height, f0.stride, f0.cstride, stream_ptr())
    for d, w in zip(dev, want):
        assert d.download().equal_all(w)
    assert hip.seed_state() == ora.seed_state()
