"""GPU (MI355X): argument checking and the size limits of the device entry points.
Errors are returned (never a silent no-op), nothing is processed and the seed registers do not
move when a call is refused; empty calls are accepted and do nothing (vfgs_hw.c has no
equivalent: its asserts vanish under NDEBUG, SURVEY 8b "Errors")."""
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


def program(hip, name="fgs_sei_10_420"):
    hip.lib.vfgs_hip_reset_state()
    rec = T.load_trace(name)
    T.replay(hip, rec)
    ora = T.OracleHW()
    T.replay(ora, rec)
    return ora


def rc_of(hip, fn, *args):
    return getattr(hip.lib, fn)(*args)


def test_refused_calls_return_codes_and_leave_state_alone(hip):
    import torch
    program(hip)
    buf = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    assert p % 256 == 0
    before = hip.seed_state()
    L = hip.lib
    cases = [
        (5, L.vfgs_hip_add_grain_frame_dev(p, p, p, 128, 64, 128, 64, None)),            # width <= 128 (vfgs_hw.c:168)
        (7, L.vfgs_hip_add_grain_frame_dev(p + 2, p, p, 192, 64, 192, 96, None)),        # pointer alignment
        (6, L.vfgs_hip_add_grain_frame_dev(p, p, p, 200, 64, 200, 100, None)),           # stride < whole blocks
        (8, L.vfgs_hip_add_grain_frame_dev(p, p, p, 200, 64, 212, 108, None)),           # row pitch not a multiple of 16 bytes
        (11, L.vfgs_hip_add_grain_frame_part_dev(p, p, p, 192, 64, 8, 16, 192, 96, None)),   # part not on a block row
        (12, L.vfgs_hip_add_grain_frame_part_dev(p, p, p, 192, 64, 48, 32, 192, 96, None)),  # part leaves the frame
        (13, L.vfgs_hip_add_grain_frames_dev(p, p, p, 192, 64, 192, 96, 2, 24584, 6144, None)),  # frame pitch alignment
        (17, L.vfgs_hip_add_grain_frame_dev(p, p, p, 31760, 16, 31808, 15904, None)),    # wider than the LFSR slice a wave keeps
        (15, L.vfgs_hip_add_grain_frame_dev(p, p, p, 16384, 70000, 16384, 8192, None)),  # plane stripe >= 2 GiB
    ]
    for want, got in cases:
        assert got == want, (want, got, L.vfgs_hip_last_error_string())
    assert hip.seed_state() == before
    assert torch.count_nonzero(buf).item() == 0
    # a pattern LUT that selects the reference's out-of-bounds slot 9.. is refused (vfgs_hw.c:49,212)
    lut = bytearray(256)
    lut[7] = 9 << 4
    hip.set_pattern_lut(0, bytes(lut))
    assert L.vfgs_hip_add_grain_frame_dev(p, p, p, 192, 64, 192, 96, None) == 4
    assert hip.seed_state() == before
    # 8-bit output needs a 10-bit path
    hip.lib.vfgs_hip_reset_state()
    hip.set_depth(8)
    assert L.vfgs_hip_add_grain_copy8_dev(p, p, p, p, p, p, 192, 64, 0, 64, 192, 96, 192, 96, 1, 0, 0, 0, 0, None) == 16


def test_empty_calls_do_nothing(hip):
    import torch
    ora = program(hip)
    buf = torch.full((1 << 20,), 7, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    before = hip.seed_state()
    assert hip.lib.vfgs_hip_add_grain_stripe_dev(p, p, p, 32, 192, 0, 192, 96, None) == 0      # no lines
    assert hip.lib.vfgs_hip_add_grain_frames_dev(p, p, p, 192, 64, 192, 96, 0, 0, 0, None) == 0  # no frames
    host = np.full(4096, 9, dtype=np.uint16)
    hip.add_grain_stripe(host.ctypes.data, host.ctypes.data, host.ctypes.data, 0, 192, 0, 192, 96)
    torch.cuda.synchronize()
    assert hip.seed_state() == before == ora.seed_state()
    assert torch.all(buf == 7).item() and np.all(host == 9)


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_444"])
def test_widest_supported_picture(hip, name):
    """31744 samples = 1984 blocks per line: the row's LFSR windows just fit the 64-dword slice."""
    from gpu_util import DevFrame, stream_ptr
    ora = program(hip, name)
    depth, sx, sy = T.trace_geometry(T.load_trace(name))
    f, _ = T.lcg_frames(31744, 48, depth, sx, sy, 1)
    want = f[0].copy()
    d = DevFrame(f[0])
    hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f[0].width, f[0].height, f[0].stride, f[0].cstride, stream_ptr())
    ora.add_grain_frame(want)
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


def test_narrowest_supported_picture_and_tall_stripe(hip):
    """width 130 (just above the 128 of vfgs_hw.c:168; 9 blocks, the last one almost entirely in the
    stride padding) x 4400 lines: one tile per row, 275 block rows."""
    from gpu_util import DevFrame, stream_ptr
    ora = program(hip, "fgs_sei_ff_test6_10_420")
    f, _ = T.lcg_frames(130, 4400, 10, 2, 2, 1)
    want = f[0].copy()
    d = DevFrame(f[0])
    hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), 130, 4400, f[0].stride, f[0].cstride, stream_ptr())
    ora.add_grain_frame(want)
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


def test_calls_alternating_between_two_streams(hip):
    """The library state is one singleton; callers may still spread their launches over streams (and
    synchronise the FRAMES themselves).  Table and LFSR images uploaded on one stream must be complete
    before a kernel on the other stream reads them: configuration changes between frames, frames
    alternate between two streams, every frame is checked."""
    import torch
    from gpu_util import DevFrame
    names = ["fgs_sei_10_420", "fgs_sei_ff_test6_10_420", "fgs_afgs1_test1_10_420", "fgs_sei_ar_test1_10_420"] * 3
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    frames, _ = T.lcg_frames(704, 400, 10, 2, 2, len(names))
    devs = [DevFrame(f) for f in frames]
    torch.cuda.synchronize()
    hip.lib.vfgs_hip_reset_state()
    ora = T.OracleHW()
    for i, (name, d, f) in enumerate(zip(names, devs, frames)):
        rec = T.load_trace(name)
        T.replay(hip, rec)           # new banks / LUTs / seed before every frame
        T.replay(ora, rec)
        st = streams[i & 1]
        hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, st.cuda_stream)
        ora.add_grain_frame(f)
    torch.cuda.synchronize()
    for i, (d, f) in enumerate(zip(devs, frames)):
        assert d.download().equal_all(f), i
    assert hip.seed_state() == ora.seed_state()
    # and without changes in between: the images uploaded for the first frame (stream 0) are read by the
    # second frame's kernel on stream 1 -- which has to wait for that upload
    frames, _ = T.lcg_frames(704, 400, 10, 2, 2, 6, state=99)
    devs = [DevFrame(f) for f in frames]
    rec = T.load_trace("fgs_sei_10_420")
    T.replay(hip, rec)
    T.replay(ora, rec)
    torch.cuda.synchronize()
    for i, (d, f) in enumerate(zip(devs, frames)):
        hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), f.width, f.height, f.stride, f.cstride, streams[i & 1].cuda_stream)
        ora.add_grain_frame(f)
    torch.cuda.synchronize()
    for i, (d, f) in enumerate(zip(devs, frames)):
        assert d.download().equal_all(f), i


@pytest.mark.parametrize("name,lowest", [("fgs_afgs1_test1_8_420", -127), ("fgs_afgs1_test1_8_420", -128), ("fgs_afgs1_test1_10_420", -127),
                                         ("fgs_afgs1_test1_10_420", -128), ("fgs_afgs1_test1_8_444", -128), ("fgs_sei_10_420", -128)])
def test_one_pattern_form_with_extreme_pattern_values(hip, name, lowest):
    """The one-pattern form keeps a negated copy of the pattern (vfgs_layout.h): full-range values +-127 must survive the negation,
    the edge filter and the overlap blend; a pattern that holds -128 (no negation in a byte; possible through vfgs_set_*_pattern,
    vfgs_hw.c:314-325, never out of the firmware) makes the host fall back to the general form.  Either way: the oracle's output."""
    from gpu_util import DevFrame, stream_ptr
    ora = program(hip, name)
    depth, sx, sy = T.trace_geometry(T.load_trace(name))
    rng = np.random.default_rng(5)
    for k in range(2):                      # slots 0 and 1 of both banks: mostly extremes
        for setter in ("set_luma_pattern", "set_chroma_pattern"):
            P = rng.choice(np.array([lowest, lowest, 127, 127, -1, 0, 1, 64], dtype=np.int8), size=4096).astype(np.int8)
            getattr(hip, setter)(k, P.tobytes())
            getattr(ora, setter)(k, P.tobytes())
    f, _ = T.lcg_frames(1936, 112, depth, sx, sy, 2)
    for fr in f:
        want = fr.copy()
        d = DevFrame(fr)
        hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), fr.width, fr.height, fr.stride, fr.cstride, stream_ptr())
        ora.add_grain_frame(want)
        assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_afgs1_test1_8_444", "fgs_afgs1_test1_8_420", "fgs_sei_8_422", "fgs_afgs1_test1_8_440"])
@pytest.mark.parametrize("shift", [2, 5, 7])
def test_packed_16_bit_form_at_its_limit(hip, name, shift):
    """8-bit one-pattern components multiply pattern and scale in 16 bits, two samples per instruction (vfgs_layout.h "packed
    16-bit form"; vfgs_hw.c:263): exact while max(scale) * 127 + 2^(shift-1) <= 32767, and the host checks exactly that when it
    digests the LUT -- a LUT beyond it keeps the general form.  Patterns of +-127 everywhere, scale LUTs whose maximum sits at
    the limit, one below and one above it and at 255, for the smallest, the usual and the largest shift: always the oracle's bytes,
    and the kernel name says which form ran."""
    from gpu_util import DevFrame, stream_ptr
    depth, sx, sy = T.trace_geometry(T.load_trace(name))
    eff = shift + 6                                   # vfgs_hw.c:349 at 8 bit
    limit = (32767 - (1 << (eff - 1))) // 127         # the largest scale the 16-bit product can hold
    rng = np.random.default_rng(shift)
    for top in sorted({min(limit, 255), min(limit, 255) - 1, min(limit + 1, 255), 255}):
        ora = program(hip, name)
        for obj in (hip, ora):
            obj.set_scale_shift(shift)
        P = rng.choice(np.array([-127, 127, 127, -127, 126, -1, 0, 1], dtype=np.int8), size=4096).astype(np.int8)
        lut = rng.integers(0, top + 1, 256).astype(np.uint8)
        lut[rng.integers(0, 256, 40)] = top           # the maximum is met often, by any intensity
        for obj in (hip, ora):
            for k in range(2):
                obj.set_luma_pattern(k, P.tobytes())
                obj.set_chroma_pattern(k, P[::-1].tobytes())
            for c in range(3):
                obj.set_scale_lut(c, lut.tobytes())
        f, _ = T.lcg_frames(1936, 112, depth, sx, sy, 2, state=top)
        for fr in f:
            want = fr.copy()
            d = DevFrame(fr)
            hip.add_grain_frame_dev(d.Y.data_ptr(), d.U.data_ptr(), d.V.data_ptr(), fr.width, fr.height, fr.stride, fr.cstride, stream_ptr())
            ora.add_grain_frame(want)
            assert d.download().equal_all(want), (top, limit)
        info = hip.last_launch_info()
        one_pattern = "fgs_sei" not in name           # (the default SEI model: eight luma patterns, one chroma pattern)
        assert info["one_c"] == (top <= limit) and info["one_y"] == (one_pattern and top <= limit), (top, limit, info["kernel"])
        assert hip.seed_state() == ora.seed_state()
