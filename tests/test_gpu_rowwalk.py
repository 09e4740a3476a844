"""GPU (MI355X): edge cases of the row-walk kernels (vfgs_kernel.hip) against the oracle, through the C ABI.

A wave streams whole rows in groups of four 1 KiB positions; the geometry that can go wrong is exactly where a row ends
relative to positions and groups (the position behind the last unit, rows that end on a position or group boundary, one-unit
tails), how many rows a wave walks (one, two or four, by picture size), stripes that begin or end inside a block row, frames
whose last block row is partial, batches, out-of-place copies, the widest row the parameter table holds (512 blocks) and rows
walked in several parts of 512 blocks (one block more, 1024, 1025, the widest picture the library takes).  Garbage in the stride
padding must survive."""
import numpy as np
import pytest

import vfgs_testlib as T
from gpu_util import DevFrame, stream_ptr

pytestmark = pytest.mark.gpu

FORMATS = ["fgs_sei_10_420", "fgs_afgs1_test1_8_420", "fgs_sei_8_420", "fgs_afgs1_test1_8_444", "fgs_sei_10_422", "fgs_sei_ff_test6_8_422", "fgs_sei_10_444",
           "fgs_sei_ff_test6_10_440", "fgs_sei_ff_test6_8_444"]
# 10-bit luma: a unit = 8 samples, a position = 512 samples, a group = 2048; 8-bit: twice that
WIDTHS = [136, 504, 512, 520, 1016, 1024, 1032, 2040, 2048, 2056, 2064, 4096, 4104]
HEIGHTS = [16, 17, 33, 64, 70]


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


def garbage_frame(width, height, depth, sx, sy, seed, stride=None, cstride=None):
    rng = np.random.default_rng(seed)
    f = T.Frame(width, height, depth, sx, sy, stride=stride, cstride=cstride)
    for p in f.planes():
        p[...] = rng.integers(0, 1 << (16 if depth > 8 else 8), p.shape).astype(f.dtype)
    return f


def device_pitches(f):
    """16-byte aligned rows: what the device entry points need (the reference's own strides are multiples of 64 samples)."""
    return f.stride, f.cstride


@pytest.mark.parametrize("name", FORMATS)
@pytest.mark.parametrize("width", WIDTHS)
def test_row_ends_on_and_around_position_and_group_boundaries(hip, name, width):
    ora, (depth, sx, sy) = program(hip, name)
    # (8-bit 4:2:x with an odd block count: chroma rows end in half a 16-byte unit, cut by the buffer range check per dword)
    for i, height in enumerate(HEIGHTS):
        f = garbage_frame(width, height, depth, sx, sy, width * 7 + i)
        want = f.copy()
        ora.add_grain_frame(want)
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        got = d.download()
        assert got.equal_all(want), (name, width, height)
        assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name,width,height", [("fgs_sei_10_420", 1920, 1080), ("fgs_afgs1_test1_8_420", 1920, 1080), ("fgs_sei_10_444", 720, 404),
                                               ("fgs_sei_10_422", 1280, 360), ("fgs_sei_10_420", 3840, 100)])
def test_stripes_that_cut_block_rows_at_every_rows_per_wave(hip, name, width, height):
    """Frames as sequences of device stripes whose edges fall anywhere inside block rows; small pictures walk two or four rows
    per wave, so first / last rows of a wave's walk get cut."""
    ora, (depth, sx, sy) = program(hip, name)
    f = garbage_frame(width, height, depth, sx, sy, height)
    want = f.copy()
    ora.add_grain_frame(want)
    d = DevFrame(f)
    cuts = [0, 2 * sy, 16, 16 + 6 * sy, 48, 50, 51 if sy == 1 else 52, height // 2 // sy * sy, height]
    cuts = sorted(set(c for c in cuts if 0 <= c <= height))
    for y0, y1 in zip(cuts, cuts[1:]):
        hip.add_grain_stripe_dev(*d.ptrs(y0), y0, f.width, y1 - y0, f.stride, f.cstride, stream_ptr())
    got = d.download()
    assert got.equal_all(want)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("width,parts", [(8192, 1), (8176, 1), (8208, 2)])
def test_widest_row_of_the_parameter_table_and_one_block_more(hip, width, parts):
    """512 blocks per row is the last width one parameter table holds; 513 blocks are walked in two parts."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f = garbage_frame(width, 48, depth, sx, sy, width)
    want = f.copy()
    ora.add_grain_frame(want)
    d = DevFrame(f)
    hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()
    info = hip.last_launch_info()
    wide_flag = info["kernel"].rstrip(">").split(",")[-2]       # grain_rw_kernel<depth, subx, suby, out8, oney, onec, WIDE, persist>
    assert info["parts_per_row"] == parts and wide_flag == ("true" if parts > 1 else "false"), info


WIDE_FORMATS = ["fgs_sei_10_420", "fgs_afgs1_test1_8_420", "fgs_sei_8_420", "fgs_afgs1_test1_8_444", "fgs_sei_10_422", "fgs_sei_10_444",
                "fgs_sei_ff_test6_10_440", "fgs_sei_ar_test1_10_420", "fgs_afgs1_test1_10_444", "fgs_sei_ff_test6_10_444", "fgs_afgs1_test3_10_422"]
# the form of the table image a wide picture gets (luma, chroma): at 4:2:0 and 4:4:4 one-pattern chroma keeps its form, and luma under
# it; every other case falls back to the general form (vfgs_host.cpp image_form, vfgs_kernel.hip launch_form)
WIDE_FORM = {"fgs_sei_10_420": (0, 1), "fgs_afgs1_test1_8_420": (1, 1), "fgs_sei_8_420": (0, 1), "fgs_afgs1_test1_8_444": (1, 1), "fgs_sei_10_422": (0, 0),
             "fgs_sei_10_444": (0, 1), "fgs_sei_ff_test6_10_440": (0, 0), "fgs_sei_ar_test1_10_420": (1, 1), "fgs_afgs1_test1_10_444": (1, 1),
             "fgs_sei_ff_test6_10_444": (0, 0), "fgs_afgs1_test3_10_422": (0, 0)}


@pytest.mark.parametrize("name", WIDE_FORMATS)
@pytest.mark.parametrize("width", [8208, 12288, 16384, 16400])
def test_rows_walked_in_parts(hip, name, width):
    """Rows of 513, 768, 1024 and 1025 blocks (two or three passes over the parameter table): every chroma format, both depths,
    one-pattern configurations (which keep their form at 4:2:0 and 4:4:4), a stripe that ends inside a block row, two frames per
    launch through the stripe + frame entry points.  The table is refilled between two barriers while the ring of register sets runs on."""
    ora, (depth, sx, sy) = program(hip, name)
    for height in (16, 40):
        f = garbage_frame(width, height, depth, sx, sy, width + height)
        want = f.copy()
        ora.add_grain_frame(want)
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        assert d.download().equal_all(want), (name, width, height)
        assert hip.seed_state() == ora.seed_state()
    info = hip.last_launch_info()
    assert info["parts_per_row"] == (width + 15) // 16 // 512 + ((width + 15) // 16 % 512 > 0)
    assert (info["one_y"], info["one_c"]) == WIDE_FORM[name], info
    assert info["rows_per_wave"] == [1, 1]
    # the LDS the kernel allocates (vfgs_layout.h lds_allocation): the 10-bit all-one-pattern kernels pad to four workgroups per CU
    if depth == 10 and WIDE_FORM[name] == (1, 1):
        assert info["lds_bytes_per_workgroup"] == 40960, info
    else:
        assert 15000 < info["lds_bytes_per_workgroup"] <= 40960, info


def test_widest_picture(hip):
    """31744 samples = 1984 blocks: four parts."""
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")
    f = garbage_frame(31744, 32, depth, sx, sy, 9)
    want = f.copy()
    ora.add_grain_frame(want)
    d = DevFrame(f)
    hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()
    assert hip.last_launch_info()["parts_per_row"] == 4


def test_wide_and_narrow_pictures_alternate_with_one_pattern_configuration(hip):
    """A one-pattern configuration at 4:2:2: pictures of up to 8192 samples use the one-pattern table image, wider ones the general
    form (the one-pattern kernels of rows walked in parts exist at 4:2:0 and 4:4:4) -- the image is rebuilt when the width class
    changes, in both directions, without a setter call in between."""
    ora, (depth, sx, sy) = program(hip, "fgs_afgs1_test3_10_422")
    for i, width in enumerate([1024, 8208, 2048, 16384, 8192]):
        f = garbage_frame(width, 32, depth, sx, sy, 50 + i)
        want = f.copy()
        ora.add_grain_frame(want)
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        assert d.download().equal_all(want), width
        assert hip.seed_state() == ora.seed_state()
        info = hip.last_launch_info()
        assert (info["one_y"], info["one_c"]) == ((1, 1) if width <= 8192 else (0, 0)), (width, info)


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_afgs1_test1_8_444"])
def test_batches_with_frame_pitch_and_out_of_place(hip, name):
    """5 frames in one launch (frame pitch larger than a frame), in place and as a copy into a second set of buffers."""
    import torch
    ora, (depth, sx, sy) = program(hip, name)
    W, H, N = 1032, 90, 5
    frames = [garbage_frame(W, H, depth, sx, sy, 100 + i) for i in range(N)]
    want = [f.copy() for f in frames]
    for w in want:
        ora.add_grain_frame(w)
    sz = frames[0].Y.itemsize
    f0 = frames[0]
    ypitch = (f0.Y.size * sz + 256 + 15) // 16 * 16
    cpitch = (f0.U.size * sz + 512 + 15) // 16 * 16

    def upload():
        Y = torch.zeros(N * ypitch, dtype=torch.uint8, device="cuda"); U = torch.zeros(N * cpitch, dtype=torch.uint8, device="cuda"); V = torch.zeros(N * cpitch, dtype=torch.uint8, device="cuda")
        for i, f in enumerate(frames):
            Y[i * ypitch:i * ypitch + f.Y.size * sz] = torch.from_numpy(f.Y.view(np.uint8).ravel().copy()).cuda()
            U[i * cpitch:i * cpitch + f.U.size * sz] = torch.from_numpy(f.U.view(np.uint8).ravel().copy()).cuda()
            V[i * cpitch:i * cpitch + f.V.size * sz] = torch.from_numpy(f.V.view(np.uint8).ravel().copy()).cuda()
        return Y, U, V

    def check(Y, U, V, cols=None):
        """cols: compare only the first cols luma columns (an out-of-place destination's stride padding is never written)"""
        torch.cuda.synchronize()
        cy = cols or f0.stride
        cc = (cols // sx) if cols else f0.cstride
        for i, w in enumerate(want):
            gy = Y[i * ypitch:i * ypitch + w.Y.size * sz].cpu().numpy().view(w.dtype).reshape(w.Y.shape)
            gu = U[i * cpitch:i * cpitch + w.U.size * sz].cpu().numpy().view(w.dtype).reshape(w.U.shape)
            gv = V[i * cpitch:i * cpitch + w.V.size * sz].cpu().numpy().view(w.dtype).reshape(w.V.shape)
            # rows the picture has (the allocation's padding rows below it are never touched by either side)
            assert np.array_equal(gy[:H, :cy], w.Y[:H, :cy]) and np.array_equal(gu[:H // sy, :cc], w.U[:H // sy, :cc]) and np.array_equal(gv[:H // sy, :cc], w.V[:H // sy, :cc]), i

    Y, U, V = upload()
    hip.add_grain_frames_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), W, H, f0.stride, f0.cstride, N, ypitch, cpitch, stream_ptr())
    check(Y, U, V)
    st = hip.seed_state()
    assert st == ora.seed_state()
    # the same batch again, out of place, from a fresh copy of the inputs and the same seed state as before
    ora2, _ = program(hip, name)
    sY, sU, sV = upload()
    dY, dU, dV = torch.zeros_like(sY), torch.zeros_like(sU), torch.zeros_like(sV)
    hip.add_grain_copy_dev(sY.data_ptr(), sU.data_ptr(), sV.data_ptr(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(), W, H, 0, H,
                           f0.stride, f0.cstride, N, ypitch, cpitch, stream_ptr())
    check(dY, dU, dV, cols=(W + 15) // 16 * 16)
    assert not dY.view(torch.int16 if depth > 8 else torch.uint8).reshape(N, -1)[:, :f0.Y.size].reshape(N, -1, f0.stride)[:, :H, (W + 15) // 16 * 16:].any()   # ... and stays untouched
    assert hip.seed_state() == st


@pytest.mark.parametrize("name", ["fgs_afgs1_test1_8_420", "fgs_sei_8_420", "fgs_sei_ff_test6_8_422"])
@pytest.mark.parametrize("width", [144, 720, 3856, 8176])
def test_odd_block_counts_at_8_bit_420_and_422(hip, name, width):
    """8-bit 4:2:x rows of an odd number of blocks (SD video: 720 = 45 blocks) end in HALF a 16-byte unit.  The row walk leaves
    that to the raw-buffer range check (a row's descriptor holds exactly the row's bytes; checked per dword): the samples and
    the garbage in the stride padding behind them must come out as the reference leaves them (whole blocks).
    8176 = 511 blocks: the widest odd row the parameter table holds."""
    ora, (depth, sx, sy) = program(hip, name)
    assert ((width + 15) // 16) % 2 == 1
    for i, height in enumerate([16, 33, 70] if width < 4000 else [33]):
        f = garbage_frame(width, height, depth, sx, sy, width + i)
        want = f.copy()
        ora.add_grain_frame(want)
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        assert d.download().equal_all(want), (name, width, height)
        assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_afgs1_test1_8_420", "fgs_sei_8_420"])
def test_rows_wider_than_the_parameter_table_with_odd_block_count(hip, name):
    """513 blocks (8208 samples) at 8-bit 4:2:0: one block more than the parameter table holds (two parts) AND chroma rows that end
    in half a 16-byte unit."""
    ora, (depth, sx, sy) = program(hip, name)
    f = garbage_frame(8208, 33, depth, sx, sy, 4)
    want = f.copy()
    ora.add_grain_frame(want)
    d = DevFrame(f)
    hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
    assert d.download().equal_all(want)
    assert hip.seed_state() == ora.seed_state()


def copy8_case(hip, ora, f, nframes=1):
    """f: 10-bit frame(s) with garbage; runs the fused 8-bit output into 0x5a-filled planes and checks every byte of them: whole
    grain blocks of the picture's rows hold (grained + 2) >> 2, everything else still 0x5a; the source is untouched."""
    import torch
    frames = f if isinstance(f, list) else [f]
    f0 = frames[0]
    sx, sy, w, h = f0.subx, f0.suby, f0.width, f0.height
    want = [x.copy() for x in frames]
    for x in want:
        ora.add_grain_frame(x)
    f8 = T.Frame(w, h, 8, sx, sy)
    Y = torch.from_numpy(np.stack([x.Y for x in frames]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([x.U for x in frames]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([x.V for x in frames]).view(np.uint8)).cuda()
    n = len(frames)
    dY = torch.full((n,) + f8.Y.shape, 0x5a, dtype=torch.uint8, device="cuda")
    dU = torch.full((n,) + f8.U.shape, 0x5a, dtype=torch.uint8, device="cuda")
    dV = torch.full((n,) + f8.V.shape, 0x5a, dtype=torch.uint8, device="cuda")
    hip.add_grain_copy8_dev(Y.data_ptr(), U.data_ptr(), V.data_ptr(), dY.data_ptr(), dU.data_ptr(), dV.data_ptr(), w, h, 0, h,
                            f0.stride, f0.cstride, f8.stride, f8.cstride, n, Y[0].numel(), U[0].numel(), dY[0].numel(), dU[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    nblk = (w + 15) // 16
    for i, wf in enumerate(want):
        crows = (h + sy - 1) // sy        # (an odd number of lines at 4:2:0: the last line still has its chroma row, vfgs_main.c:672-681)
        for got, w16, rows, cols in ((dY[i], wf.Y, h, nblk * 16), (dU[i], wf.U, crows, nblk * 16 // sx), (dV[i], wf.V, crows, nblk * 16 // sx)):
            g = got.cpu().numpy()
            # samples > 1023 exist in the garbage: the reference's yuv_to_8bit stores the low byte of (v + 2) >> 2
            exp = ((w16[:rows, :cols].astype(np.int32) + 2) >> 2).astype(np.uint8)
            assert np.array_equal(g[:rows, :cols], exp), (i, w, h)
            assert (g[rows:] == 0x5a).all() and (g[:, cols:] == 0x5a).all(), (i, w, h)
    for x, t in zip(frames, Y):
        assert np.array_equal(t.cpu().numpy().view(np.uint16).reshape(x.Y.shape), x.Y)
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_ar_test1_10_420", "fgs_afgs1_test1_10_420", "fgs_sei_10_422", "fgs_sei_10_444", "fgs_sei_ff_test6_10_440",
                                  "fgs_sei_ff_test1_10_420"])
@pytest.mark.parametrize("width", [136, 504, 512, 520, 1016, 1032, 2048, 2056, 4104])
def test_fused_8bit_output_row_geometry(hip, name, width):
    """The narrowed destination on the row walk (general and one-pattern forms, every 10-bit chroma format): rows that end on and
    around position and group boundaries, heights that end inside a block row, destination padding untouched."""
    ora, (depth, sx, sy) = program(hip, name)
    assert depth == 10
    for i, height in enumerate([16, 33, 70]):
        f = garbage_frame(width, height, depth, sx, sy, width * 3 + i)
        # (garbage above 10 bit: out8 of such a sample wraps like the reference's uint8 store, yuv.c:216-258 -- but the kernel's
        # "+2 >> 2 per 16-bit half" must not carry into the neighbour: keep samples <= 0xfffd so the test states what is defined)
        for p in f.planes():
            np.minimum(p, 0xfffd, out=p)
        copy8_case(hip, ora, f)
    info = hip.last_launch_info()
    assert info["out8"] == 1 and info["kernel"].startswith("grain_rw_kernel<10,")


@pytest.mark.parametrize("name,width", [("fgs_sei_10_420", 8208), ("fgs_sei_ar_test1_10_420", 16384), ("fgs_sei_10_444", 8208)])
def test_fused_8bit_output_wide_rows(hip, name, width):
    ora, (depth, sx, sy) = program(hip, name)
    f = garbage_frame(width, 40, depth, sx, sy, width)
    for p in f.planes():
        np.minimum(p, 0xfffd, out=p)
    copy8_case(hip, ora, f)
    assert hip.last_launch_info()["parts_per_row"] > 1


@pytest.mark.parametrize("name,w,h", [("fgs_sei_10_420", 1920, 1080), ("fgs_sei_ff_test1_10_420", 1920, 1080), ("fgs_sei_ar_test1_10_420", 3840, 2160)])
def test_fused_8bit_output_full_size_batches(hip, name, w, h):
    """The timed shapes of the fused 8-bit output: 1080p (narrow chroma rows: two positions) and 2160p, several frames per launch."""
    ora, (depth, sx, sy) = program(hip, name)
    frames, _ = T.lcg_frames(w, h, depth, sx, sy, 3)
    copy8_case(hip, ora, frames)


def batch_case(hip, ora, frames, part=None, out_of_place=False):
    """frames (equal geometry) in ONE launch through the batch entry points; every plane byte against the oracle."""
    import torch
    f0 = frames[0]
    n, sz = len(frames), f0.Y.itemsize
    want = [x.copy() for x in frames]
    for x in want:
        ora.add_grain_frame(x)
    Y = torch.from_numpy(np.stack([x.Y for x in frames]).view(np.uint8)).cuda()
    U = torch.from_numpy(np.stack([x.U for x in frames]).view(np.uint8)).cuda()
    V = torch.from_numpy(np.stack([x.V for x in frames]).view(np.uint8)).cuda()
    dY, dU, dV = (torch.zeros_like(Y), torch.zeros_like(U), torch.zeros_like(V)) if out_of_place else (Y, U, V)
    py, ph = part if part else (0, f0.height)
    cy = py // f0.suby
    yo, co = py * f0.stride * sz, cy * f0.cstride * sz
    if out_of_place:
        hip.add_grain_copy_dev(Y.data_ptr() + yo, U.data_ptr() + co, V.data_ptr() + co, dY.data_ptr() + yo, dU.data_ptr() + co, dV.data_ptr() + co,
                               f0.width, f0.height, py, ph, f0.stride, f0.cstride, n, Y[0].numel(), U[0].numel(), stream_ptr())
    else:
        hip.add_grain_frames_part_dev(Y.data_ptr() + yo, U.data_ptr() + co, V.data_ptr() + co, f0.width, f0.height, py, ph, f0.stride, f0.cstride,
                                      n, Y[0].numel(), U[0].numel(), stream_ptr())
    torch.cuda.synchronize()
    gy, gu, gv = (t.cpu().numpy().view(f0.dtype) for t in (dY, dU, dV))
    c0, c1 = py // f0.suby, (py + ph + f0.suby - 1) // f0.suby
    # (out of place only whole grain blocks reach the destination; in place the stride padding keeps the caller's bytes)
    yc = (f0.width + 15) // 16 * 16 if out_of_place else f0.stride
    cc = yc // f0.subx if out_of_place else f0.cstride
    for i, wf in enumerate(want):
        assert np.array_equal(gy[i].reshape(wf.Y.shape)[py:py + ph, :yc], wf.Y[py:py + ph, :yc]), i
        assert np.array_equal(gu[i].reshape(wf.U.shape)[c0:c1, :cc], wf.U[c0:c1, :cc]), i
        assert np.array_equal(gv[i].reshape(wf.V.shape)[c0:c1, :cc], wf.V[c0:c1, :cc]), i
        if out_of_place:
            assert not gy[i].reshape(wf.Y.shape)[:, yc:].any() and not gu[i].reshape(wf.U.shape)[:, cc:].any()
        if not out_of_place and part:        # rows outside the part are untouched
            assert np.array_equal(gy[i].reshape(wf.Y.shape)[:py], frames[i].Y[:py]) and np.array_equal(gy[i].reshape(wf.Y.shape)[py + ph:], frames[i].Y[py + ph:])
    assert hip.seed_state() == ora.seed_state()


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_10_444", "fgs_sei_10_422", "fgs_sei_10_440"])
@pytest.mark.parametrize("w,h,n", [(384, 224, 230), (1000, 70, 640), (136, 33, 1100)])
def test_persistent_luma_workgroups(hip, name, w, h, n):
    """General-form luma, launches of several rounds of luma workgroups: P persistent workgroups share the luma tasks (one staging
    of the table image each, the block parameters per task).  Hundreds of small frames per launch reach that regime with little
    data: tasks of every frame position, a last block row that is partial (70 = 4 x 16 + 6 lines, 33 = 2 x 16 + 1), workgroups
    whose last task index falls behind the end, odd heights at 4:2:0; in place, as a stripe of every frame, and out of place."""
    ora, (depth, sx, sy) = program(hip, name)
    frames = [garbage_frame(w, h, depth, sx, sy, 1000 + i) for i in range(n)]
    batch_case(hip, ora, frames)
    info = hip.last_launch_info()
    assert info["persistent_luma_workgroups"] > 0 and info["kernel"].endswith(",true>"), info
    assert info["persistent_luma_workgroups"] <= 4 * hip.device_info()["cu_count"]
    if h >= 64:
        ora2, _ = program(hip, name)
        batch_case(hip, ora2, frames, part=(16, 32))
        ora3, _ = program(hip, name)
        batch_case(hip, ora3, frames, out_of_place=True)
        assert hip.last_launch_info()["persistent_luma_workgroups"] > 0


def test_persistence_is_for_general_form_luma_of_small_pictures_only(hip):
    ora, (depth, sx, sy) = program(hip, "fgs_afgs1_test1_8_420")        # one-pattern luma: never
    batch_case(hip, ora, [garbage_frame(384, 224, depth, sx, sy, i) for i in range(230)])
    assert hip.last_launch_info()["persistent_luma_workgroups"] == 0
    ora, (depth, sx, sy) = program(hip, "fgs_sei_10_420")               # few tasks: never
    batch_case(hip, ora, [garbage_frame(384, 224, depth, sx, sy, i) for i in range(8)])
    assert hip.last_launch_info()["persistent_luma_workgroups"] == 0
    ora, (depth, sx, sy) = program(hip, "fgs_sei_8_420")                # 8 bit: the general-form kernels are bound by LDS instructions, not by
    batch_case(hip, ora, [garbage_frame(384, 224, depth, sx, sy, i) for i in range(230)])   # the staging: -7 % with persistence (profiles/r04_ab2)
    assert hip.last_launch_info()["persistent_luma_workgroups"] == 0


@pytest.mark.parametrize("name", ["fgs_sei_10_420", "fgs_sei_10_444"])
def test_fused_8bit_output_with_persistent_luma_workgroups(hip, name):
    ora, (depth, sx, sy) = program(hip, name)
    frames = [garbage_frame(384, 224, depth, sx, sy, 300 + i) for i in range(230)]
    for f in frames:
        for p in f.planes():
            np.minimum(p, 0xfffd, out=p)
    copy8_case(hip, ora, frames)
    info = hip.last_launch_info()
    assert info["out8"] == 1 and info["persistent_luma_workgroups"] > 0, info


def test_fuzz_fused_8bit_output_and_wide_rows(hip):
    """Randomised geometry for the paths that moved onto the row walk in round 4: the fused 8-bit output (any width / height /
    10-bit format, one or several frames) and rows walked in parts (random widths above 8192 samples, both depths)."""
    rng = np.random.default_rng(20241004)
    names10 = ["fgs_sei_10_420", "fgs_sei_10_422", "fgs_sei_10_444", "fgs_sei_10_440", "fgs_sei_ar_test1_10_420", "fgs_afgs1_test1_10_420"]
    for it in range(24):
        name = names10[it % len(names10)]
        ora, (depth, sx, sy) = program(hip, name)
        w = int(rng.integers(130, 4200))
        h = int(rng.integers(16, 120))
        n = int(rng.integers(1, 4))
        frames = [garbage_frame(w, h, depth, sx, sy, 7000 + 10 * it + i) for i in range(n)]
        for f in frames:
            for p in f.planes():
                np.minimum(p, 0xfffd, out=p)
        copy8_case(hip, ora, frames)
    wide = ["fgs_sei_10_420", "fgs_sei_8_420", "fgs_afgs1_test1_8_444", "fgs_sei_10_444", "fgs_afgs1_test1_8_420", "fgs_sei_ff_test6_8_422"]
    for it in range(12):
        name = wide[it % len(wide)]
        ora, (depth, sx, sy) = program(hip, name)
        w = int(rng.integers(8193, 24000))
        h = int(rng.integers(16, 50))
        f = garbage_frame(w, h, depth, sx, sy, 9000 + it)
        want = f.copy()
        ora.add_grain_frame(want)
        d = DevFrame(f)
        hip.add_grain_frame_dev(*d.ptrs(), f.width, f.height, f.stride, f.cstride, stream_ptr())
        assert d.download().equal_all(want), (name, w, h)
        assert hip.seed_state() == ora.seed_state()
        assert hip.last_launch_info()["parts_per_row"] == ((w + 15) // 16 + 511) // 512
