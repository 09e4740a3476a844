#!/usr/bin/env python3
"""Benchmark of the hot path (grain synthesis over Y/Cb/Cr) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by
torch.distributed.run, one rank per GPU).  Prints ONE JSON line on rank 0.

Workload (BASELINE.json `metric` / configs[4]): 7680x4320 10-bit 4:2:0, cfg fgs_sei
(8 luma patterns, per-sample pattern selection), seed 12345, synthetic frames of uniformly
random 10-bit samples generated on the GPU and resident in HBM before the timed region.

A "step" processes `N x batch` frames (default batch 8): every frame is split into N stripes of
whole 16-line block rows, rank r owns stripe r of every frame (no collective, no halo --
DESIGN.md "multi-GPU") and runs them in ONE launch, so per-GPU work per step is `batch`
frames' worth at every N ("weak").  At N=1 a step is `batch` whole frames in one launch
(`--batch 1` = one launch per frame; both are reported in DESIGN.md).

`value`   = luma pixels of all frames of all ranks / wall time          [Mpixels/s]
`roofline`= algorithmic bytes per launch (4 B per Y/Cb/Cr sample: one read + one write of
            2 bytes) / mean launch duration from HIP events on the launching stream.
`cpu_baseline` = the REAL reference hardware layer (oracle/_ref/libvfgs_ref.so, kind
            "reference") if that prebuilt file travelled to this box, else this repo's oracle
            ("port"), one host core, on a bounded number of frames of the same workload.
`parity_checked` = after the timed region one more launch of exactly the timed shape (same entry point, same frames per
            launch, in place) runs on freshly generated frames from a reset seed state, and every frame of it is compared
            with the oracle on the host (each rank its own stripes); true only if every byte and the seed registers agree.

`configs` (N=1 only) = BASELINE.json configs[0..3] -- 1080p fgs_sei / ff_test1, 2160p ar_test1, 2160p 8-bit 4:4:4 AFGS1 -- each at the
            stated frames per launch: launch duration (HIP events, plain stream), fraction of the 8 TB/s peak, the kernel the
            library reports having dispatched (vfgs_hip_last_launch_info), and a parity check of that shape against the oracle.
A parity failure anywhere withholds the numbers: the line then carries `error` instead of `value`, and the exit code is 1.

`--gpus N` without a launcher around it (WORLD_SIZE unset): this process starts `python -m torch.distributed.run` with N
ranks as a CHILD process -- before anything touches the GPU -- and relays its output and exit code.
`--scaling weak` (default): a step = N x batch frames, per-GPU work constant in N.  `--scaling strong`: a step = batch
frames whatever N is, every frame's block rows split over the N ranks (the latency view: one frame's stripes on N GPUs).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

W, H, DEPTH, SUBX, SUBY = 7680, 4320, 10, 2, 2
TRACE = "fgs_sei_10_420"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); float4 copy reaches 6.29 TB/s


def split_rows(nbr, n):
    """nbr block rows -> n contiguous parts, sizes differ by at most one."""
    base, extra = divmod(nbr, n)
    out, r = [], 0
    for i in range(n):
        k = base + (1 if i < extra else 0)
        out.append((r, k))
        r += k
    return out


def cpu_baseline(seconds=10.0, trace=TRACE, w=W, h=H, depth=DEPTH, sx=SUBX, sy=SUBY):
    """Reference (or oracle) hot path on ONE host core over whole frames of the given workload (default: the headline's)."""
    import numpy as np
    import vfgs_testlib as T

    rec = T.load_trace(trace)
    lib = T.oracle_lib()
    frame, _ = T.lcg_frames(w, h, depth, sx, sy, 1)
    f = frame[0]
    note = None
    if T.have_reference():
        kind, hw, timed_lib = "reference", T.ReferenceHW(), str(T.REF_SO.relative_to(ROOT))
        T.replay(hw, rec)
        line = C.cast(hw.lib.vfgs_add_grain_line, C.c_void_p)

        def run(fr):
            lib.vfgs_oracle_drive_lines(line, C.c_void_p(fr.Y.ctypes.data), C.c_void_p(fr.U.ctypes.data),
                                        C.c_void_p(fr.V.ctypes.data), fr.width, fr.height, fr.stride, fr.cstride, fr.Y.itemsize, sy)
    else:
        kind, hw, timed_lib = "port", T.OracleHW(), str(T.ORACLE_SO.relative_to(ROOT))
        note = "oracle/_ref/libvfgs_ref.so (the reference compiled in the build container) did not travel to this box: the repo's CPU restatement was timed instead"
        T.replay(hw, rec)

        def run(fr):
            hw.add_grain_frame(fr)
    pristine = [p.copy() for p in f.planes()]
    n, spent, best = 0, 0.0, 1e9
    while spent < seconds and n < 200:
        for p, q in zip(f.planes(), pristine):
            np.copyto(p, q)               # untimed: same content every frame
        t0 = time.perf_counter()
        run(f)
        dt = time.perf_counter() - t0
        spent += dt
        best = min(best, dt)
        n += 1
    out = {"value": round(w * h * n / spent / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": kind, "lib": timed_lib,
           "sample": f"{n} frames {w}x{h} {depth}-bit {'4:4:4' if sx == 1 and sy == 1 else '4:2:0'} {trace.rsplit('_', 2)[0]}, hot path only "
                     f"(line loop of vfgs_main.c:664-682), mean {spent / n * 1e3:.1f} ms/frame, best {best * 1e3:.1f} ms/frame",
           "host_cpus": os.cpu_count()}
    if note:
        out["note"] = note
    return out


# everything that shapes the grain kernels' code object AND the launches the host makes of them: a profile (rocprof kernel
# stats, PMC traffic) is only quoted for the sources it was taken with
PROFILED_SOURCES = ("vfgs_kernel.hip", "vfgs_layout.h", "vfgs_host.cpp")


def kernel_sha():
    import hashlib
    h = hashlib.sha256()
    for name in PROFILED_SOURCES:
        h.update(name.encode() + b"\0" + (ROOT / "versatilefilmgrain_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


# BASELINE.json configs[0..3] (configs[4] is the headline workload above): name, w, h, depth, (subx, suby), trace, variants.
# Every configuration is reported at 8 frames per launch (the headline's batch) and at a larger batch (1080p: 400 MB, 2160p: 800 MB) -- a
# launch pays ~5 us of fill, drain and kernel boundary, which is a quarter of a 100 MB launch.  A variant is
# (frames per launch, how the frames lie in memory, content): "pitch" = one allocation, frames at a constant pitch
# (vfgs_hip_add_grain_frames_dev); "list" = every frame three allocations of its own, in shuffled order with odd gaps, handed over as
# a list of plane pointers (vfgs_hip_add_grain_frame_list_dev); content "uniform" = random over the full code range (the
# worst case for LUT / pattern divergence), "ramp" = smooth diagonal ramp + -4..+4 noise (SURVEY 8d's secondary, natural-like content).
CONFIGS = [
    ("1920x1080 10-bit 4:2:0, cfg fgs_sei", 1920, 1080, 10, (2, 2), "fgs_sei_10_420", [(8, "pitch", "uniform"), (32, "pitch", "uniform"), (32, "list", "uniform")]),
    ("1920x1080 10-bit 4:2:0, cfg fgs_sei_ff_test1", 1920, 1080, 10, (2, 2), "fgs_sei_ff_test1_10_420", [(8, "pitch", "uniform"), (32, "pitch", "uniform")]),
    ("3840x2160 10-bit 4:2:0, cfg fgs_sei_ar_test1", 3840, 2160, 10, (2, 2), "fgs_sei_ar_test1_10_420", [(8, "pitch", "uniform"), (16, "pitch", "uniform")]),
    ("3840x2160 8-bit 4:4:4, cfg fgs_afgs1_test1", 3840, 2160, 8, (1, 1), "fgs_afgs1_test1_8_444", [(8, "pitch", "uniform"), (16, "pitch", "uniform")]),
    ("7680x4320 10-bit 4:2:0, cfg fgs_sei", W, H, DEPTH, (SUBX, SUBY), TRACE, [(8, "pitch", "ramp")]),
]


def fill_uniform(t, depth, g, chunk=1 << 26):
    """uniform random samples into a flat device tensor, a bounded temporary at a time"""
    import torch
    for o in range(0, t.numel(), chunk):
        n = min(chunk, t.numel() - o)
        t[o:o + n] = torch.randint(0, 1 << depth, (n,), dtype=torch.int32, device=t.device, generator=g).to(t.dtype)


def fill_ramp(t, depth, g, frame, rows, cols):
    """one plane of frame `frame`: smooth diagonal ramp inside the legal range + -4..+4 noise (tools/bench_config.py --content ramp)"""
    import torch
    r = torch.arange(rows, device=t.device, dtype=torch.int32).view(rows, 1)
    c = torch.arange(cols, device=t.device, dtype=torch.int32).view(1, cols)
    lo, hi = (16 << (depth - 8)), (235 << (depth - 8))
    base = lo + ((r * 3 + c * 2 + frame * 37) >> 3) % (hi - lo)
    noise = torch.randint(-4, 5, (rows, cols), dtype=torch.int32, device=t.device, generator=g)
    t.view(rows, cols).copy_((base + noise).clamp(0, (1 << depth) - 1).to(t.dtype))


def bench_configs(h, stream, steps_ms=60.0, cpu_seconds=2.5, out=None):
    """BASELINE.json configs[0..3] (+ the headline workload with natural-like content) on this GPU: device-resident frames, in place,
    plain stream, HIP events on the launching stream; one post-timing launch of the timed shape per entry is compared with the
    oracle, frame by frame; the reference's CPU path is timed beside every size (one core, a bounded sample)."""
    import random
    import numpy as np
    import torch
    import vfgs_testlib as T
    out = [] if out is None else out     # (the caller's list: entries that finished survive an exception in a later one)
    for name, w, hh, depth, (sx, sy), trace, variants in CONFIGS:
        rec = T.load_trace(trace)
        dt = torch.int16 if depth > 8 else torch.uint8
        npdt = np.uint16 if depth > 8 else np.uint8
        sz = 2 if depth > 8 else 1
        cw, ch = w // sx, hh // sy
        ny, nc = hh * w, ch * cw
        frame_bytes = sz * (ny + 2 * nc)
        cpu = cpu_baseline(cpu_seconds, trace, w, hh, depth, sx, sy) if cpu_seconds > 0 else None
        for batch, layout, content in variants:
            h.lib.vfgs_hip_reset_state()
            T.replay(h, rec)
            pool = max(3, int(1.0e9 // (frame_bytes * batch)) + 1)       # > 1 GB cycled through: nothing is served by the 256 MiB Infinity Cache
            g = torch.Generator(device="cuda").manual_seed(11)
            rnd = random.Random(5)
            keep = []                     # every allocation of this variant
            sets = []                     # per buffer set: the planes of its frames, [(Y, U, V)] as flat tensors
            for _ in range(pool):
                if layout == "pitch":
                    b = torch.empty(batch * (ny + 2 * nc), dtype=dt, device="cuda")
                    keep.append(b)
                    sets.append([(b[f * ny:(f + 1) * ny], b[batch * ny + f * nc:batch * ny + (f + 1) * nc],
                                  b[batch * (ny + nc) + f * nc:batch * (ny + nc) + (f + 1) * nc]) for f in range(batch)])
                else:
                    planes = [None] * (3 * batch)
                    order = list(range(3 * batch))
                    rnd.shuffle(order)
                    for k in order:
                        planes[k] = torch.empty(ny if k % 3 == 0 else nc, dtype=dt, device="cuda")
                        keep.append(torch.empty(rnd.randrange(1, 64) * 4096, dtype=torch.uint8, device="cuda"))      # a gap of odd size behind it
                    keep.append(planes)
                    sets.append([(planes[3 * f], planes[3 * f + 1], planes[3 * f + 2]) for f in range(batch)])

            def fill(frames, gen, kind):
                for f, (y, u, v) in enumerate(frames):
                    for t, rows, cols in ((y, hh, w), (u, ch, cw), (v, ch, cw)):
                        if kind == "ramp":
                            fill_ramp(t, depth, gen, f, rows, cols)
                        else:
                            fill_uniform(t, depth, gen)
            for frames in sets:
                fill(frames, g, content)
            lists = [h.frame_list([(y.data_ptr(), u.data_ptr(), v.data_ptr()) for y, u, v in frames]) for frames in sets] if layout == "list" else None

            def step(i):
                if lists:
                    h.add_grain_frame_list_dev(lists[i % pool], w, hh, w, cw, stream)
                else:
                    y, u, v = sets[i % pool][0]
                    h.add_grain_frames_dev(y.data_ptr(), u.data_ptr(), v.data_ptr(), w, hh, w, cw, batch, sz * ny, sz * nc, stream)
            t0, n = time.perf_counter(), 0
            while (time.perf_counter() - t0) * 1e3 < 40.0:          # untimed pre-roll
                for _ in range(8):
                    step(n)
                    n += 1
                torch.cuda.synchronize()
            per_launch_ms = (time.perf_counter() - t0) * 1e3 / n
            steps = max(20, min(400, int(steps_ms / per_launch_ms)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(steps):
                step(i)
            e1.record()
            torch.cuda.synchronize()
            launch_us = e0.elapsed_time(e1) / steps * 1e3
            info = h.last_launch_info()
            # parity of the timed shape: fresh frames of the same content, reset seed state, every frame against the oracle
            h.lib.vfgs_hip_reset_state()
            T.replay(h, rec)
            ora = T.OracleHW()
            T.replay(ora, rec)
            fill(sets[0], torch.Generator(device="cuda").manual_seed(4242), content)
            src = [tuple(t.cpu().numpy().view(npdt) for t in fr) for fr in sets[0]]
            step(0)
            torch.cuda.synchronize()
            bad = 0
            for f, (ys, us, vs) in enumerate(src):
                fr = T.Frame(w, hh, depth, sx, sy, stride=w, cstride=cw)
                fr.Y[:hh] = ys.reshape(hh, w)           # (1080 lines: the allocation is padded to 1088, yuv.c:54-87)
                fr.U[:ch] = us.reshape(ch, cw)
                fr.V[:ch] = vs.reshape(ch, cw)
                ora.add_grain_frame(fr)
                gy, gu, gv = (t.cpu().numpy().view(npdt) for t in sets[0][f])
                bad += not (np.array_equal(fr.Y[:hh].ravel(), gy) and np.array_equal(fr.U[:ch].ravel(), gu) and np.array_equal(fr.V[:ch].ravel(), gv))
            parity = bad == 0 and h.seed_state() == ora.seed_state()
            nbytes = 2 * frame_bytes * batch
            gbs = nbytes / (launch_us * 1e-6) / 1e9
            e = {"workload": name + f", seed 12345, {'uniform random samples' if content == 'uniform' else 'ramp + -4..+4 noise'}", "content": content,
                 "frames_per_launch": batch, "frame_layout": "one allocation, constant frame pitch" if layout == "pitch" else "every plane its own allocation (frame list)",
                 "steps": steps, "launch_us": round(launch_us, 2), "algorithmic_bytes_per_launch": nbytes, "achieved": round(gbs, 1),
                 "frac": round(gbs / HBM_PEAK_GBS, 4), "mpixels_per_s": round(batch * w * hh / launch_us, 1),
                 "kernel": info["kernel"] if info else None, "workgroups_per_frame": info["workgroups_per_frame"] if info else None,
                 "persistent_luma_workgroups": info["persistent_luma_workgroups"] if info else None,
                 "parity_checked": bool(parity)}
            if cpu:
                e["cpu_baseline"] = cpu
            out.append(e)
            del sets, keep, lists
            torch.cuda.empty_cache()
    return out


def gpu_numa_cpus(index, sysfs="/sys"):
    """CPUs of the NUMA node GPU `index` (HIP order) hangs off, from sysfs -- nothing here touches the GPU.  None if it cannot be told."""
    try:
        nodes = []
        base = Path(sysfs) / "class/kfd/kfd/topology/nodes"
        for d in sorted(base.iterdir(), key=lambda p: int(p.name)):
            try:      # (a container sees only the properties of the GPUs it was given: the others are not this process's devices)
                props = dict(l.split() for l in (d / "properties").read_text().splitlines() if len(l.split()) == 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                nodes.append(props)
        # ROCR_VISIBLE_DEVICES filters what the runtime sees; HIP_VISIBLE_DEVICES then renumbers THAT list, and
        # CUDA_VISIBLE_DEVICES is only HIP's alias for it (launchers often set both: applied once, HIP takes precedence)
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES" if os.environ.get("HIP_VISIBLE_DEVICES") else "CUDA_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis and all(x.strip().isdigit() for x in vis.split(",")):
                nodes = [nodes[int(x)] for x in vis.split(",") if int(x) < len(nodes)]
        p = nodes[index]
        loc, dom = int(p["location_id"]), int(p.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        cpus = set()
        for part in (Path(sysfs) / "bus/pci/devices" / bdf / "local_cpulist").read_text().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return bdf, cpus
    except Exception:
        return None


def pin_to_gpu(index):
    """Pin this process to the CPUs next to its GPU (before the first GPU call: the runtime's threads inherit the mask)."""
    got = gpu_numa_cpus(index)
    if not got:
        return "not pinned (GPU -> NUMA node not found in sysfs)"
    bdf, cpus = got
    cpus &= os.sched_getaffinity(0)
    if not cpus:
        return f"not pinned (no allowed CPU next to {bdf})"
    os.sched_setaffinity(0, cpus)
    return f"{len(cpus)} CPUs local to {bdf} ({min(cpus)}-{max(cpus)})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8, help="frames per launch per rank (a step = gpus*batch frames)")
    ap.add_argument("--pool", type=int, default=4, help="distinct step-sized buffer sets cycled through (total >> 256 MiB Infinity Cache)")
    # the first ~10 ms of launches of a process run 4-8 % slower (clocks / TLBs settling); an untimed, time-based
    # pre-roll in front of --warmup makes the line independent of how few --steps/--warmup the caller asks for
    ap.add_argument("--preroll-ms", type=float, default=150.0, help="untimed launches for at least this long before --warmup")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the same-process copy ceiling")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="developer option: run the N-rank code path with all ranks on cuda:0")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: a step is gpus*batch frames; strong: a step is batch frames split over the ranks")
    ap.add_argument("--no-region", action="store_true", help="skip the second leg (the same launches inside an overlap region)")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-timing parity launch")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE.json configs[0..3] leg (N=1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  A child process (never exec: this box forbids replacing a process image once
        # a GPU has been touched, and nothing here has touched one yet), its stdout relayed line by line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        sys.exit(child.wait())

    import torch
    import torch.distributed as dist
    import vfgs_testlib as T
    from versatilefilmgrain_amd import hw

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.rehearse_on_one_gpu:
        local = 0
    affinity = pin_to_gpu(local) if world > 1 else "not pinned (one rank)"      # before anything touches the GPU
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local)
    if world > 1 and not args.rehearse_on_one_gpu:
        # (KFD node order was assumed to be HIP's device order: say so if the device the runtime gave this rank sits elsewhere)
        try:
            got, real = gpu_numa_cpus(local), int(torch.cuda.get_device_properties(local).pci_bus_id)
            if got and real != int(got[0].split(":")[1], 16):
                affinity += f"; MISMATCH: sysfs says {got[0]}, the runtime's device {local} is on bus {real:02x}"
        except Exception:      # noqa: BLE001  (an attribute this torch build lacks: nothing to cross-check)
            pass
    if world > 1:
        # The data path has no collective (stripes are independent, DESIGN.md "multi-GPU"): the process group only
        # carries the barrier around the timed region and the max / gather of the timings -- host-side, over gloo.
        # (gloo announces its connections on fd 1, from C++, whenever it makes them: the contract is ONE line on stdout, so the
        # process's fd 1 is pointed at stderr for the rest of the run and the JSON line goes to the saved descriptor)
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        dist.init_process_group("gloo")

    else:
        json_out = sys.stdout
    h = hw.VfgsHip(device=local)
    T.replay(h, T.load_trace(TRACE))     # programs banks/LUTs/shift/depth/subsampling/seed 12345

    nbr = (H + 15) // 16
    row0, nrows = split_rows(nbr, world)[rank]
    part_y, part_h = row0 * 16, min(nrows * 16, H - row0 * 16)
    # rank r: its stripe of each of these frames (weak: per-GPU work constant in N; strong: the job's work constant in N)
    frames_per_launch = world * args.batch if args.scaling == "weak" else args.batch
    stride, cstride = W, W // SUBX
    ypitch = part_h * stride * 2                 # bytes between consecutive frames' stripes
    cpitch = (part_h // SUBY) * cstride * 2
    pool = max(2, min(args.pool, args.steps + args.warmup))
    g = torch.Generator(device="cuda").manual_seed(1 + rank)
    # one allocation per buffer set: [Y stripes of all frames | U | V], so that the copy-ceiling kernels can stream
    # exactly the bytes of a grain launch in ONE launch of their own
    ny, nc = frames_per_launch * part_h * stride, frames_per_launch * (part_h // SUBY) * cstride
    sets = [torch.randint(0, 1024, (ny + 2 * nc,), dtype=torch.int16, device="cuda", generator=g) for _ in range(pool)]
    set_bytes = (ny + 2 * nc) * 2
    ptrs = [(b.data_ptr(), b.data_ptr() + 2 * ny, b.data_ptr() + 2 * (ny + nc)) for b in sets]
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        y, u, v = ptrs[i % pool]
        h.add_grain_frames_part_dev(y, u, v, W, H, part_y, part_h, stride, cstride, frames_per_launch, ypitch, cpitch, stream)

    def barrier():
        if world > 1:
            dist.barrier()

    # untimed pre-roll (time based), then the contract's warmup
    t0 = time.perf_counter()
    n_pre = 0
    while (time.perf_counter() - t0) * 1e3 < args.preroll_ms:
        for _ in range(8):
            step(n_pre)
            n_pre += 1
        torch.cuda.synchronize()
    preroll_ms = (time.perf_counter() - t0) * 1e3
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    launch_ms = ev0.elapsed_time(ev1) / args.steps     # same stream as the kernels (torch's current stream)
    launch_info = h.last_launch_info()                 # what the timed launches dispatched (the library's own record)

    # ---- second leg (reported beside the contract's numbers, never as them): the same K launches inside an overlap region
    # (include/vfgs_hip.h: independent frames, the library alternates two internal streams, so one launch's tail overlaps the
    # next one's head).  Kernels overlap here, so a per-kernel duration (rocprof) no longer measures throughput; the figure
    # is bytes / wall time of the region on the device.
    region = None
    region_launches = max(args.steps, 100)     # (the region's first and last launch run alone: a short region is mostly edges)
    if not args.no_region:
        # (the first region creates the internal streams and their queues; and launches alternating between two queues take a
        # few dozen launches to reach their steady rate -- 8 warm-up launches: 0.74, 50: 0.76, profiles/r03_ab32_region_warmup.log)
        for _ in range(2):
            h.overlap_begin(stream)
            for i in range(32):
                step(i)
            h.overlap_end(stream)
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        h.overlap_begin(stream)
        for i in range(region_launches):
            step(args.warmup + i)
        h.overlap_end(stream)
        r1.record()
        torch.cuda.synchronize()
        region = r0.elapsed_time(r1) / region_launches
        barrier()

    # ---- same process, same buffers, same launch size: what a pure streaming kernel reaches on THIS device now ----
    ceilings = {}
    if not args.no_ceiling:
        def timed(fn, reps=24):
            for k in range(6):
                fn(k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(reps):
                fn(k)
            e1.record()
            torch.cuda.synchronize()
            return 2.0 * set_bytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9     # every byte read once + written once
        base = [b.data_ptr() for b in sets]
        cus = h.device_info()["cu_count"]
        # bench-only streaming kernels, a library of their own (tools/bench_diag.hip; not part of the product or its ABI)
        from versatilefilmgrain_amd import build as vbuild
        dlib = C.CDLL(str(vbuild.build_diag()))
        dlib.vfgs_bench_diag_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]

        def diag(src, dst, mode, grid):
            rc = dlib.vfgs_bench_diag_stream(src, dst, set_bytes, mode, grid, cus, stream)
            assert rc == 0, f"vfgs_bench_diag_stream: hipError {rc}"
        ceilings["inplace_rmw_4KiB_per_wave_8wg_per_cu"] = timed(lambda k: diag(0, base[k % pool], 1, 8 * cus))
        ceilings["inplace_rmw_4KiB_per_wave_12wg_per_cu"] = timed(lambda k: diag(0, base[k % pool], 1, 12 * cus))
        ceilings["inplace_rmw_4KiB_per_wave_one_wg_per_16KiB"] = timed(lambda k: diag(0, base[k % pool], 2, 0))
        ceilings["inplace_rmw_nontemporal_one_wg_per_16KiB"] = timed(lambda k: diag(0, base[k % pool], 3, 0))
        ceilings["out_of_place_uint4_copy"] = timed(lambda k: diag(base[k % pool], base[(k + 1) % pool], 0, 8 * cus))

    # ---- parity of what was timed: one more launch of the same shape, from pristine frames and a reset seed state, every
    # frame against the oracle (this rank's stripes; rows outside them are not this rank's business and stay zero) ----
    parity = None
    if not args.no_parity:
        import numpy as np
        rec = T.load_trace(TRACE)
        h.lib.vfgs_hip_reset_state()
        T.replay(h, rec)
        ora = T.OracleHW()
        T.replay(ora, rec)
        gp = torch.Generator(device="cuda").manual_seed(977 + rank)
        sets[0].copy_(torch.randint(0, 1024, (ny + 2 * nc,), dtype=torch.int16, device="cuda", generator=gp))
        src = sets[0].cpu().numpy().view(np.uint16)
        step(0)
        torch.cuda.synchronize()
        got = sets[0].cpu().numpy().view(np.uint16)
        crow0, crows = part_y // SUBY, part_h // SUBY
        bad = 0
        for f in range(frames_per_launch):
            fr = T.Frame(W, H, DEPTH, SUBX, SUBY, stride=stride, cstride=cstride)
            yo, uo, vo = f * part_h * stride, ny + f * crows * cstride, ny + nc + f * crows * cstride
            fr.Y[part_y:part_y + part_h] = src[yo:yo + part_h * stride].reshape(part_h, stride)
            fr.U[crow0:crow0 + crows] = src[uo:uo + crows * cstride].reshape(crows, cstride)
            fr.V[crow0:crow0 + crows] = src[vo:vo + crows * cstride].reshape(crows, cstride)
            ora.add_grain_frame(fr)
            ok = (np.array_equal(fr.Y[part_y:part_y + part_h].ravel(), got[yo:yo + part_h * stride])
                  and np.array_equal(fr.U[crow0:crow0 + crows].ravel(), got[uo:uo + crows * cstride])
                  and np.array_equal(fr.V[crow0:crow0 + crows].ravel(), got[vo:vo + crows * cstride]))
            bad += not ok
        parity = bad == 0 and h.seed_state() == ora.seed_state()
        if not parity:
            print(f"bench.py: rank {rank}: PARITY FAILURE: {bad} of {frames_per_launch} frames differ from the oracle", file=sys.stderr)

    n_seen, per_rank_us = 1, [round(launch_ms * 1e3, 2)]
    if world > 1:
        if parity is not None:
            pt_ = torch.tensor([1 if parity else 0], dtype=torch.int64)
            dist.all_reduce(pt_, op=dist.ReduceOp.MIN)
            parity = bool(pt_.item())
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        n_seen = int(one.item())
        lus = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lus, torch.tensor([launch_ms * 1e3], dtype=torch.float64))
        per_rank_us = [round(float(x.item()), 2) for x in lus]

    if rank == 0:
        frames_total = args.steps * frames_per_launch          # whole frames finished by all ranks together
        mpix = frames_total * W * H / elapsed / 1e6
        samples_per_launch = frames_per_launch * (part_h * W + 2 * (part_h // SUBY) * (W // SUBX))
        bytes_per_launch = 4 * samples_per_launch              # 2 B read + 2 B written per sample
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        # PMC-derived HBM bytes per launch: only valid for the kernel source they were collected with
        traffic, traffic_source = None, None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists() and world == 1:
            rec = json.loads(tf.read_text())
            if rec.get("batch") == args.batch and rec.get("kernel_sha16") == kernel_sha() and rec.get("sources") == list(PROFILED_SOURCES):
                traffic = rec.get("bytes_per_launch")
                traffic_source = {"file": "profiles/hbm_traffic.json", "stale": False, "date": rec.get("date"), "kernel_sha16": rec.get("kernel_sha16"),
                                  "sources": rec.get("sources"), "how": rec.get("correction")}
            else:
                traffic_source = {"file": "profiles/hbm_traffic.json", "stale": True, "kernel_sha16_now": kernel_sha(),
                                  "kernel_sha16_profiled": rec.get("kernel_sha16"), "sources": list(PROFILED_SOURCES)}
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": launch_info["kernel"] if launch_info else None,
                "kernel_launch": launch_info,       # vfgs_hip_last_launch_info(): form of the table image, workgroups, rows per wave ...
                "launch_us": round(launch_ms * 1e3, 2),
                "algorithmic_bytes_per_launch": bytes_per_launch}
        if region is not None:
            roof["overlap_region"] = {"launch_us": round(region * 1e3, 2), "achieved": round(bytes_per_launch / (region * 1e-3) / 1e9, 1),
                                      "frac": round(bytes_per_launch / (region * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "mpixels_per_s_this_rank": round(frames_per_launch * W * part_h / (region * 1e-3) / 1e6, 1),
                                      "launches": region_launches,
                                      "note": "same launches between vfgs_hip_overlap_begin/_end (two run at a time); wall time of the region / launches, rank 0"}
        if ceilings:
            best = max(ceilings.values())
            roof["copy_ceiling_gbs"] = round(best, 1)
            roof["frac_of_ceiling"] = round(achieved / best, 4)
            roof["copy_ceilings_gbs"] = {k: round(v, 1) for k, v in ceilings.items()}
            plain = max(v for k, v in ceilings.items() if "nontemporal" not in k)
            roof["frac_of_plain_ceiling"] = round(achieved / plain, 4)
            roof["copy_ceiling_note"] = ("pure streaming kernels (tools/bench_diag.hip), same process, same buffers, same bytes per launch, measured right after "
                                         "the timed region; copy_ceiling_gbs is the best of them (the nontemporal stream needs line-aligned accesses by waves that "
                                         "live for one 4 KiB item; the grain kernel has the aligned accesses, its waves stream whole rows: DESIGN.md 5), frac_of_plain_ceiling is against the best cached one")
        out = {
            "metric": "Mpixels/s (Y+UV) + achieved HBM GB/s vs roofline, 4320p 10-bit 4:2:0",
            "value": round(mpix, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "7680x4320 10-bit 4:2:0, cfg fgs_sei (8 luma patterns), seed 12345, uniform random samples",
                       "frames_per_step": frames_per_launch, "stripe_split": f"{world} x block-row stripes",
                       "pool_frames": pool * frames_per_launch, "msamples_per_s": round(mpix * 1.5, 1),
                       "preroll_ms": round(preroll_ms, 1), "preroll_launches": n_pre,
                       "sync_backend": "gloo (barrier + max only; no collective on the data path)" if world > 1 else "none",
                       "n_ranks_seen": n_seen, "launch_us_per_rank": per_rank_us, "cpu_affinity_rank0": affinity},
            "roofline": roof,
            "parity_checked": parity,
        }
        if world == 1 and not args.no_configs:
            # A failure of this leg for want of a RESOURCE (device or host memory, a fixture that did not travel) must not cost the
            # headline its line: it is recorded, and the entries that had finished stay.  Anything else -- an error from the library
            # or from HIP in the middle of an entry -- is a failed run: recorded too, and the line loses its parity claim.
            import torch
            out["configs"] = []
            try:
                bench_configs(h, stream, cpu_seconds=0.0 if args.no_cpu else 2.5, out=out["configs"])
            except (torch.cuda.OutOfMemoryError, MemoryError, FileNotFoundError) as ex:
                out["configs_error"] = f"{type(ex).__name__}: {ex}"
            except Exception as ex:      # noqa: BLE001
                out["configs_error"] = f"{type(ex).__name__}: {ex}"
                parity = False
            if not all(c["parity_checked"] for c in out["configs"]):
                parity = False
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(8.0)
        if parity is False:
            # a kernel that produced wrong bytes gets no benchmark line: the numbers are withheld and the exit code says so
            bad_cfg = [c["workload"] + f" x{c['frames_per_launch']} ({c['frame_layout']})" for c in out.get("configs", []) if not c["parity_checked"]]
            out = {"metric": out["metric"], "error": "parity failure: output differs from the oracle" if "configs_error" not in out or bad_cfg
                   else "the configs leg failed: " + out["configs_error"], "parity_checked": False,
                   "n_gpus": world, "failed_configs": bad_cfg, **({"configs_error": out["configs_error"]} if "configs_error" in out else {})}
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
