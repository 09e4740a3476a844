"""CPU, world_size 2 and 8 (gloo): the multi-GPU decomposition and its torch.distributed plumbing.

The product kernels need a GPU; what can be checked here is what the N > 1 path rests on:
 * bench.split_rows partitions the block rows exactly (contiguous, sizes differ by <= 1);
 * stripes are INDEPENDENT given the seed registers: each rank fast-forwards the seed state
   machine to its stripe (no data dependence, no halo rows), processes only its lines, and the
   gathered stripes equal the whole frame -- done with the oracle's line API;
 * the barrier / max-reduction bench.py uses work under gloo.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vfgs_testlib as T

sys.path.insert(0, str(T.ROOT))


def test_split_rows_partitions_exactly():
    from bench import split_rows
    for nbr in (68, 135, 270, 9, 8, 1):
        for n in (1, 2, 3, 4, 8):
            parts = split_rows(nbr, n)
            assert len(parts) == n
            assert parts[0][0] == 0 and sum(k for _, k in parts) == nbr
            for (r0, k0), (r1, _) in zip(parts, parts[1:]):
                assert r1 == r0 + k0
            sizes = [k for _, k in parts]
            assert max(sizes) - min(sizes) <= 1
    assert [k for _, k in split_rows(270, 8)] == [34, 34, 34, 34, 34, 34, 33, 33]


def _rank(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench import split_rows
    w, h, nfr = 320, 208, 2
    rec = T.load_trace("fgs_sei_10_420")
    ora = T.OracleHW()
    T.replay(ora, rec)
    frames, _ = T.lcg_frames(w, h, 10, 2, 2, nfr)
    nbr = (h + 15) // 16
    row0, nrows = split_rows(nbr, world)[rank]
    y0, y1 = row0 * 16, min((row0 + nrows) * 16, h)
    dummy = T.Frame(w, 16, 10, 2, 2)
    outs = []
    for f in frames:
        # the seed state machine runs over every line (vfgs_hw.c:291-298); samples are touched only in the stripe
        for y in range(h):
            if y0 <= y < y1:
                ora.add_grain_line(f.Y[y].ctypes.data, f.U[y // 2].ctypes.data, f.V[y // 2].ctypes.data, y, w)
            else:
                ora.add_grain_line(dummy.Y[0].ctypes.data, dummy.U[0].ctypes.data, dummy.V[0].ctypes.data, y, w)
        outs.append(f)
    dist.barrier()
    # gather the stripes on every rank: rows outside a rank's stripe are still the input
    for f in outs:
        for plane, div in ((f.Y, 1), (f.U, 2), (f.V, 2)):
            t = torch.from_numpy(plane.astype(np.int32))
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            for r, (rr0, kk) in enumerate(split_rows(nbr, world)):
                a, b = rr0 * 16 // div, min((rr0 + kk) * 16, h) // div
                plane[a:b] = parts[r][a:b].numpy().astype(plane.dtype)
    # the timing reduction of bench.py
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        whole = T.OracleHW()
        T.replay(whole, rec)
        want, _ = T.lcg_frames(w, h, 10, 2, 2, nfr)
        for x in want:
            whole.add_grain_frame(x)
        ok = all(a.equal_all(b) for a, b in zip(outs, want)) and whole.seed_state() == ora.seed_state()
        ret.put((ok, float(tmax.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_stripes_assemble_to_whole_frame(world):
    """world 8 = the node the scaling bench runs on (13 block rows -> stripes of 2,2,2,2,2,1,1,1 rows)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world) % 2000
    procs = [ctx.Process(target=_rank, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    ok, tmax = ret.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok
    assert tmax == float(world)
