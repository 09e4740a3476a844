#!/bin/bash
# Developer helper for gpurun: shader / fabric / memory clocks and package power while bench.py's grain launches run, against the
# pure nontemporal stream of the same bytes (is the kernel running into a power or clock limit that the stream does not see?)
cd $GRAFT_REPO_ROOT
smi() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Socket Power|Graphics Package" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.1; done; }
echo "== idle"; smi 3
echo "== grain kernel (bench.py, 6000 steps)"
python3 bench.py --no-cpu --no-ceiling --no-parity --steps 40000 --warmup 30 > /tmp/b.json 2>/dev/null &
sleep 6; smi 12; wait
python3 -c "import json; d=json.load(open('/tmp/b.json')); print('frac', d['roofline']['frac'])"
echo "== copy ceilings only (nontemporal stream last)"
python3 - <<'PY' &
import sys, ctypes as C, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from versatilefilmgrain_amd import build as vbuild
dlib = C.CDLL(str(vbuild.build_diag()))
dlib.vfgs_bench_diag_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
n = 1592524800 // 2
bufs = [torch.zeros(n // 2, dtype=torch.int16, device='cuda') for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream
t0 = time.time(); k = 0
while time.time() - t0 < 6:
    for _ in range(50):
        dlib.vfgs_bench_diag_stream(0, bufs[k % 3].data_ptr(), n, 3, 0, 256, st); k += 1
    torch.cuda.synchronize()
print('nt stream launches', k, 'GB/s', 2 * n * k / (time.time() - t0) / 1e9)
PY
sleep 3; smi 12; wait
