"""CPU: what the gfx950 code objects inside libvfgs_hip.so say about the kernels' resources.

The residency of the grain kernels is a measured choice: no kernel spills a vector register or uses scratch, the
kernels with a general-form plane hold a CU at four workgroups by their 40 KB image, the 10-bit all-one-pattern kernels are held at
four by LDS they allocate and do not use (six would fit; vfgs_layout.h lds_allocation), the 8-bit all-one-pattern kernels fit six per CU
(79 registers, 22.8 KB).  A compiler or source change that moves one of these moves the measured numbers with it: this test reads the
AMDGPU metadata notes of the product library (no GPU needed) and says which.
"""
import re
import shutil
import struct
import subprocess
from pathlib import Path

import pytest

import versatilefilmgrain_amd.build as B

READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
pytestmark = pytest.mark.skipif(not Path(READELF).exists(), reason="needs llvm-readelf")

LDS_PER_CU = 163840
VGPRS_PER_SIMD = 512


def code_objects(path):
    """The gfx950 ELF images of every clang offload bundle in a linked library."""
    d = Path(path).read_bytes()
    out = []
    for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), d):
        b = m.start()
        (n,) = struct.unpack_from("<Q", d, b + 24)
        p = b + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(d[b + off:b + off + size])
    return out


def kernel_records(elf, tmp):
    f = tmp / "co.elf"
    f.write_bytes(elf)
    text = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
    recs = []
    for blk in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else None
        recs.append({"name": g("name"), "lds": int(g("group_segment_fixed_size")), "vgpr": int(g("vgpr_count")),
                     "vgpr_spill": int(g("vgpr_spill_count")), "scratch": int(g("private_segment_fixed_size"))})
    return recs


@pytest.fixture(scope="module")
def grain_kernels(tmp_path_factory):
    B.build()
    tmp = tmp_path_factory.mktemp("co")
    ks = {}
    for co in code_objects(B.LIB):
        for r in kernel_records(co, tmp):
            m = re.match(r"_ZN4vfgs15grain_rw_kernelILi(\d+)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEE", r["name"])
            if m:
                depth, csubx, csuby, out8, oney, onec, wide, persist = map(int, m.groups())
                ks[(depth, csubx, csuby, out8, oney, onec, wide, persist)] = r
    return ks


def test_every_instantiation_is_in_the_library(grain_kernels):
    # 64 at 10 bit (with the narrowed destination), 24 at 8 bit: the dispatcher's table (vfgs_kernel.hip launch_depth)
    assert len([k for k in grain_kernels if k[0] == 10]) == 64 and len([k for k in grain_kernels if k[0] == 8]) == 24


def test_no_vector_register_spills_and_no_scratch(grain_kernels):
    bad = {k: r for k, r in grain_kernels.items() if r["vgpr_spill"] or r["scratch"]}
    assert not bad, bad


def workgroups_per_cu(r):
    by_lds = LDS_PER_CU // r["lds"]
    granule = (r["vgpr"] + 7) // 8 * 8
    by_vgpr = min(8, VGPRS_PER_SIMD // granule)      # waves per SIMD = workgroups per CU (a workgroup is four waves, one per SIMD)
    return min(by_lds, by_vgpr)


def test_resident_workgroups_per_cu_are_the_measured_choice(grain_kernels):
    for (depth, csubx, csuby, out8, oney, onec, wide, persist), r in grain_kernels.items():
        key = (depth, csubx, csuby, out8, oney, onec, wide, persist)
        n = workgroups_per_cu(r)
        if depth == 10 and oney:
            # held at four by unused LDS (+2 .. 4 % against the six that fit: profiles/r06_ab13_workgroups_per_cu.log; over general-form
            # chroma five would fit: r06_ab23), or by the 40 KB image of general-form chroma at 4:4:4
            assert r["lds"] == (40960 if onec or csubx * csuby > 1 else 39968) and n == 4, (key, r)
        elif depth == 8 and oney and onec and not wide:
            # the packed 16-bit form with a ring of two register sets: six (profiles/r06_ab5_ring_depth_six_waves.log)
            assert n == 6, (key, r)
        elif not (oney and onec) and not (oney and csubx * csuby > 1):
            # a general-form image of luma geometry (luma, or 4:4:4 chroma): 40 KB, four
            assert r["lds"] <= 40960 and n == 4, (key, r)
        else:
            assert n >= 4, (key, r)
