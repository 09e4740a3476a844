"""CPU: the evidence chain of bench.py -- the profile a bench line quotes (profiles/hbm_traffic.json) is tied to a hash over
everything that shapes the grain kernels' code object and the launches the host makes of them."""
import json
import shutil
import sys
import warnings

import vfgs_testlib as T

sys.path.insert(0, str(T.ROOT))


def test_profile_hash_covers_kernel_layout_and_host_sources(tmp_path, monkeypatch):
    import bench
    assert set(bench.PROFILED_SOURCES) == {"vfgs_kernel.hip", "vfgs_layout.h", "vfgs_host.cpp"}
    csrc = tmp_path / "versatilefilmgrain_amd" / "csrc"
    csrc.mkdir(parents=True)
    for name in bench.PROFILED_SOURCES:
        shutil.copyfile(T.ROOT / "versatilefilmgrain_amd" / "csrc" / name, csrc / name)
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    base = bench.kernel_sha()
    monkeypatch.setattr(bench, "ROOT", T.ROOT)
    assert base == bench.kernel_sha()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    for name in bench.PROFILED_SOURCES:           # an edit of ANY of them makes every recorded profile stale
        p = csrc / name
        orig = p.read_bytes()
        p.write_bytes(orig + b"\n// edit\n")
        assert bench.kernel_sha() != base, name
        p.write_bytes(orig)
        assert bench.kernel_sha() == base


def test_recorded_traffic_names_its_sources():
    import bench
    rec = json.loads((T.ROOT / "profiles" / "hbm_traffic.json").read_text())
    assert rec["sources"] == list(bench.PROFILED_SOURCES) and len(rec["kernel_sha16"]) == 16 and rec["batch"] == 8
    assert 0.99 < rec["bytes_per_launch"] / 1592524800 < 1.05       # PMC traffic within 5 % of the algorithmic bytes
    if rec["kernel_sha16"] != bench.kernel_sha():
        warnings.warn("profiles/hbm_traffic.json was recorded for other sources than the tree's: bench.py will say \"stale\": true "
                      "(re-run tools/gpu_bench_profile.sh after the last kernel-affecting change)")
