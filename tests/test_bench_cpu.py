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


def test_rank_pinning_reads_the_gpus_numa_cpus_from_sysfs(tmp_path, monkeypatch):
    """bench.py pins every rank to the CPUs next to its GPU before the first GPU call: KFD topology node -> PCI address ->
    local_cpulist.  A container sees only the properties of the GPUs it was given (the others: permission denied)."""
    import bench
    nodes = tmp_path / "class/kfd/kfd/topology/nodes"
    for i, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0x0d00), (1024, 0x2600)]):      # two CPU nodes, two GPUs
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text(f"cpu_cores_count 64\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    (nodes / "4").mkdir()                               # a GPU of another container: unreadable
    (nodes / "4" / "properties").mkdir()
    for bdf, cpus in (("0000:0d:00.0", "0-63,128-191"), ("0000:26:00.0", "64-127,192-255")):
        d = tmp_path / "bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    bdf, cpus = bench.gpu_numa_cpus(1, sysfs=str(tmp_path))
    assert bdf == "0000:26:00.0" and cpus == set(range(64, 128)) | set(range(192, 256))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")      # the process sees only the second GPU, as its device 0
    assert bench.gpu_numa_cpus(0, sysfs=str(tmp_path))[0] == "0000:26:00.0"
    assert bench.gpu_numa_cpus(3, sysfs=str(tmp_path)) is None and bench.gpu_numa_cpus(0, sysfs=str(tmp_path / "nowhere")) is None
    # CUDA_VISIBLE_DEVICES is HIP's alias for HIP_VISIBLE_DEVICES and launchers set both: one filter, not two in a row
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "1")
    assert bench.gpu_numa_cpus(0, sysfs=str(tmp_path))[0] == "0000:26:00.0"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bench.gpu_numa_cpus(0, sysfs=str(tmp_path))[0] == "0000:26:00.0"      # ... and alone it still counts
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,0")                              # the runtime's own filter comes first
    assert bench.gpu_numa_cpus(0, sysfs=str(tmp_path))[0] == "0000:0d:00.0"       # CUDA's "1" of ROCR's (second, first) = the first GPU
