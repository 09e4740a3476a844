#!/usr/bin/env python3
"""Developer helper (build container): copy what tools/gpu_bench_profile.sh and tools/gpu_profiles.sh left in gpurun_out/ into profiles/
(ROUND prefix, default r06) and print the numbers DESIGN.md quotes.  usage: python3 tools/dev/collect_evidence.py [r06]"""
import glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
s = json.load(open("gpurun_out/profile_summary.json"))
print("rocprof", s["kernel_stats"]["Calls"], "calls, mean", s["kernel_stats"]["AverageNs"], "ns; source hash", s["hbm_traffic"]["kernel_sha16"], "traffic", s["hbm_traffic"]["bytes_per_launch"])
json.dump(s["hbm_traffic"], open("profiles/hbm_traffic.json", "w"), indent=1)
shutil.copy("gpurun_out/profile_summary.json", f"profiles/{R}_profile_summary.json")
for a, b in (("bench_default", "bench_default"), ("bench_driver_args", "bench_driver_args"), ("bench_b1", "bench_one_frame_per_launch")):
    shutil.copy(f"gpurun_out/{a}.json", f"profiles/{R}_{b}.json")
shutil.copy(max(glob.glob("gpurun_out/prof_kt/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime), f"profiles/{R}_kernel_stats.csv")
shutil.copy("gpurun_out/pytest_gpu.log", f"profiles/{R}_pytest_gpu_final.log")
for f in glob.glob(f"gpurun_out/{R}_config*_kernel_stats.csv") + [f"gpurun_out/{R}_config_kernel_stats.json", f"gpurun_out/{R}_config_lines.jsonl", f"gpurun_out/{R}_host_pipeline.jsonl"]:
    shutil.copy(f, "profiles/")
for n in ("bench_driver_args", "bench_default", "bench_b1"):
    d = json.loads(open(f"gpurun_out/{n}.json").read().strip().split("\n")[-1])
    r = d["roofline"]
    print(n, "ms/step", d["ms_per_step"], "frac", r["frac"], "region", r.get("overlap_region", {}).get("frac"), "GB/s", r["achieved"], "traffic", r["traffic"])
    for c in d.get("configs", []):
        print("    %-50s x%-3d %-14s %.4f" % (c["workload"][:50], c["frames_per_launch"], c["frame_layout"][:14], c["frac"]))
    print("    cpu", d.get("cpu_baseline", {}).get("value"))
