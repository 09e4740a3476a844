"""Developer helper: writes tools/dev/vfgs_kernel_timeline.hip.txt = vfgs_kernel.hip + s_memrealtime marks in a sample of the
luma waves (sums in a device array, read by vfgs_hip_debug_timeline; tools/dev/timeline.sh builds and runs it)."""
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
s = (ROOT / "versatilefilmgrain_amd/csrc/vfgs_kernel.hip").read_text()


def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)


rep("namespace vfgs {\n", "namespace vfgs {\n__device__ unsigned long long g_tl[16];\n__device__ __forceinline__ unsigned long long tl_now() { return __builtin_amdgcn_s_memrealtime(); }\n")
rep("	const int pt = comp ? 1 : 0;\n", "	const int pt = comp ? 1 : 0;\n	const unsigned long long tl0 = tl_now();\n	const bool tl_on = lane == 0 && comp == 0 && (blockIdx.x % 61) == 0;     // a sample of the luma waves: the marks must not disturb\n	auto tl_mark = [&](int i) { if (tl_on) atomicAdd(&g_tl[i], tl_now() - tl0); };\n")
rep("	constexpr int STEP = kWavesPerWG * 64 * 16;\n", "	tl_mark(7);\n	constexpr int STEP = kWavesPerWG * 64 * 16;\n")
rep("#pragma unroll\n	for (int it = 0; it < NIT; it++)\n		*(u32x4*)(lds + min(", "	tl_mark(8);\n#pragma unroll\n	for (int it = 0; it < NIT; it++)\n		*(u32x4*)(lds + min(")
rep("	param_table(0, wc0, wu0);\n	__syncthreads();\n", "	param_table(0, wc0, wu0);\n	tl_mark(1);\n	__syncthreads();\n	tl_mark(2);\n	if (tl_on) atomicAdd(&g_tl[0], 1ull);\n")
rep("			else walk_row(std::false_type(), k, g_lo, g_hi);\n", "			else walk_row(std::false_type(), k, g_lo, g_hi);\n			if (k == k0 && h == 0) tl_mark(4);\n")
rep("	// the last position of my last row\n", "	tl_mark(5);\n	// the last position of my last row\n")
rep("	store_unit<DW, STA>(pdst, laned + (NU - 1) * UBD, outp);\n}\n", "	store_unit<DW, STA>(pdst, laned + (NU - 1) * UBD, outp);\n	__builtin_amdgcn_s_waitcnt(0);\n	tl_mark(6);\n}\n")
rep("ImageLayout layout_of(", "extern \"C\" int vfgs_hip_debug_timeline(unsigned long long* out, int reset)\n{\n	if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tl), z, sizeof z); }\n	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tl), 16 * sizeof(unsigned long long));\n}\n\nImageLayout layout_of(")
(ROOT / "tools/dev/vfgs_kernel_timeline.hip.txt").write_text(s)
print("written")
