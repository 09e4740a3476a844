// Bench-only library (tools/bin/libvfgs_bench_diag.so, built by versatilefilmgrain_amd/build.py next to the product library but
// NOT part of it, its header or its C ABI): pure streaming kernels that move the SAME bytes as a grain
// launch (every byte read once and written once) with no arithmetic.  bench.py runs them in its own
// process, on its own buffers and at its own launch size, and reports the best of them as the copy
// ceiling of the chip next to the grain kernel's rate ("fraction of peak AND fraction of copy
// ceiling"; the roofline peak is the 8 TB/s datasheet figure, which no copy kernel reaches).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfgs {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// out of place, 16 bytes per lane, grid-stride
__global__ __launch_bounds__(256) void diag_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i] + 1u;
}

// in place, 4 KiB contiguous per wave and step (4 x 16 bytes per lane in flight), persistent waves
__global__ __launch_bounds__(256) void diag_rmw(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	size_t base = wave * 256;
	for (; base + 256 <= n; base += nwaves * 256)
	{
		u32x4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = buf[base + u * 64 + lane];
#pragma unroll
		for (int u = 0; u < 4; u++) buf[base + u * 64 + lane] = v[u] + 1u;
	}
	if (base < n)       // the last partial chunk
		for (size_t i = base + lane; i < n; i += 64) buf[i] = buf[i] + 1u;
}

// the same, one 4 KiB chunk per wave, as many workgroups as chunks (the hardware dispatcher deals them out)
__global__ __launch_bounds__(256) void diag_rmw_np(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t base = wave * 256;
	if (base + 256 <= n)
	{
		u32x4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = buf[base + u * 64 + lane];
#pragma unroll
		for (int u = 0; u < 4; u++) buf[base + u * 64 + lane] = v[u] + 1u;
	}
	else
		for (size_t i = base + lane; i < n; i += 64) buf[i] = buf[i] + 1u;
}

// diag_rmw_np with nontemporal loads and stores (aux bit 1 of the buffer instructions): lines are streamed past the L2
// instead of being allocated in it.  The fastest in-place stream found on gfx950 -- but only for line-aligned 1 KiB wave
// accesses and short-lived waves; the grain kernel's accesses are shifted by half a block and do not gain.
__global__ __launch_bounds__(256) void diag_rmw_np_nt(uint8_t* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const int lane = threadIdx.x & 63;
	const size_t base = wave * 256;
	if (base >= n) return;
	const size_t left = (n - base) * 16;
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + base * 16), 0, (uint32_t)(left < 4096 ? left : 4096), 0x00020000);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * 64 + lane) * 16, 0, 2);   // out of range: zeros
#pragma unroll
	for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, (u * 64 + lane) * 16, 0, 2);  // out of range: dropped
}

}  // namespace

// mode 0: out-of-place copy src -> dst;  1: in-place read-modify-write of dst, persistent, `grid` workgroups of 4 waves
// (0 = 8 per CU);  2: in-place, one workgroup per 16 KiB;  3: as 2 with nontemporal loads and stores.  bytes % 16 == 0.
static hipError_t launch_diag_stream(const void* src, void* dst, size_t bytes, int mode, int grid, int cu_count, hipStream_t stream)
{
	const size_t n = bytes / 16;
	if (n == 0) return hipSuccess;
	if (grid <= 0) grid = cu_count * 8;
	if (mode == 0) hipLaunchKernelGGL(diag_copy, dim3(grid), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, n);
	else if (mode == 1) hipLaunchKernelGGL(diag_rmw, dim3(grid), dim3(256), 0, stream, (u32x4*)dst, n);
	else if (mode == 2) hipLaunchKernelGGL(diag_rmw_np, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (u32x4*)dst, n);
	else if (mode == 3) hipLaunchKernelGGL(diag_rmw_np_nt, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, (uint8_t*)dst, n);
	else return hipErrorInvalidValue;
	return hipGetLastError();
}

}  // namespace vfgs

// C entry for bench.py (ctypes): 0 or a hipError_t.  Pointers and size: multiples of 16 bytes.
extern "C" int vfgs_bench_diag_stream(const void* src, void* dst, uint64_t bytes, int mode, int grid, int cu_count, void* stream)
{
	if (((uintptr_t)src | (uintptr_t)dst | bytes) & 15) return (int)hipErrorInvalidValue;
	if (mode < 0 || mode > 3 || !dst || (mode == 0 && !src)) return (int)hipErrorInvalidValue;
	return (int)vfgs::launch_diag_stream(src, dst, (size_t)bytes, mode, grid, cu_count, (hipStream_t)stream);
}
