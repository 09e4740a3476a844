// Developer probe (round 5): WHY does the row walk (a wave streams K positions of 1 KiB through a ring of four register sets: wait for
// position u, store position u - 1, refill with position u + 4) top out at 0.73-0.75 of the HBM peak where waves that move one 4 KiB item
// and end reach 0.82-0.83?  Same bytes (in place, nontemporal, every byte read once and written once), no arithmetic, no LDS:
//   item     one 4 KiB item per wave: 4 loads, 4 stores, end                                  (bench_diag's ceiling kernel)
//   seq K    K items per wave, one after the other: 4 loads, 4 stores, 4 loads, 4 stores ...   (longer-lived waves, nothing overlapped inside a wave)
//   ring K   K x 4 positions per wave through the ring of four: the grain kernel's order        (store u - 1, load u + 4 interleaved)
//   dbl K    K items per wave, the NEXT item's 4 loads issued before this item's 4 stores       (two register sets of four)
//   ringw K  as ring, but the store of a position waits until its refill has been issued first (load u + 4, then store u - 1)
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/walk_probe tools/walk_probe.hip ; tools/bin/walk_probe [MiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const uint8_t* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
__device__ __forceinline__ u32x4 ld(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2); }
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, uint32_t off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2); }

// MODE 0 item/seq, 1 ring, 2 dbl, 3 ringw; a wave owns K consecutive items (K x 4 KiB); waves of a workgroup own consecutive wave regions
template <int MODE, int K>
__global__ __launch_bounds__(256) void walk(uint8_t* buf, size_t bytes)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t base = wave * (size_t)K * 4096;
	if (base >= bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, (uint32_t)std::min<size_t>(bytes - base, (size_t)K * 4096));
	if constexpr (MODE == 0)
	{
#pragma unroll 1
		for (int k = 0; k < K; k++)
		{
			u32x4 v[4];
#pragma unroll
			for (int u = 0; u < 4; u++) v[u] = ld(r, k * 4096 + u * 1024 + lane16);
#pragma unroll
			for (int u = 0; u < 4; u++) st(r, k * 4096 + u * 1024 + lane16, v[u] + 1u);
		}
	}
	else if constexpr (MODE == 1 || MODE == 3)
	{
		u32x4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
		for (int k = 0; k < K; k++)
		{
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const u32x4 t = v[u] + 1u;                                    // (waits for position 4k + u)
				const uint32_t here = k * 4096 + u * 1024 + lane16;
				if (MODE == 1) st(r, here, t);
				v[u] = ld(r, k + 1 < K ? here + 4096 : 0x80000000u);          // refill: four positions ahead (none behind the end: out of range)
				if (MODE == 3) st(r, here, t);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	else
	{
		u32x4 a[4], b[4];
#pragma unroll
		for (int u = 0; u < 4; u++) a[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
		for (int k = 0; k < K; k += 2)
		{
#pragma unroll
			for (int u = 0; u < 4; u++) b[u] = ld(r, k + 1 < K ? (k + 1) * 4096 + u * 1024 + lane16 : 0x80000000u);
#pragma unroll
			for (int u = 0; u < 4; u++) st(r, k * 4096 + u * 1024 + lane16, a[u] + 1u);
			__builtin_amdgcn_sched_barrier(0);
			if (k + 1 < K)
			{
#pragma unroll
				for (int u = 0; u < 4; u++) a[u] = ld(r, k + 2 < K ? (k + 2) * 4096 + u * 1024 + lane16 : 0x80000000u);
#pragma unroll
				for (int u = 0; u < 4; u++) st(r, (k + 1) * 4096 + u * 1024 + lane16, b[u] + 1u);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
}

template <int MODE, int K>
static double run(uint8_t* buf, size_t bytes, int reps)
{
	const size_t waves = (bytes + (size_t)K * 4096 - 1) / ((size_t)K * 4096);
	const unsigned grid = (unsigned)((waves + 3) / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((walk<MODE, K>), dim3(grid), dim3(256), 0, 0, buf, bytes);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((walk<MODE, K>), dim3(grid), dim3(256), 0, 0, buf, bytes);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

int main(int argc, char** argv)
{
	const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1536) << 20;
	uint8_t* buf;
	if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
	hipMemset(buf, 1, bytes);
	hipDeviceSynchronize();
	const int reps = 15;
	printf("buffer %zu MiB, in place, nontemporal 16-byte accesses, GB/s (read + write) and fraction of 8 TB/s, median of %d launches\n", bytes >> 20, reps);
#define LINE(name, MODE, K) { const double g = run<MODE, K>(buf, bytes, reps); printf("%-10s K=%-3d %8.1f  %.4f\n", name, K, g, g / 8000.0); fflush(stdout); }
	for (int round = 0; round < 2; round++)
	{
		LINE("item", 0, 1)
		LINE("seq", 0, 2) LINE("seq", 0, 4) LINE("seq", 0, 16)
		LINE("ring", 1, 2) LINE("ring", 1, 4) LINE("ring", 1, 16)
		LINE("ringw", 3, 4) LINE("ringw", 3, 16)
		LINE("dbl", 2, 2) LINE("dbl", 2, 4) LINE("dbl", 2, 16)
	}
	return 0;
}
