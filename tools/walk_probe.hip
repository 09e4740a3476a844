// Developer probe (round 5): WHY does the row walk (a wave streams K positions of 1 KiB through a ring of four register sets: wait for
// position u, store position u - 1, refill with position u + 4) top out at 0.73-0.75 of the HBM peak where waves that move one 4 KiB item
// and end reach 0.82-0.83?  Same bytes (in place, nontemporal, every byte read once and written once), no arithmetic, no LDS:
//   item     one 4 KiB item per wave: 4 loads, 4 stores, end                                  (bench_diag's ceiling kernel)
//   seq K    K items per wave, one after the other: 4 loads, 4 stores, 4 loads, 4 stores ...   (longer-lived waves, nothing overlapped inside a wave)
//   ring K   K x 4 positions per wave through the ring of four: the grain kernel's order        (store u - 1, load u + 4 interleaved)
//   dbl K    K items per wave, the NEXT item's 4 loads issued before this item's 4 stores       (two register sets of four)
//   ringw K  as ring, but the store of a position waits until its refill has been issued first (load u + 4, then store u - 1)
//   ring8 K  the ring with EIGHT register sets (refill eight positions ahead)
//   split K  even waves only LOAD their K items (and fold them into one register), odd waves only STORE theirs: the same bytes each
//            way, but no load of any wave ever waits behind a store of the same wave (one in-order vmcnt per wave on gfx9)
//   stride G  persistent waves (G workgroups per CU), wave w moves items w, w + W, w + 2 W ... (W = all waves): long-lived waves whose
//            accesses of one iteration form ONE contiguous window of W x 4 KiB, like the item kernel's
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/walk_probe tools/walk_probe.hip ; tools/bin/walk_probe [MiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const uint8_t* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
__device__ __forceinline__ u32x4 ld(__amdgpu_buffer_rsrc_t r, uint32_t off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2); }
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, uint32_t off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2); }

// MODE 0 item/seq, 1 ring, 2 dbl, 3 ringw; a wave owns K consecutive items (K x 4 KiB); waves of a workgroup own consecutive wave regions
template <int MODE, int K, int LDS = 0>
__global__ __launch_bounds__(256) void walk(uint8_t* buf, size_t bytes)
{
	// LDS > 0: a static allocation that is never used, only to hold the CU to 160 KiB / LDS workgroups (the grain kernels' occupancy)
	if constexpr (LDS > 0)
	{
		__shared__ uint32_t pad[LDS / 4];
		if (bytes == 1) pad[threadIdx.x] = 0;      // (never true: keeps the allocation)
		if (bytes == 2) buf[0] = (uint8_t)pad[threadIdx.x ^ 1];
	}
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
	else if constexpr (MODE == 4)
	{
		// (a pair of waves shares 2 K items: the even one reads all of them, the odd one writes all of them)
		const size_t pair = wave >> 1;
		const size_t pbase = pair * (size_t)K * 8192;
		if (pbase >= bytes) return;
		const __amdgpu_buffer_rsrc_t pr = rsrc(buf + pbase, (uint32_t)std::min<size_t>(bytes - pbase, (size_t)K * 8192));
		if (wave & 1)
		{
			const u32x4 t = {lane16, 1u, 2u, 3u};
#pragma unroll 1
			for (int k = 0; k < 2 * K; k++)
#pragma unroll
				for (int u = 0; u < 4; u++) st(pr, k * 4096 + u * 1024 + lane16, t);
		}
		else
		{
			u32x4 acc = {0u, 0u, 0u, 0u};
			u32x4 v[4];
#pragma unroll
			for (int u = 0; u < 4; u++) v[u] = ld(pr, u * 1024 + lane16);
#pragma unroll 1
			for (int k = 0; k < 2 * K; k++)
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					acc ^= v[u];
					v[u] = ld(pr, k + 1 < 2 * K ? (k + 1) * 4096 + u * 1024 + lane16 : 0x80000000u);
					__builtin_amdgcn_sched_barrier(0);
				}
			if (acc.x == 0x12345u) st(pr, lane16, acc);     // (practically never: keeps the loads alive)
		}
	}
	else if constexpr (MODE == 5)
	{
		u32x4 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
		for (int k = 0; k < K; k += 2)
		{
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const u32x4 t = v[u] + 1u;
				const uint32_t here = k * 4096 + u * 1024 + lane16;
				st(r, here, t);
				v[u] = ld(r, k + 2 < K ? here + 8192 : 0x80000000u);
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

// the ring over regions of any size and place: wave w walks `region` bytes (a multiple of 1 KiB; the last group may be partly out of range: masked
// by the descriptor) from offset + w * pitch -- the grain kernel's rows (region = pitch = 15 KiB at 4320p) and what alignment is worth
__global__ __launch_bounds__(256) void walk_rows(uint8_t* buf, size_t bytes, uint32_t region, size_t pitch, size_t offset)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t base = offset + wave * pitch;
	if (base + region > bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, region);
	const int groups = (int)((region + 4095) / 4096);
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
	for (int k = 0; k < groups; k++)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const u32x4 t = v[u] + 1u;
			const uint32_t here = k * 4096 + u * 1024 + lane16;
			st(r, here, t);                                                   // (behind the region's end: dropped)
			v[u] = ld(r, k + 1 < groups ? here + 4096 : 0x80000000u);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// Round 6 (the verdict's question: is it the BYTES THE CHIP HAS OPEN AT ONCE -- resident waves x bytes a wave still has to walk -- that
// orders all of the results above?).  The ring again, K groups of 4 KiB per wave, but a wave's groups are NOT consecutive: the W waves
// of a "chunk" (W = the waves the chip holds at once, or a fraction / multiple of it) interleave their groups, wave w's group g sits at
// (g x W + w) x 4 KiB of the chunk.  Wave life, loads in flight per wave, instruction stream: those of `ring K`; what changes is that
// at any moment the W resident waves work inside ONE contiguous window of W x 4 KiB (instead of W x K x 4 KiB) that moves through the
// chunk as they advance together.  FRONTS = 2: two such windows (the chunk's two halves), as the grain kernel's two-frame fronts.
template <int K, int LDS, int FRONTS>
__global__ __launch_bounds__(256) void walk_sring(uint8_t* buf, size_t bytes, uint32_t W)
{
	if constexpr (LDS > 0)
	{
		__shared__ uint32_t pad[LDS / 4];
		if (bytes == 1) pad[threadIdx.x] = 0;
		if (bytes == 2) buf[0] = (uint8_t)pad[threadIdx.x ^ 1];
	}
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t chunk = wave / W;
	uint32_t w = (uint32_t)(wave % W);
	const size_t cbytes = (size_t)W * K * 4096;
	size_t cbase = chunk * cbytes;
	uint32_t Wf = W;
	if constexpr (FRONTS == 2)
	{
		// even waves sweep the chunk's first half, odd waves its second half
		Wf = W / 2;
		cbase += (size_t)(w & 1) * (cbytes / 2);
		w >>= 1;
	}
	if (cbase >= bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + cbase, (uint32_t)std::min<size_t>(bytes - cbase, cbytes / FRONTS));
	const uint32_t gstep = Wf * 4096u;
	u32x4 v[4];
#pragma unroll
	for (int u = 0; u < 4; u++) v[u] = ld(r, w * 4096u + u * 1024 + lane16);
#pragma unroll 1
	for (int k = 0; k < K; k++)
	{
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			const u32x4 t = v[u] + 1u;
			const uint32_t here = (uint32_t)k * gstep + w * 4096u + u * 1024 + lane16;
			st(r, here, t);
			v[u] = ld(r, k + 1 < K ? here + gstep : 0x80000000u);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// Round 6, second question: is it the DISTANCE between the load and the store of the same bytes?  The ring of four register sets stores a
// position four positions (plus the kernel's one step of lane rotation) after it asked for it; a wave needs one position per ~1.8 us, a
// load takes ~1 us: a shallower ring covers the latency too and writes each line back sooner.  DEPTH register sets, K groups of 4 KiB per wave,
// contiguous assignment; DEPTH = 4 is `ring`.
template <int K, int LDS, int DEPTH>
__global__ __launch_bounds__(256) void walk_depth(uint8_t* buf, size_t bytes)
{
	if constexpr (LDS > 0)
	{
		__shared__ uint32_t pad[LDS / 4];
		if (bytes == 1) pad[threadIdx.x] = 0;
		if (bytes == 2) buf[0] = (uint8_t)pad[threadIdx.x ^ 1];
	}
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t base = wave * (size_t)K * 4096;
	if (base >= bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, (uint32_t)std::min<size_t>(bytes - base, (size_t)K * 4096));
	constexpr int NPOS = K * 4;
	u32x4 v[DEPTH];
#pragma unroll
	for (int u = 0; u < DEPTH; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
	for (int p = 0; p < NPOS; p += DEPTH)
	{
#pragma unroll
		for (int u = 0; u < DEPTH; u++)
		{
			const u32x4 t = v[u] + 1u;
			const uint32_t here = (uint32_t)(p + u) * 1024 + lane16;
			st(r, p + u < NPOS ? here : 0x80000000u, t);
			v[u] = ld(r, p + u + DEPTH < NPOS ? here + DEPTH * 1024 : 0x80000000u);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// ... and with the kernel's ONE MORE position of delay: the lanes compute bytes shifted against the units they move, so a position's units are
// complete only when the NEXT position has been computed (vfgs_kernel.hip: lane rotation) -- the store of position p is issued at step p + 1.
template <int K, int LDS, int DEPTH>
__global__ __launch_bounds__(256) void walk_depth_lag(uint8_t* buf, size_t bytes)
{
	if constexpr (LDS > 0)
	{
		__shared__ uint32_t pad[LDS / 4];
		if (bytes == 1) pad[threadIdx.x] = 0;
		if (bytes == 2) buf[0] = (uint8_t)pad[threadIdx.x ^ 1];
	}
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t base = wave * (size_t)K * 4096;
	if (base >= bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, (uint32_t)std::min<size_t>(bytes - base, (size_t)K * 4096));
	constexpr int NPOS = K * 4;
	u32x4 v[DEPTH];
	u32x4 prev = {0u, 0u, 0u, 0u};
#pragma unroll
	for (int u = 0; u < DEPTH; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
	for (int p = 0; p < NPOS; p += DEPTH)
	{
#pragma unroll
		for (int u = 0; u < DEPTH; u++)
		{
			const u32x4 t = v[u] + 1u;                                      // (waits for position p + u)
			const uint32_t here = (uint32_t)(p + u) * 1024 + lane16;
			v[u] = ld(r, p + u + DEPTH < NPOS ? here + DEPTH * 1024 : 0x80000000u);
			st(r, p + u > 0 ? here - 1024 : 0x80000000u, prev);            // the position BEFORE this one
			prev = t;
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	st(r, (uint32_t)(NPOS - 1) * 1024 + lane16, prev);
}

// ... and with arithmetic between the load and the store, in the kernel's order: wait for position p, ask for position p + DEPTH, compute
// (`work` rounds of four dependent v_mad_u32_u24; the 10-bit luma kernel spends ~95 vector instructions per position: work = 24), store
// position p - 1.  Does the shallow ring keep its advantage when the wave has something to do while its load is in flight?
template <int K, int LDS, int DEPTH>
__global__ __launch_bounds__(256) void walk_depth_work(uint8_t* buf, size_t bytes, int work, int stages)
{
	static_assert(LDS >= 16384, "the gathers below index 16 KiB");
	__shared__ uint32_t pad[LDS / 4];
	for (int i = threadIdx.x; i < 4096; i += 256) pad[i] = 0;      // (zeros: the gathers below leave the data alone)
	__syncthreads();
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	const size_t base = wave * (size_t)K * 4096;
	if (base >= bytes) return;
	const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, (uint32_t)std::min<size_t>(bytes - base, (size_t)K * 4096));
	constexpr int NPOS = K * 4;
	u32x4 v[DEPTH];
	u32x4 prev = {0u, 0u, 0u, 0u};
#pragma unroll
	for (int u = 0; u < DEPTH; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll 1
	for (int p = 0; p < NPOS; p += DEPTH)
	{
#pragma unroll
		for (int u = 0; u < DEPTH; u++)
		{
			u32x4 t = v[u] + 1u;
			const uint32_t here = (uint32_t)(p + u) * 1024 + lane16;
			v[u] = ld(r, p + u + DEPTH < NPOS ? here + DEPTH * 1024 : 0x80000000u);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
			for (int i = 0; i < work; i++)
			{
				t.x = __umul24(t.x, 0x10001u) + 0x9e37u; t.y = __umul24(t.y, 0x10001u) + 0x79b9u;
				t.z = __umul24(t.z, 0x10001u) + 0x7f4au; t.w = __umul24(t.w, 0x10001u) + 0x7c15u;
			}
			// `stages` dependent round trips to LDS, four scattered dword gathers each (the kernel: block parameters -> scale / pattern
			// look-ups -> bank rows: three to four)
#pragma unroll 1
			for (int i = 0; i < stages; i++)
			{
				const uint32_t x = pad[t.x & 4095], y = pad[t.y & 4095], z = pad[t.z & 4095], q = pad[t.w & 4095];
				t.x += y; t.y += z; t.z += q; t.w += x;
			}
			__builtin_amdgcn_sched_barrier(0);
			st(r, p + u > 0 ? here - 1024 : 0x80000000u, prev);
			prev = t;
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	st(r, (uint32_t)(NPOS - 1) * 1024 + lane16, prev);
}

template <int K, int LDS, int DEPTH>
static double run_depth_work(uint8_t* buf, size_t bytes, int reps, int work, int stages = 0)
{
	const size_t waves = (bytes + (size_t)K * 4096 - 1) / ((size_t)K * 4096);
	const unsigned grid = (unsigned)((waves + 3) / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((walk_depth_work<K, LDS, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, bytes, work, stages);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((walk_depth_work<K, LDS, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, bytes, work, stages);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

template <int K, int LDS, int DEPTH, bool LAG = false>
static double run_depth(uint8_t* buf, size_t bytes, int reps)
{
	const size_t waves = (bytes + (size_t)K * 4096 - 1) / ((size_t)K * 4096);
	const unsigned grid = (unsigned)((waves + 3) / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	auto launch = [&] {
		if constexpr (LAG) hipLaunchKernelGGL((walk_depth_lag<K, LDS, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, bytes);
		else hipLaunchKernelGGL((walk_depth<K, LDS, DEPTH>), dim3(grid), dim3(256), 0, 0, buf, bytes);
	};
	for (int i = 0; i < 3; i++) launch();
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		launch();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

template <int K, int LDS, int FRONTS>
static double run_sring(uint8_t* buf, size_t bytes, int reps, uint32_t W)
{
	const size_t cbytes = (size_t)W * K * 4096;
	const size_t use = bytes / cbytes * cbytes;          // whole chunks only
	const size_t waves = use / ((size_t)K * 4096);
	const unsigned grid = (unsigned)(waves / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((walk_sring<K, LDS, FRONTS>), dim3(grid), dim3(256), 0, 0, buf, use, W);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((walk_sring<K, LDS, FRONTS>), dim3(grid), dim3(256), 0, 0, buf, use, W);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * use / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

static double run_rows(uint8_t* buf, size_t bytes, int reps, uint32_t region, size_t pitch, size_t offset)
{
	const size_t waves = (bytes - offset) / pitch;
	const unsigned grid = (unsigned)((waves + 3) / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL(walk_rows, dim3(grid), dim3(256), 0, 0, buf, bytes, region, pitch, offset);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL(walk_rows, dim3(grid), dim3(256), 0, 0, buf, bytes, region, pitch, offset);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * (double)waves * region / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

__global__ __launch_bounds__(256) void walk_stride(uint8_t* buf, size_t bytes)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const uint32_t lane16 = (threadIdx.x & 63) * 16;
	for (size_t base = wave * 4096; base < bytes; base += nwaves * 4096)
	{
		const __amdgpu_buffer_rsrc_t r = rsrc(buf + base, (uint32_t)std::min<size_t>(bytes - base, 4096));
		u32x4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++) v[u] = ld(r, u * 1024 + lane16);
#pragma unroll
		for (int u = 0; u < 4; u++) st(r, u * 1024 + lane16, v[u] + 1u);
	}
}

static double run_stride(uint8_t* buf, size_t bytes, int reps, int wg_per_cu)
{
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const unsigned grid = (unsigned)(prop.multiProcessorCount * wg_per_cu);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL(walk_stride, dim3(grid), dim3(256), 0, 0, buf, bytes);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL(walk_stride, dim3(grid), dim3(256), 0, 0, buf, bytes);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float t;
		hipEventElapsedTime(&t, e0, e1);
		ms.push_back(t);
	}
	std::sort(ms.begin(), ms.end());
	return 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e9;
}

template <int MODE, int K, int LDS = 0>
static double run(uint8_t* buf, size_t bytes, int reps)
{
	const size_t waves = (bytes + (size_t)K * 4096 - 1) / ((size_t)K * 4096);
	const unsigned grid = (unsigned)((waves + 3) / 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 3; i++) hipLaunchKernelGGL((walk<MODE, K, LDS>), dim3(grid), dim3(256), 0, 0, buf, bytes);
	std::vector<float> ms;
	for (int rep = 0; rep < reps; rep++)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((walk<MODE, K, LDS>), dim3(grid), dim3(256), 0, 0, buf, bytes);
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
	const bool window_only = argc > 2 && !strcmp(argv[2], "window");
	uint8_t* buf;
	if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
	hipMemset(buf, 1, bytes);
	hipDeviceSynchronize();
	const int reps = 15;
	printf("buffer %zu MiB, in place, nontemporal 16-byte accesses, GB/s (read + write) and fraction of 8 TB/s, median of %d launches\n", bytes >> 20, reps);
#define LINE(name, MODE, K) { const double g = run<MODE, K>(buf, bytes, reps); printf("%-10s K=%-3d %8.1f  %.4f\n", name, K, g, g / 8000.0); fflush(stdout); }
	{
		// ---- round 6: the open window.  "open MiB" = waves the chip holds x the bytes a wave walks (what a contiguous assignment keeps
		// open); "window MiB" = the contiguous region the resident waves work in at one moment
		hipDeviceProp_t prop;
		hipGetDeviceProperties(&prop, 0);
		const int cus = prop.multiProcessorCount;
		const bool depth_only = argc > 2 && !strcmp(argv[2], "depth");
#define LINEW(name, K, LDS, FRONTS, WMUL) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const uint32_t W = (uint32_t)(cus * wpc * WMUL); \
			const double g = run_sring<K, LDS, FRONTS>(buf, bytes, reps, W); \
			printf("%-8s K=%-2d %2d waves/CU  W = %5.2f x resident  fronts %d   open %6.1f MiB  window %6.1f MiB  %8.1f  %.4f\n", name, K, wpc, (double)WMUL, FRONTS, \
			       cus * wpc * K * 4096.0 / 1048576, W * 4096.0 / FRONTS / 1048576, g, g / 8000.0); fflush(stdout); }
#define LINEC(name, MODE, K, LDS) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const double g = run<MODE, K, LDS>(buf, bytes, reps); \
			printf("%-8s K=%-2d %2d waves/CU  contiguous                        open %6.1f MiB  window %6.1f MiB  %8.1f  %.4f\n", name, K, wpc, \
			       cus * wpc * K * 4096.0 / 1048576, cus * wpc * K * 4096.0 / 1048576, g, g / 8000.0); fflush(stdout); }
		for (int round = 0; round < (depth_only ? 0 : 2); round++)
		{
			LINEC("item", 0, 1, 0) LINEC("ring", 1, 2, 0) LINEC("ring", 1, 4, 0) LINEC("ring", 1, 16, 0)
			LINEC("item", 0, 1, 40960) LINEC("ring", 1, 2, 40960) LINEC("ring", 1, 4, 40960) LINEC("ring", 1, 16, 40960) LINEC("ring", 1, 4, 54608) LINEC("ring", 1, 4, 81920)
			LINEW("sring", 4, 0, 1, 1) LINEW("sring", 4, 0, 1, 0.5) LINEW("sring", 4, 0, 1, 2) LINEW("sring", 16, 0, 1, 1) LINEW("sring", 2, 0, 1, 1)
			LINEW("sring", 4, 40960, 1, 1) LINEW("sring", 4, 40960, 1, 0.5) LINEW("sring", 4, 40960, 1, 0.25) LINEW("sring", 4, 40960, 1, 2) LINEW("sring", 4, 40960, 1, 4)
			LINEW("sring", 16, 40960, 1, 1) LINEW("sring", 16, 40960, 1, 0.5) LINEW("sring", 2, 40960, 1, 1)
			LINEW("sring", 4, 40960, 2, 1) LINEW("sring", 4, 40960, 2, 2) LINEW("sring", 16, 40960, 2, 1)
			LINEW("sring", 4, 54608, 1, 1) LINEW("sring", 4, 81920, 1, 1)
		}
		if (argc > 2 && !strcmp(argv[2], "depth"))
		{
#define LINED(K, LDS, DEPTH) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const double g = run_depth<K, LDS, DEPTH>(buf, bytes, reps); \
			printf("depth %d  K=%-2d %2d waves/CU   a position is stored %d positions after its load was issued  %8.1f  %.4f\n", DEPTH, K, wpc, DEPTH, g, g / 8000.0); fflush(stdout); }
			for (int round = 0; round < 2; round++)
			{
				LINED(4, 40960, 1) LINED(4, 40960, 2) LINED(4, 40960, 4) LINED(4, 40960, 8)
				LINED(4, 32768, 1) LINED(4, 32768, 2) LINED(4, 32768, 4)
				LINED(4, 0, 1) LINED(4, 0, 2) LINED(4, 0, 4)
#define LINEDL(K, LDS, DEPTH) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const double g = run_depth<K, LDS, DEPTH, true>(buf, bytes, reps); \
			printf("depth %d  K=%-2d %2d waves/CU   ... and stored one position later still (the kernel's lane rotation): %d positions after its load  %8.1f  %.4f\n", DEPTH, K, wpc, DEPTH + 1, g, g / 8000.0); fflush(stdout); }
				LINEDL(4, 40960, 1) LINEDL(4, 40960, 2) LINEDL(4, 40960, 4) LINEDL(4, 32768, 1) LINEDL(4, 32768, 2) LINEDL(4, 32768, 4)
#define LINEDW(K, LDS, DEPTH, WORK) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const double g = run_depth_work<K, LDS, DEPTH>(buf, bytes, reps, WORK); \
			printf("depth %d  K=%-2d %2d waves/CU   kernel order, %3d vector instructions per position between load and store  %8.1f  %.4f\n", DEPTH, K, wpc, 4 * WORK, g, g / 8000.0); fflush(stdout); }
				LINEDW(4, 32768, 1, 0) LINEDW(4, 32768, 1, 12) LINEDW(4, 32768, 1, 24) LINEDW(4, 32768, 1, 48) LINEDW(4, 32768, 1, 96)
				LINEDW(4, 32768, 2, 0) LINEDW(4, 32768, 2, 12) LINEDW(4, 32768, 2, 24) LINEDW(4, 32768, 2, 48) LINEDW(4, 32768, 2, 96)
				LINEDW(4, 32768, 4, 0) LINEDW(4, 32768, 4, 12) LINEDW(4, 32768, 4, 24) LINEDW(4, 32768, 4, 48) LINEDW(4, 32768, 4, 96)
				LINEDW(2, 32768, 1, 24) LINEDW(2, 32768, 2, 24) LINEDW(2, 32768, 4, 24)
#define LINEDS(K, LDS, DEPTH, WORK, ST) { const int wpc = LDS ? 4 * (163840 / LDS) : 32; const double g = run_depth_work<K, LDS, DEPTH>(buf, bytes, reps, WORK, ST); \
			printf("depth %d  K=%-2d %2d waves/CU   kernel order, %3d vector instructions and %d dependent LDS round trips per position  %8.1f  %.4f\n", DEPTH, K, wpc, 4 * WORK, ST, g, g / 8000.0); fflush(stdout); }
				LINEDS(4, 32768, 1, 16, 2) LINEDS(4, 32768, 1, 16, 4) LINEDS(4, 32768, 1, 16, 8)
				LINEDS(4, 32768, 2, 16, 2) LINEDS(4, 32768, 2, 16, 4) LINEDS(4, 32768, 2, 16, 8)
				LINEDS(4, 32768, 4, 16, 2) LINEDS(4, 32768, 4, 16, 4) LINEDS(4, 32768, 4, 16, 8)
				// fewer waves, deeper rings: the same loads in flight per CU
				LINEDS(4, 53248, 1, 16, 2) LINEDS(4, 53248, 2, 16, 2) LINEDS(4, 53248, 4, 16, 2)
				LINEDS(4, 65536, 1, 16, 2) LINEDS(4, 65536, 2, 16, 2) LINEDS(4, 65536, 4, 16, 2) LINEDS(8, 65536, 8, 16, 2)
				LINEDS(8, 32768, 1, 16, 2) LINEDS(8, 32768, 2, 16, 2) LINEDS(8, 32768, 4, 16, 2)
				LINEDS(8, 65536, 1, 16, 2) LINEDS(8, 65536, 2, 16, 2) LINEDS(8, 65536, 4, 16, 2)
				LINED(16, 40960, 1) LINED(16, 40960, 2) LINED(16, 40960, 4)
				LINED(2, 40960, 1) LINED(2, 40960, 2) LINED(2, 40960, 4)
			}
			return 0;
		}
		if (window_only) return 0;
	}
	for (int round = 0; round < 2; round++)
	{
		LINE("item", 0, 1)
		LINE("seq", 0, 2) LINE("seq", 0, 4) LINE("seq", 0, 16)
		LINE("ring", 1, 2) LINE("ring", 1, 4) LINE("ring", 1, 16)
		LINE("ringw", 3, 4) LINE("ringw", 3, 16)
		LINE("dbl", 2, 2) LINE("dbl", 2, 4) LINE("dbl", 2, 16)
		for (int g : {1, 2, 3, 4, 8}) { const double gb = run_stride(buf, bytes, reps, g); printf("%-10s G=%-3d %8.1f  %.4f\n", "stride", g, gb, gb / 8000.0); fflush(stdout); }
		LINE("ring", 1, 3) LINE("ring", 1, 5) LINE("ring", 1, 6) LINE("ring", 1, 7) LINE("ring", 1, 8) LINE("ring", 1, 15)      // (regions that are not powers of two)
		struct { const char* name; uint32_t region; size_t pitch, offset; } rows[] = {
			{"4K/4K", 4096, 4096, 0}, {"4K/4K+1K", 4096, 4096, 1024}, {"4K/4K+256", 4096, 4096, 256}, {"3K/4K", 3072, 4096, 0}, {"4K/5K", 4096, 5120, 0},
			{"8K/8K", 8192, 8192, 0}, {"7.5K/7.5K", 7680, 7680, 0}, {"7.5K/8K", 7680, 8192, 0},
			{"16K/16K", 16384, 16384, 0}, {"15K/15K", 15360, 15360, 0}, {"15K/16K", 15360, 16384, 0}, {"16K/16K+1K", 16384, 16384, 1024}, {"16K/17K", 16384, 17408, 0},
			{"32K/32K", 32768, 32768, 0}, {"30K/30K", 30720, 30720, 0}, {"30K/32K", 30720, 32768, 0}, {"3.75K/3.75K", 3840, 3840, 0}, {"3.75K/4K", 3840, 4096, 0}};
		for (auto& q : rows) { const double gb = run_rows(buf, bytes, reps, q.region, q.pitch, q.offset); printf("%-10s %-12s %8.1f  %.4f\n", "rows", q.name, gb, gb / 8000.0); fflush(stdout); }
		LINE("ring8", 5, 2) LINE("ring8", 5, 4) LINE("ring8", 5, 16)
		LINE("split", 4, 1) LINE("split", 4, 2) LINE("split", 4, 4) LINE("split", 4, 16)
	}
	// the same streams at the occupancy of the grain kernels: 160 KiB / LDS workgroups of four waves per CU
#define LINEL(name, MODE, K, LDS) { const double g = run<MODE, K, LDS>(buf, bytes, reps); printf("%-10s K=%-3d %2d waves/CU %8.1f  %.4f\n", name, K, 4 * (163840 / LDS), g, g / 8000.0); fflush(stdout); }
	for (int round = 0; round < 2; round++)
	{
		LINEL("item", 0, 1, 20480) LINEL("item", 0, 1, 27304) LINEL("item", 0, 1, 32768) LINEL("item", 0, 1, 40960) LINEL("item", 0, 1, 54608) LINEL("item", 0, 1, 81920)
		LINEL("ring", 1, 2, 20480) LINEL("ring", 1, 2, 27304) LINEL("ring", 1, 2, 32768) LINEL("ring", 1, 2, 40960) LINEL("ring", 1, 2, 54608)
		LINEL("split", 4, 4, 20480) LINEL("split", 4, 4, 40960) LINEL("split", 4, 4, 81920) LINEL("ring8", 5, 4, 40960) LINEL("ring8", 5, 4, 54608)
		LINEL("ring", 1, 4, 20480) LINEL("ring", 1, 4, 27304) LINEL("ring", 1, 4, 32768) LINEL("ring", 1, 4, 40960) LINEL("ring", 1, 4, 54608) LINEL("ring", 1, 4, 81920)
	}
	return 0;
}
