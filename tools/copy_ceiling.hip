// Developer microbenchmark: what a pure streaming kernel reaches on this chip, in the two
// shapes that matter here: out-of-place copy and in-place read-modify-write, 16 B per lane.
// hipcc --offload-arch=gfx950 -O3 -o copy_ceiling copy_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void k_copy(const u32x4* src, u32x4* dst, size_t n)
{
	size_t stride = (size_t)gridDim.x * blockDim.x;
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride)
	{
		u32x4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) dst[i + u * stride] = v[u] + 1u;
	}
	for (; i < n; i += stride) dst[i] = src[i] + 1u;
}

// contiguous 1 KiB per wave per access, UNROLL accesses (rows) per wave like the grain kernel
template <int UNROLL>
__global__ void k_rmw_rows(u32x4* __restrict__ buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	for (size_t base = wave * 64 * UNROLL; base + 64 * UNROLL <= n; base += nwaves * 64 * UNROLL)
	{
		u32x4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) v[u] = buf[base + u * 64 + lane];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) buf[base + u * 64 + lane] = v[u] + 1u;
	}
}

// same with raw buffer instructions (range-checked), optionally a few lanes out of range
template <int UNROLL, bool OOB>
__global__ void k_rmw_rows_buf(u32x4* buf, size_t n)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
	const int lane = threadIdx.x & 63;
	__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)(n * 16), 0x00020000);
	for (size_t base = wave * 64 * UNROLL; base + 64 * UNROLL <= n; base += nwaves * 64 * UNROLL)
	{
		u32x4 v[UNROLL];
		uint32_t off[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++)
		{
			off[u] = (uint32_t)((base + u * 64 + lane) * 16);
			if (OOB && lane >= 62) off[u] = 0x80000000u;
			v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[u], 0, 0);
		}
#pragma unroll
		for (int u = 0; u < UNROLL; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u] + 1u, rs, off[u], 0, 0);
	}
}

// the grain kernel's actual shape: a wave moves `rows` row segments of `upt` 16-byte units that
// start 16 bytes before a multiple of upt*16 (half-block shift), rows `pitch` bytes apart
template <int ROWS>
__global__ void k_tiles(const uint8_t* src, uint8_t* dst, int pitch, int ntx, int upt, int nrowgroups, int shift)
{
	const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	for (int item = wave; item < ntx * nrowgroups; item += nwaves)
	{
		const int tx = item % ntx, rg = item / ntx;
		const long off = (long)(tx * upt + lane) * 16 - shift;
		const bool ok = lane < upt && off >= 0 && off + 16 <= pitch;
		u32x4 v[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++) if (ok) v[r] = *(const u32x4*)(src + (size_t)(rg * ROWS + r) * pitch + off);
#pragma unroll
		for (int r = 0; r < ROWS; r++) if (ok) *(u32x4*)(dst + (size_t)(rg * ROWS + r) * pitch + off) = v[r] + 1u;
	}
}

// chroma-shaped rows: 8-byte (not 16-byte) aligned 16-byte pieces, as one b128 or as two b64 accesses
template <int ROWS, bool SPLIT>
__global__ void k_tiles8(const uint8_t* src, uint8_t* dst, int pitch, int ntx, int upt, int nrowgroups)
{
	typedef uint32_t v4a8 __attribute__((ext_vector_type(4), aligned(8)));
	typedef uint32_t v2 __attribute__((ext_vector_type(2)));
	const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	for (int item = wave; item < ntx * nrowgroups; item += nwaves)
	{
		const int tx = item % ntx, rg = item / ntx;
		const long off = (long)(tx * upt + lane) * 16 - 8;
		const bool ok = lane < upt && off >= 0 && off + 16 <= pitch;
		u32x4 v[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++)
			if (ok)
			{
				const uint8_t* p = src + (size_t)(rg * ROWS + r) * pitch + off;
				if (SPLIT) { v2 a = *(const v2*)p, b2 = *(const v2*)(p + 8); v[r] = u32x4{a.x, a.y, b2.x, b2.y}; }
				else v[r] = *(const v4a8*)p;
			}
#pragma unroll
		for (int r = 0; r < ROWS; r++)
			if (ok)
			{
				uint8_t* p = dst + (size_t)(rg * ROWS + r) * pitch + off;
				u32x4 o = v[r] + 1u;
				if (SPLIT) { *(v2*)p = v2{o.x, o.y}; *(v2*)(p + 8) = v2{o.z, o.w}; }
				else *(v4a8*)p = o;
			}
	}
}

int main()
{
	const size_t bytes = 199065600ull / 2;       // one 4320p 10-bit 4:2:0 frame
	const size_t n = bytes / 16;
	const int nbuf = 16;                          // cycle through > 256 MiB so the Infinity Cache cannot hold it
	u32x4 *a[nbuf], *bb[nbuf];
	for (int i = 0; i < nbuf; i++) { hipMalloc(&a[i], bytes); hipMemset(a[i], 1, bytes); hipMalloc(&bb[i], bytes); hipMemset(bb[i], 2, bytes); }
#define b bb[i]
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	auto report = [&](const char* name, int blocks, int threads, float ms, int reps) {
		printf("%-28s grid %5d x %4d : %7.2f us  %7.1f GB/s (read+write)\n", name, blocks, threads, ms / reps * 1e3, 2.0 * bytes / (ms / reps * 1e-3) / 1e9);
	};
	const int reps = 64;
	for (int wpc : {8, 16, 32})
	{
		const int threads = 256, blocks = cus * wpc / 4;
		float ms;
		auto run = [&](const char* name, auto launch) {
			for (int r = 0; r < 8; r++) launch(r % nbuf);
			hipEventRecord(e0);
			for (int r = 0; r < reps; r++) launch(r % nbuf);
			hipEventRecord(e1); hipEventSynchronize(e1);
			hipEventElapsedTime(&ms, e0, e1); report(name, blocks, threads, ms, reps);
		};
		run("copy out-of-place x1", [&](int i) { k_copy<1><<<blocks, threads>>>(a[i], b, n); });
		run("copy out-of-place x4", [&](int i) { k_copy<4><<<blocks, threads>>>(a[i], b, n); });
		run("rmw in-place x1 (grid-stride)", [&](int i) { k_copy<1><<<blocks, threads>>>(a[i], a[i], n); });
		run("rmw in-place x4 (grid-stride)", [&](int i) { k_copy<4><<<blocks, threads>>>(a[i], a[i], n); });
		run("rmw in-place rows x1", [&](int i) { k_rmw_rows<1><<<blocks, threads>>>(a[i], n); });
		run("rmw in-place rows x4", [&](int i) { k_rmw_rows<4><<<blocks, threads>>>(a[i], n); });
		run("rmw in-place rows x6", [&](int i) { k_rmw_rows<6><<<blocks, threads>>>(a[i], n); });
		{
			const int pitch = 15360, rows = 6480;   // 4320 luma rows + chroma as 2160 more rows of the same pitch
			run("tiles 62u shift16 in-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 16, 62, rows / 4, 16); });
			run("tiles 62u shift16 out-of-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)b, pitch, 16, 62, rows / 4, 16); });
			run("tiles 64u aligned in-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 15, 64, rows / 4, 0); });
			run("tiles 64u aligned out-of-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)b, pitch, 15, 64, rows / 4, 0); });
			run("tiles 64u shift16 out-of-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)b, pitch, 16, 64, rows / 4, 16); });
			run("tiles 62u shift8 b128 in-place", [&](int i) { k_tiles8<4, false><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 16, 62, rows / 4); });
			run("tiles 62u shift8 2xb64 in-place", [&](int i) { k_tiles8<4, true><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 16, 62, rows / 4); });
			run("tiles 60u aligned-ish out-of-place", [&](int i) { k_tiles<4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)b, pitch, 16, 60, rows / 4, 0); });
		}
		run("rmw rows x4 buffer ops", [&](int i) { k_rmw_rows_buf<4, false><<<blocks, threads>>>(a[i], n); });
		run("rmw rows x6 buffer ops", [&](int i) { k_rmw_rows_buf<6, false><<<blocks, threads>>>(a[i], n); });
		run("rmw rows x6 buffer ops, 2 lanes OOB", [&](int i) { k_rmw_rows_buf<6, true><<<blocks, threads>>>(a[i], n); });
	}
	return 0;
}
