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

// ROWS rows x SEGS consecutive segments of `upt` 16-byte units per wave item
template <int ROWS, int SEGS>
__global__ void k_tiles_seg(const uint8_t* src, uint8_t* dst, int pitch, int ntx, int upt, int nrowgroups, int shift)
{
	const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	for (int item = wave; item < ntx * nrowgroups; item += nwaves)
	{
		const int tx = item % ntx, rg = item / ntx;
		u32x4 v[ROWS][SEGS];
		bool ok[SEGS];
		long off[SEGS];
#pragma unroll
		for (int g = 0; g < SEGS; g++)
		{
			off[g] = (long)((tx * SEGS + g) * upt + lane) * 16 - shift;
			ok[g] = lane < upt && off[g] >= 0 && off[g] + 16 <= pitch;
		}
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int g = 0; g < SEGS; g++) if (ok[g]) v[r][g] = *(const u32x4*)(src + (size_t)(rg * ROWS + r) * pitch + off[g]);
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int g = 0; g < SEGS; g++) if (ok[g]) *(u32x4*)(dst + (size_t)(rg * ROWS + r) * pitch + off[g]) = v[r][g] + 1u;
	}
}

// the grain kernel's exact item: 4 luma rows x 62 units (shift 16 B) + U and V: 2 rows x 31 units (shift 8 B), 3 planes
// CMODE 0: chroma 2 rows x 31 lanes per access;  1: chroma items cover 2 luma tiles: 1 row x 62 lanes per access, 2 accesses
template <int CMODE>
__global__ void k_item420(uint8_t* Y, uint8_t* U, uint8_t* V, int ypitch, int cpitch, int ntx, int nquads)
{
	typedef uint32_t v4a8 __attribute__((ext_vector_type(4), aligned(8)));
	const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	for (int item = wave; item < ntx * nquads; item += nwaves)
	{
		const int tx = item % ntx, q = item / ntx;
		const long yoff = (long)(tx * 62 + lane) * 16 - 16;
		const bool yok = lane < 62 && yoff >= 0 && yoff + 16 <= ypitch;
		long coff; bool cok; int crow;
		if (CMODE == 0) { coff = (long)(tx * 31 + (lane & 31)) * 16 - 8; cok = (lane & 31) < 31 && coff >= 0 && coff + 16 <= cpitch; crow = 2 * q + (lane >> 5); }
		else { coff = (long)((tx & ~1) * 31 + lane) * 16 - 8; cok = lane < 62 && coff >= 0 && coff + 16 <= cpitch; crow = 2 * q + (tx & 1); }
		u32x4 vy[4], vu, vv;
#pragma unroll
		for (int r = 0; r < 4; r++) if (yok) vy[r] = *(const u32x4*)(Y + (size_t)(4 * q + r) * ypitch + yoff);
		if (cok) { vu = *(const v4a8*)(U + (size_t)crow * cpitch + coff); vv = *(const v4a8*)(V + (size_t)crow * cpitch + coff); }
#pragma unroll
		for (int r = 0; r < 4; r++) if (yok) *(u32x4*)(Y + (size_t)(4 * q + r) * ypitch + yoff) = vy[r] + 1u;
		if (cok) { *(v4a8*)(U + (size_t)crow * cpitch + coff) = vu + 1u; *(v4a8*)(V + (size_t)crow * cpitch + coff) = vv + 1u; }
	}
}

// branch-free (raw buffer ops, out-of-range lanes dropped by the hardware) versions of the tile shapes:
// a wave item = ROWS rows x SEGS segments of `upt` 16-byte units, shifted by `shift` bytes; rows `pitch` apart
template <int ROWS, int SEGS>
__global__ void k_tiles_buf(uint8_t* buf, uint32_t bytes, int pitch, int ntx, int upt, int nrowgroups, int shift)
{
	const int wave = __builtin_amdgcn_readfirstlane((int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	__amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)bytes, 0x00020000);
	for (int item = wave; item < ntx * nrowgroups; item += nwaves)
	{
		const int tx = item % ntx, rg = item / ntx;
		u32x4 v[ROWS][SEGS];
		uint32_t off[ROWS][SEGS];
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int g = 0; g < SEGS; g++)
			{
				const int x = ((tx * SEGS + g) * upt + lane) * 16 - shift;
				const bool ok = lane < upt && x >= 0 && x + 16 <= pitch;
				off[r][g] = ok ? (uint32_t)((rg * ROWS + r) * pitch + x) : 0x80000000u;
				v[r][g] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[r][g], 0, 0);
			}
#pragma unroll
		for (int r = 0; r < ROWS; r++)
#pragma unroll
			for (int g = 0; g < SEGS; g++) __builtin_amdgcn_raw_buffer_store_b128(v[r][g] + 1u, rs, off[r][g], 0, 0);
	}
}

// branch-free exact 4:2:0 items over three planes.  MODE 0: 4 luma rows x 62u + U,V as 2 rows x 31 lanes (8 B shift);
// MODE 1: 2 luma rows x 2 x 62u + U,V as 1 row x 62 lanes
template <int MODE>
__global__ void k_item420_buf(uint8_t* Yb, uint32_t ybytes, uint8_t* Ub, uint8_t* Vb, uint32_t cbytes, int ypitch, int cpitch, int nblk2)
{
	const int wave = __builtin_amdgcn_readfirstlane((int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	__amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yb, 0, (int)ybytes, 0x00020000);
	__amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)Ub, 0, (int)cbytes, 0x00020000);
	__amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)cbytes, 0x00020000);
	const int ntx = MODE == 0 ? 16 : 8, ngroups = MODE == 0 ? 1080 : 2160;
	for (int item = wave; item < ntx * ngroups; item += nwaves)
	{
		const int tx = item % ntx, q = item / ntx;
		u32x4 vy[4], vu, vv;
		uint32_t oy[4], oc;
		if (MODE == 0)
		{
			const int x = (tx * 62 + lane) * 16 - 16;
			const bool ok = lane < 62 && x >= 0 && x + 16 <= ypitch;
#pragma unroll
			for (int r = 0; r < 4; r++) oy[r] = ok ? (uint32_t)((4 * q + r) * ypitch + x) : 0x80000000u;
			const int xc = (tx * 31 + (lane & 31)) * 16 - 8;
			const bool okc = (lane & 31) < 31 && xc >= 0 && xc + 16 <= cpitch;
			oc = okc ? (uint32_t)((2 * q + (lane >> 5)) * cpitch + xc) : 0x80000000u;
		}
		else
		{
#pragma unroll
			for (int r = 0; r < 4; r++)
			{
				const int x = ((tx * 2 + (r & 1)) * 62 + lane) * 16 - 16;
				const bool ok = lane < 62 && x >= 0 && x + 16 <= ypitch;
				oy[r] = ok ? (uint32_t)((2 * q + (r >> 1)) * ypitch + x) : 0x80000000u;
			}
			const int xc = (tx * 62 + lane) * 16 - 8;
			const bool okc = lane < 62 && xc >= 0 && xc + 16 <= cpitch;
			oc = okc ? (uint32_t)(q * cpitch + xc) : 0x80000000u;
		}
#pragma unroll
		for (int r = 0; r < 4; r++) vy[r] = __builtin_amdgcn_raw_buffer_load_b128(ry, oy[r], 0, 0);
		vu = __builtin_amdgcn_raw_buffer_load_b128(ru, oc, 0, 0);
		vv = __builtin_amdgcn_raw_buffer_load_b128(rv, oc, 0, 0);
#pragma unroll
		for (int r = 0; r < 4; r++) __builtin_amdgcn_raw_buffer_store_b128(vy[r] + 1u, ry, oy[r], 0, 0);
		__builtin_amdgcn_raw_buffer_store_b128(vu + 1u, ru, oc, 0, 0);
		__builtin_amdgcn_raw_buffer_store_b128(vv + 1u, rv, oc, 0, 0);
	}
}

// plane-specialised items: every item is ONE row x 4 consecutive segments of 62 units in ONE plane
// (luma rows shifted 16 B, chroma rows shifted 8 B); item order: all Y rows, then U rows, then V rows, or interleaved per row group
template <int ORDER>
__global__ void k_item_planes(uint8_t* Yb, uint32_t ybytes, uint8_t* Ub, uint8_t* Vb, uint32_t cbytes, int ypitch, int cpitch)
{
	const int wave = __builtin_amdgcn_readfirstlane((int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int nwaves = (int)(((size_t)gridDim.x * blockDim.x) >> 6);
	const int lane = threadIdx.x & 63;
	__amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yb, 0, (int)ybytes, 0x00020000);
	__amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)Ub, 0, (int)cbytes, 0x00020000);
	__amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)cbytes, 0x00020000);
	const int yitems = 4320 * 4, citems = 2160 * 2;
	for (int item = wave; item < yitems + 2 * citems; item += nwaves)
	{
		int plane, row, tx;
		if (ORDER == 0)
		{
			if (item < yitems) { plane = 0; row = item / 4; tx = item % 4; }
			else if (item < yitems + citems) { plane = 1; row = (item - yitems) / 2; tx = (item - yitems) % 2; }
			else { plane = 2; row = (item - yitems - citems) / 2; tx = (item - yitems - citems) % 2; }
		}
		else
		{   // per luma row pair: 8 Y items, 2 U items, 2 V items
			const int g = item / 12, k = item % 12;
			if (k < 8) { plane = 0; row = 2 * g + k / 4; tx = k % 4; }
			else if (k < 10) { plane = 1; row = g; tx = k - 8; }
			else { plane = 2; row = g; tx = k - 10; }
		}
		const int pitch = plane ? cpitch : ypitch, shift = plane ? 8 : 16;
		u32x4 v[4];
		uint32_t off[4];
#pragma unroll
		for (int g = 0; g < 4; g++)
		{
			const int x = ((tx * 4 + g) * 62 + lane) * 16 - shift;
			const bool ok = lane < 62 && x >= 0 && x + 16 <= pitch;
			off[g] = ok ? (uint32_t)(row * pitch + x) : 0x80000000u;
		}
		if (plane == 0) {
#pragma unroll
			for (int g = 0; g < 4; g++) v[g] = __builtin_amdgcn_raw_buffer_load_b128(ry, off[g], 0, 0);
#pragma unroll
			for (int g = 0; g < 4; g++) __builtin_amdgcn_raw_buffer_store_b128(v[g] + 1u, ry, off[g], 0, 0);
		} else if (plane == 1) {
#pragma unroll
			for (int g = 0; g < 4; g++) v[g] = __builtin_amdgcn_raw_buffer_load_b128(ru, off[g], 0, 0);
#pragma unroll
			for (int g = 0; g < 4; g++) __builtin_amdgcn_raw_buffer_store_b128(v[g] + 1u, ru, off[g], 0, 0);
		} else {
#pragma unroll
			for (int g = 0; g < 4; g++) v[g] = __builtin_amdgcn_raw_buffer_load_b128(rv, off[g], 0, 0);
#pragma unroll
			for (int g = 0; g < 4; g++) __builtin_amdgcn_raw_buffer_store_b128(v[g] + 1u, rv, off[g], 0, 0);
		}
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
			run("tiles 2rows x 2x62u shift16 in-place", [&](int i) { k_tiles_seg<2, 2><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 8, 62, rows / 2, 16); });
			run("tiles 1row x 4x62u shift16 in-place", [&](int i) { k_tiles_seg<1, 4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 4, 62, rows, 16); });
			run("tiles 1row x 4x64u aligned in-place", [&](int i) { k_tiles_seg<1, 4><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 4, 64, rows, 0); });
			run("tiles 1row x 6x62u shift16 in-place", [&](int i) { k_tiles_seg<1, 6><<<blocks, threads>>>((uint8_t*)a[i], (uint8_t*)a[i], pitch, 3, 62, rows, 16); });
			run("BUF tiles 4rows x 62u shift16", [&](int i) { k_tiles_buf<4, 1><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 16, 62, rows / 4, 16); });
			run("BUF tiles 4rows x 64u aligned", [&](int i) { k_tiles_buf<4, 1><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 15, 64, rows / 4, 0); });
			run("BUF tiles 6rows x 62u shift16", [&](int i) { k_tiles_buf<6, 1><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 16, 62, rows / 6, 16); });
			run("BUF tiles 2rows x 2x62u shift16", [&](int i) { k_tiles_buf<2, 2><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 8, 62, rows / 2, 16); });
			run("BUF tiles 1row x 4x62u shift16", [&](int i) { k_tiles_buf<1, 4><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 4, 62, rows, 16); });
			run("BUF tiles 1row x 4x64u aligned", [&](int i) { k_tiles_buf<1, 4><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 4, 64, rows, 0); });
			run("BUF tiles 1row x 6x62u shift16", [&](int i) { k_tiles_buf<1, 6><<<blocks, threads>>>((uint8_t*)a[i], (uint32_t)bytes, pitch, 3, 62, rows, 16); });
			run("BUF item420 MODE0 (4Y x 62u, UV 2x31)", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item420_buf<0><<<blocks, threads>>>(Yp, 15360u * 4320u, Up, Vp, 7680u * 2160u, 15360, 7680, 0); });
			run("BUF item420 MODE1 (2Y x 2x62u, UV 1x62)", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item420_buf<1><<<blocks, threads>>>(Yp, 15360u * 4320u, Up, Vp, 7680u * 2160u, 15360, 7680, 0); });
			run("BUF plane items 1row x 4x62u, planes in turn", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item_planes<0><<<blocks, threads>>>(Yp, 15360u * 4320u, Up, Vp, 7680u * 2160u, 15360, 7680); });
			run("BUF plane items 1row x 4x62u, interleaved", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item_planes<1><<<blocks, threads>>>(Yp, 15360u * 4320u, Up, Vp, 7680u * 2160u, 15360, 7680); });
			run("item420 (4Y rows + U,V 2x31) in-place", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item420<0><<<blocks, threads>>>(Yp, Up, Vp, 15360, 7680, 16, 1080); });
			run("item420 chroma 1x62 per tile pair in-place", [&](int i) { uint8_t* Yp = (uint8_t*)a[i]; uint8_t* Up = Yp + (size_t)15360 * 4320; uint8_t* Vp = Up + (size_t)7680 * 2160;
				k_item420<1><<<blocks, threads>>>(Yp, Up, Vp, 15360, 7680, 16, 1080); });
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
