// Developer test (GPU box): a wave_shl:1 DPP move right next to a buffer store that reads the same registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ side, int iters)
{
	const int lane = threadIdx.x & 63;
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(in + wave * 256 * (size_t)iters), 0, 1024 * iters, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + wave * 256 * (size_t)iters), 0, 1024 * iters, 0x00020000);
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(side + wave * 128 * (size_t)iters), 0, 512 * iters, 0x00020000);
	const uint32_t soff = lane == 0 ? 0u : 0x80000000u;     // as the grain kernel's "pre" store: one lane in range
	for (int it = 0; it < iters; it++)
	{
		u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, it * 1024, 2);
		u32x2 ab = {v.x ^ 0x5a5a5a5a, (v.z + 0x00010001u) ^ v.w};
		uint32_t x = 0x0222, y = 0x0222;
		const int so = it * 512;
		if (MODE == 0)
			asm volatile("v_pk_min_i16 %2, %2, %5\n\tv_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
			             "buffer_store_dwordx2 %2, %3, %4, %6 offen nt\n\t"
			             "v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf"
			             : "+v"(x), "+v"(y), "+v"(ab) : "v"(soff), "s"(rs), "s"(0x7fff7fff), "s"(so) : "memory");
		else if (MODE == 1)
			asm volatile("v_pk_min_i16 %2, %2, %5\n\ts_nop 1\n\tbuffer_store_dwordx2 %2, %3, %4, %6 offen nt\n\t"
			             "v_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
			             "v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf"
			             : "+v"(x), "+v"(y), "+v"(ab) : "v"(soff), "s"(rs), "s"(0x7fff7fff), "s"(so) : "memory");
		else if (MODE == 2)
			asm volatile("v_pk_min_i16 %2, %2, %5\n\ts_nop 1\n\tv_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
			             "s_nop 3\n\tbuffer_store_dwordx2 %2, %3, %4, %6 offen nt\n\ts_nop 3\n\t"
			             "v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf"
			             : "+v"(x), "+v"(y), "+v"(ab) : "v"(soff), "s"(rs), "s"(0x7fff7fff), "s"(so) : "memory");
		else
			asm volatile("v_pk_min_i16 %2, %2, %5\n\ts_nop 1\n\tv_mov_b32_dpp %0, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
			             "v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
			             "buffer_store_dwordx2 %2, %3, %4, %6 offen nt"
			             : "+v"(x), "+v"(y), "+v"(ab) : "v"(soff), "s"(rs), "s"(0x7fff7fff), "s"(so) : "memory");
		const u32x4 o = {x, y, ab.x, ab.y};
		__builtin_amdgcn_raw_buffer_store_b128(o, ro, lane * 16, it * 1024, 2);
	}
}

int main()
{
	const int iters = 4;
	for (int mode = 0; mode < 4; mode++)
		for (int grid : {1, 16, 300, 4096})
		{
			const size_t waves = (size_t)grid * 4, n = waves * 256 * iters;
			std::vector<uint32_t> h(n), o(n);
			uint32_t s = 12345 + mode;
			for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s | 1u; }
			uint32_t *di, *dout, *dside;
			hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dside, n * 2);
			hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
			long bad = 0, badlane[64] = {}, badx = 0;
			for (int rep = 0; rep < 50; rep++)
			{
				hipMemset(dout, 0xff, n * 4);
				if (mode == 0) k<0><<<grid, 256>>>(di, dout, dside, iters);
				else if (mode == 1) k<1><<<grid, 256>>>(di, dout, dside, iters);
				else if (mode == 2) k<2><<<grid, 256>>>(di, dout, dside, iters);
				else k<3><<<grid, 256>>>(di, dout, dside, iters);
				hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
				for (size_t u = 0; u < n / 4; u++)
				{
					const int lane = u & 63;
					const uint32_t ea = lane < 63 ? o[4 * (u + 1) + 2] : 0x0222;
					if (o[4 * u] != ea) { bad++; badx++; badlane[lane]++; }
					if (o[4 * u + 1] != ea) { bad++; badlane[lane]++; }
				}
			}
			printf("mode %d grid %5d: %ld wrong dwords (%ld in the first move)", mode, grid, bad, badx);
			if (bad) { printf("  lanes:"); for (int l = 0; l < 64; l++) if (badlane[l]) printf(" %d:%ld", l, badlane[l]); }
			printf("\n");
			hipFree(di); hipFree(dout); hipFree(dside);
		}
	return 0;
}
