// Developer test (GPU box): are the DPP wavefront shifts (wave_shl:1 / wave_shr:1) reliable on gfx950?
// hipcc -O3 --offload-arch=gfx950 -o /tmp/dpp_test tools/dev/dpp_test.hip && /tmp/dpp_test
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
	for (int it = 0; it < iters; it++)
	{
		u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, it * 1024, 0);
		// some VALU work on the values right before the shifts (as the clip of the grain kernel)
		uint32_t a = __builtin_amdgcn_perm(v.x, v.y, 0x07060100) ^ 0x5a5a5a5a, b = (v.z + 0x00010001u) ^ v.w;
		uint32_t x, y;
		if (MODE == 0)
		{   // shl, shl back to back
			x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x130, 0xf, 0xf, false);
			y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, false);
		}
		else if (MODE == 1)
		{   // shl, store of the sources, shl
			x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x130, 0xf, 0xf, false);
			const u32x2 sd = {a, b};
			__builtin_amdgcn_raw_buffer_store_b64(sd, rs, lane * 8, it * 512, 0);
			__builtin_amdgcn_sched_barrier(0);
			y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, false);
		}
		else
		{   // shr pair
			x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x138, 0xf, 0xf, false);
			y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, false);
		}
		const u32x4 o = {x, y, a, b};
		__builtin_amdgcn_raw_buffer_store_b128(o, ro, lane * 16, it * 1024, 0);
	}
}

int main()
{
	const int iters = 64;
	for (int mode = 0; mode < 3; mode++)
		for (int grid : {1, 4, 64, 4096})
		{
			const size_t waves = (size_t)grid * 4, n = waves * 256 * iters;
			std::vector<uint32_t> h(n), o(n);
			uint32_t s = 12345 + mode;
			for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s | 1u; }
			uint32_t *di, *dout, *dside;
			hipMalloc(&di, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dside, n * 2);
			hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice);
			long bad = 0, badlane[64] = {};
			for (int rep = 0; rep < 20; rep++)
			{
				hipMemset(dout, 0xff, n * 4);
				if (mode == 0) k<0><<<grid, 256>>>(di, dout, dside, iters);
				else if (mode == 1) k<1><<<grid, 256>>>(di, dout, dside, iters);
				else k<2><<<grid, 256>>>(di, dout, dside, iters);
				hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
				for (size_t u = 0; u < n / 4; u++)
				{
					const int lane = u & 63;
					const uint32_t a = o[4 * u + 2], b = o[4 * u + 3];
					uint32_t ea = 0, eb = 0;
					if (mode < 2) { if (lane < 63) { ea = o[4 * (u + 1) + 2]; eb = o[4 * (u + 1) + 3]; } }
					else { if (lane > 0) { ea = o[4 * (u - 1) + 2]; eb = o[4 * (u - 1) + 3]; } }
					if (o[4 * u] != ea) { bad++; badlane[lane]++; }
					if (o[4 * u + 1] != eb) { bad++; badlane[lane]++; }
					(void)a; (void)b;
				}
			}
			printf("mode %d grid %5d: %ld wrong dwords", mode, grid, bad);
			if (bad) { printf("  lanes:"); for (int l = 0; l < 64; l++) if (badlane[l]) printf(" %d:%ld", l, badlane[l]); }
			printf("\n");
			hipFree(di); hipFree(dout); hipFree(dside);
		}
	return 0;
}
