// Developer probe (round 6): do the d16 LDS loads of gfx950 keep the other half of their destination register?
// (LLVM treats targets with SRAM ECC as "d16 loads write all 32 bits"; the packed 16-bit grain form would like to gather two
// scale bytes into the two halves of one register without a vector instruction.)  Also times the three ways of getting
// two gathered bytes into one register: d16 + d16_hi, u8 + u8 + v_lshl_or, u8 + d16_hi + v_or.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/d16_probe tools/dev/d16_probe.hip && tools/bin/d16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint32_t* out)
{
	__shared__ uint8_t t[256];
	t[threadIdx.x] = (uint8_t)(threadIdx.x * 7 + 1);
	t[threadIdx.x + 64] = (uint8_t)(threadIdx.x * 7 + 1);
	t[threadIdx.x + 128] = (uint8_t)(threadIdx.x * 7 + 1);
	t[threadIdx.x + 192] = (uint8_t)(threadIdx.x * 7 + 1);
	__syncthreads();
	const uint32_t a0 = threadIdx.x, a1 = (threadIdx.x * 5 + 3) & 255;
	uint32_t r = 0xAAAA5555u, q = 0xAAAA5555u;
	// hi after lo
	asm volatile("ds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(a0), "v"(a1) : "memory");
	// hi alone on a register with a known low half
	asm volatile("ds_read_u8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(q) : "v"(a1) : "memory");
	out[threadIdx.x * 4 + 0] = r;
	out[threadIdx.x * 4 + 1] = q;
	out[threadIdx.x * 4 + 2] = t[a0] | (t[a1] << 16);
	// both in flight, no wait in between
	uint32_t z = 0;
	asm volatile("ds_read_u8_d16 %0, %1\n\tds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(z) : "v"(a0), "v"(a1) : "memory");
	out[threadIdx.x * 4 + 3] = z;
}

template <int MODE>
__global__ void rate(uint32_t* out, int iters)
{
	__shared__ uint8_t t[256];
	for (int i = threadIdx.x; i < 256; i += blockDim.x) t[i] = (uint8_t)(i * 7 + 1);
	__syncthreads();
	uint32_t acc = 0, a = threadIdx.x & 255;
	for (int it = 0; it < iters; it++)
	{
		uint32_t s[8];
#pragma unroll
		for (int k = 0; k < 8; k++)
		{
			const uint32_t a0 = (a + 13 * k) & 255, a1 = (a * 3 + 7 * k) & 255;
			if (MODE == 0) asm volatile("ds_read_u8_d16 %0, %1\n\tds_read_u8_d16_hi %0, %2" : "=&v"(s[k]) : "v"(a0), "v"(a1));
			else if (MODE == 1) s[k] = t[a0] | ((uint32_t)t[a1] << 16);
			else
			{
				uint32_t hi;
				asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(hi) : "v"(a1));
				s[k] = t[a0];
				asm volatile("s_waitcnt lgkmcnt(0)\n\tv_or_b32 %0, %0, %1" : "+v"(s[k]) : "v"(hi));
			}
		}
		if (MODE == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
#pragma unroll
		for (int k = 0; k < 8; k++) acc += s[k];
		a = (a + acc) & 255;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
	uint32_t* d;
	hipMalloc(&d, 1 << 24);
	hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
	uint32_t h[256];
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	int keep = 0, zero = 0, other = 0, pair_ok = 0, inflight_ok = 0;
	for (int i = 0; i < 64; i++)
	{
		const uint32_t want = h[4 * i + 2];
		if (h[4 * i + 0] == want) pair_ok++;
		if (h[4 * i + 3] == want) inflight_ok++;
		const uint32_t q = h[4 * i + 1];
		if ((q & 0xffff) == 0x5555) keep++; else if ((q & 0xffff) == 0) zero++; else other++;
	}
	printf("d16 then d16_hi (waited)   : %d/64 lanes hold both bytes\n", pair_ok);
	printf("d16 then d16_hi (in flight): %d/64 lanes hold both bytes\n", inflight_ok);
	printf("d16_hi alone: low half kept %d, zeroed %d, other %d   (lane 0: %08x)\n", keep, zero, other, h[1]);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 4000;
	for (int mode = 0; mode < 3; mode++)
	{
		for (int rep = 0; rep < 2; rep++)
		{
			hipEventRecord(e0);
			if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
			if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
			if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256 * 8), dim3(256), 0, 0, d, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("mode %d (%s): %.3f ms\n", mode, mode == 0 ? "d16 + d16_hi" : mode == 1 ? "u8 + u8 + lshl_or (compiler)" : "u8 + d16_hi + or", ms);
		}
	}
	return 0;
}
