// Developer microbenchmark: issue rate of the VALU instructions the grain kernel uses, per
// SIMD, at 1/2/4/8 waves per SIMD (gfx950).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define OPS_PER_ITER 64

#define DEFK(NAME, ASM)                                                                              \
__global__ void k_##NAME(uint32_t* out, int iters)                                                   \
{                                                                                                    \
	uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 0x01020304u, d = 5;                        \
	uint32_t r0 = a, r1 = b, r2 = a ^ b, r3 = a + b;                                                  \
	for (int i = 0; i < iters; i++)                                                                   \
	{                                                                                                \
		asm volatile(REP16(ASM) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc", "s10", "s11", "s12", "s13"); \
	}                                                                                                \
	out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                                   \
}

// each ASM string = 4 independent instructions (r0..r3 chains) -> REP16 = 64 instructions
DEFK(add,     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5\n")
DEFK(and_,    "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %5\n")
DEFK(perm,    "v_perm_b32 %0, %0, %4, %6\n v_perm_b32 %1, %1, %5, %6\n v_perm_b32 %2, %2, %4, %6\n v_perm_b32 %3, %3, %5, %6\n")
DEFK(mad24,   "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %5, %4\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %5, %4\n")
DEFK(mul24,   "v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %5\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %5\n")
DEFK(mul24sdwa, "v_mul_i32_i24_sdwa %0, sext(%0), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_i32_i24_sdwa %1, sext(%1), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_i32_i24_sdwa %2, sext(%2), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_i32_i24_sdwa %3, sext(%3), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n")
DEFK(pkadd,   "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %5\n")
DEFK(pkmax,   "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %5\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %5\n")
DEFK(pkmad,   "v_pk_mad_i16 %0, %0, %4, %5\n v_pk_mad_i16 %1, %1, %5, %4\n v_pk_mad_i16 %2, %2, %4, %5\n v_pk_mad_i16 %3, %3, %5, %4\n")
DEFK(pkmul,   "v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %5\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %5\n")
DEFK(pkashr,  "v_pk_ashrrev_i16 %0, %7, %0\n v_pk_ashrrev_i16 %1, %7, %1\n v_pk_ashrrev_i16 %2, %7, %2\n v_pk_ashrrev_i16 %3, %7, %3\n")
DEFK(ashr,    "v_ashrrev_i32 %0, %7, %0\n v_ashrrev_i32 %1, %7, %1\n v_ashrrev_i32 %2, %7, %2\n v_ashrrev_i32 %3, %7, %3\n")
DEFK(bfe,     "v_bfe_i32 %0, %0, %7, 8\n v_bfe_i32 %1, %1, %7, 8\n v_bfe_i32 %2, %2, %7, 8\n v_bfe_i32 %3, %3, %7, 8\n")
DEFK(add3,    "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %5, %4\n")
DEFK(lshladd, "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %5\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %5\n")
DEFK(mullo,   "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %5\n")
DEFK(fma,     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %5, %4\n")
DEFK(pkfma,   "v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %5, %4\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %5, %4\n")
DEFK(cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n")
DEFK(cndmask_s, "v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %5, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %5, s[10:11]\n")
DEFK(cndmask_i, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n")
DEFK(bfi,     "v_bfi_b32 %0, %6, %4, %0\n v_bfi_b32 %1, %6, %5, %1\n v_bfi_b32 %2, %6, %4, %2\n v_bfi_b32 %3, %6, %5, %3\n")
DEFK(cmp,     "v_cmp_gt_i32 vcc, %0, %4\n v_cmp_gt_i32 vcc, %1, %5\n v_cmp_gt_i32 vcc, %2, %4\n v_cmp_gt_i32 vcc, %3, %5\n")
DEFK(cmp_s,   "v_cmp_gt_i32_e64 s[10:11], %0, %4\n v_cmp_gt_i32_e64 s[12:13], %1, %5\n v_cmp_gt_i32_e64 s[10:11], %2, %4\n v_cmp_gt_i32_e64 s[12:13], %3, %5\n")
DEFK(cmpcnd,  "v_cmp_gt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_gt_i32 vcc, %1, %5\n v_cndmask_b32 %1, %1, %4, vcc\n")
DEFK(minmax,  "v_max_i32 %0, %0, %4\n v_min_i32 %1, %1, %5\n v_max_i32 %2, %2, %4\n v_min_i32 %3, %3, %5\n")
DEFK(dpp,     "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
DEFK(med3,    "v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %5, %4\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %5, %4\n")
DEFK(mad16,   "v_mad_i32_i16 %0, %0, %4, %5\n v_mad_i32_i16 %1, %1, %5, %4 op_sel:[1,0,0,0]\n v_mad_i32_i16 %2, %2, %4, %5\n v_mad_i32_i16 %3, %3, %5, %4 op_sel:[1,0,0,0]\n")
DEFK(bfeu,    "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 16, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 16, 8\n")
DEFK(lshr,    "v_lshrrev_b32 %0, 24, %0\n v_lshrrev_b32 %1, 24, %1\n v_lshrrev_b32 %2, 24, %2\n v_lshrrev_b32 %3, 24, %3\n")
DEFK(dot2,    "v_dot2_i32_i16 %0, %4, %5, %0\n v_dot2_i32_i16 %1, %5, %4, %1\n v_dot2_i32_i16 %2, %4, %5, %2\n v_dot2_i32_i16 %3, %5, %4, %3\n")
DEFK(dot4,    "v_dot4_i32_i8 %0, %4, %5, %0\n v_dot4_i32_i8 %1, %5, %4, %1\n v_dot4_i32_i8 %2, %4, %5, %2\n v_dot4_i32_i8 %3, %5, %4, %3\n")

typedef void (*kfn)(uint32_t*, int);
struct K { const char* name; kfn fn; };
#define E(NAME) { #NAME, k_##NAME }

int main()
{
	K ks[] = { E(add), E(and_), E(perm), E(mad24), E(mul24), E(mul24sdwa), E(pkadd), E(pkmax), E(pkmad), E(pkmul), E(pkashr), E(ashr), E(bfe),
	           E(add3), E(lshladd), E(mullo), E(fma), E(pkfma), E(cndmask), E(cndmask_s), E(cndmask_i), E(bfi), E(cmp), E(cmp_s), E(cmpcnd), E(minmax), E(dpp), E(med3), E(mad16), E(bfeu), E(lshr), E(dot2), E(dot4) };
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	uint32_t* out;
	hipMalloc(&out, (size_t)cus * 2048 * 4 * 2);
	const int iters = 2000;
	printf("%-10s", "op");
	for (int wps : {1, 2, 4, 8}) printf("  w/SIMD=%d cyc/instr", wps);
	printf("   (cycles per wave-instruction per SIMD at %.2f GHz nominal)\n", prop.clockRate / 1e6);
	for (auto& k : ks)
	{
		printf("%-10s", k.name);
		for (int wps : {1, 2, 4, 8})
		{
			const int threads = 256 * (wps > 4 ? 4 : wps);       // wps waves on each of 4 SIMDs
			const int blocks = cus * (wps > 4 ? wps / 4 : 1);
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, 10);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			const double instr_per_simd = (double)iters * OPS_PER_ITER * wps;
			const double cyc = ms * 1e-3 * prop.clockRate * 1e3 / instr_per_simd;
			printf("  %18.2f", cyc);
		}
		printf("\n");
	}
	return 0;
}
