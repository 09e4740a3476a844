/*
 * TEST INFRASTRUCTURE -- hw-programming trace recorder.
 *
 * Implements the ten entry points of the reference hardware-layer interface
 * (/root/reference/src/vfgs_hw.h:51-62) as a *recorder*: every setter call is
 * appended, with its payload, to the file named by $VFGS_TRACE_OUT.  Linked
 * with the reference's own (unmodified, compiled in place) firmware and CLI it
 * captures exactly what the firmware layer programs into the hardware layer
 * for a given cfg file (vfgs_fw.c:517-644, :663-708; vfgs_main.c:750-760).
 * The traces are committed under tests/golden/traces/ so that the GPU box,
 * which has no /root/reference, can still program every implementation
 * (oracle, reference .so, HIP library) identically.
 *
 * Record layout (little endian):
 *   file   : 'V','F','G','T', u32 version(1)
 *   record : u32 op, i32 a, i32 b, u32 nbytes, payload[nbytes]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum {
	OP_LUMA_PATTERN = 1, OP_CHROMA_PATTERN = 2, OP_SCALE_LUT = 3, OP_PATTERN_LUT = 4,
	OP_SEED = 5, OP_SCALE_SHIFT = 6, OP_DEPTH = 7, OP_LEGAL_RANGE = 8, OP_CHROMA_SUBSAMPLING = 9
};

static FILE* out;

static void rec(unsigned op, int a, int b, const void* payload, unsigned n)
{
	if (!out)
	{
		const char* name = getenv("VFGS_TRACE_OUT");
		unsigned version = 1;
		if (!name)
		{
			fprintf(stderr, "trace_shim: VFGS_TRACE_OUT not set\n");
			exit(2);
		}
		out = fopen(name, "wb");
		if (!out)
		{
			perror(name);
			exit(2);
		}
		fwrite("VFGT", 1, 4, out);
		fwrite(&version, 4, 1, out);
	}
	fwrite(&op, 4, 1, out);
	fwrite(&a, 4, 1, out);
	fwrite(&b, 4, 1, out);
	fwrite(&n, 4, 1, out);
	if (n)
		fwrite(payload, 1, n, out);
	fflush(out);
}

/* The firmware always hands over a 64*64 byte buffer (vfgs_fw.c:519, :666);
 * the whole buffer is recorded so a replay reads what the reference would. */
void vfgs_set_luma_pattern(int index, signed char* P)          { rec(OP_LUMA_PATTERN, index, 0, P, 4096); }
void vfgs_set_chroma_pattern(int index, signed char* P)        { rec(OP_CHROMA_PATTERN, index, 0, P, 4096); }
void vfgs_set_scale_lut(int c, unsigned char lut[])            { rec(OP_SCALE_LUT, c, 0, lut, 256); }
void vfgs_set_pattern_lut(int c, unsigned char lut[])          { rec(OP_PATTERN_LUT, c, 0, lut, 256); }
void vfgs_set_seed(unsigned int seed)                          { rec(OP_SEED, (int)seed, 0, NULL, 0); }
void vfgs_set_scale_shift(int shift)                           { rec(OP_SCALE_SHIFT, shift, 0, NULL, 0); }
void vfgs_set_depth(int depth)                                 { rec(OP_DEPTH, depth, 0, NULL, 0); }
void vfgs_set_legal_range(int legal)                           { rec(OP_LEGAL_RANGE, legal, 0, NULL, 0); }
void vfgs_set_chroma_subsampling(int subx, int suby)           { rec(OP_CHROMA_SUBSAMPLING, subx, suby, NULL, 0); }

void vfgs_add_grain_line(void* Y, void* U, void* V, int y, int width)
{
	(void)Y; (void)U; (void)V; (void)y; (void)width;
}
