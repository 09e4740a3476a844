/* Developer probe (GPU box): does vfgs_add_grain_stripe on the caller's pageable frame buffer slow down the caller's own file I/O
 * into / out of that buffer (the runtime registers the pages of a pageable copy with the GPU)?  fread / stripe / fwrite per frame,
 * each phase timed; PROBE_NOOP=1 skips the library call for comparison.
 * gcc -O2 -Iinclude tools/dev/stripe_io_probe.c -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd -o tools/bin/stripe_io_probe */
#define _POSIX_C_SOURCE 200809L
#include "vfgs_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
	const int w = 7680, h = 4320, nfr = argc > 1 ? atoi(argv[1]) : 6;
	const int noop = getenv("PROBE_NOOP") != NULL;
	signed char pat[64 * 64];
	unsigned char slut[256], plut[256];
	for (int i = 0; i < 64 * 64; i++) pat[i] = (signed char)((i * 37) % 200 - 100);
	for (int i = 0; i < 256; i++) { slut[i] = (unsigned char)(40 + i / 4); plut[i] = 0; }
	vfgs_set_depth(10); vfgs_set_chroma_subsampling(2, 2);
	vfgs_set_luma_pattern(0, pat); vfgs_set_chroma_pattern(0, pat); vfgs_set_chroma_pattern(1, pat);
	for (int c = 0; c < 3; c++) { vfgs_set_scale_lut(c, slut); vfgs_set_pattern_lut(c, plut); }
	vfgs_set_scale_shift(5); vfgs_set_seed(12345u);
	const size_t ny = (size_t)w * h, nc = ny / 4, bytes = (ny + 2 * nc) * 2;
	unsigned short* Y;
	if (posix_memalign((void**)&Y, 128, bytes)) return 1;
	unsigned short *U = Y + ny, *V = U + nc;
	FILE* fi = fopen("/dev/shm/stripe_probe_in.yuv", "wb");
	for (size_t i = 0; i < ny + 2 * nc; i++) Y[i] = (unsigned short)((i * 2654435761u) >> 22);
	for (int f = 0; f < nfr; f++) fwrite(Y, 1, bytes, fi);
	fclose(fi);
	fi = fopen("/dev/shm/stripe_probe_in.yuv", "rb");
	FILE* fo = fopen("/dev/shm/stripe_probe_out.yuv", "wb");
	for (int f = 0; f < nfr; f++)
	{
		const double t0 = now();
		if (fread(Y, 1, bytes, fi) != bytes) return 2;
		const double t1 = now();
		const int sl = getenv("PROBE_STRIPE_LINES") ? atoi(getenv("PROBE_STRIPE_LINES")) : h;     /* the frame in stripes of this many lines */
		if (!noop)
			for (int y = 0; y < h; y += sl)
				vfgs_add_grain_stripe(Y + (size_t)y * w, U + (size_t)(y / 2) * (w / 2), V + (size_t)(y / 2) * (w / 2), (unsigned)y, w, (unsigned)(y + sl <= h ? sl : h - y), w, w / 2);
		const double t2 = now();
		fwrite(Y, 1, bytes, fo);
		const double t3 = now();
		printf("frame %d: fread %.1f ms, %s %.1f ms, fwrite %.1f ms\n", f, (t1 - t0) * 1e3, noop ? "nothing" : "vfgs_add_grain_stripe", (t2 - t1) * 1e3, (t3 - t2) * 1e3);
	}
	fclose(fi); fclose(fo);
	remove("/dev/shm/stripe_probe_in.yuv"); remove("/dev/shm/stripe_probe_out.yuv");
	return 0;
}
