/* Developer probe (GPU box): a plain C caller -- no Python, no torch -- of the reference's frame loop (vfgs_main.c:664-682).
 * What do the first calls of a process cost, what does the first (line by line) walk cost, what the later ones?
 * gcc -O2 -Iinclude tools/dev/line_first_walk_probe.c -Lversatilefilmgrain_amd -lvfgs_hip -Wl,-rpath,$PWD/versatilefilmgrain_amd -o tools/bin/line_first_walk_probe */
#define _POSIX_C_SOURCE 200809L
#include "vfgs_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
	const int w = argc > 1 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1080;
	const double t00 = now();
	signed char pat[64 * 64];
	unsigned char slut[256], plut[256];
	for (int i = 0; i < 64 * 64; i++) pat[i] = (signed char)((i * 37) % 200 - 100);
	for (int i = 0; i < 256; i++) { slut[i] = (unsigned char)(40 + i / 4); plut[i] = 0; }
	vfgs_set_depth(10);
	vfgs_set_chroma_subsampling(2, 2);
	vfgs_set_luma_pattern(0, pat);
	vfgs_set_chroma_pattern(0, pat);
	vfgs_set_chroma_pattern(1, pat);
	for (int c = 0; c < 3; c++) { vfgs_set_scale_lut(c, slut); vfgs_set_pattern_lut(c, plut); }
	vfgs_set_scale_shift(5);
	vfgs_set_seed(12345u);
	unsigned short *Y, *U, *V;
	const size_t ny = (size_t)w * ((h + 15) & ~15), nc = ny / 4;
	if (posix_memalign((void**)&Y, 128, (ny + 2 * nc) * 2)) return 1;
	U = Y + ny; V = U + nc;
	for (size_t i = 0; i < ny + 2 * nc; i++) Y[i] = (unsigned short)((i * 2654435761u) >> 22);
	printf("setters + allocation: %.1f ms\n", (now() - t00) * 1e3);
	for (int walk = 0; walk < 4; walk++)
	{
		const double t0 = now();
		double first3[3] = {0, 0, 0};
		unsigned short *py = Y, *pu = U, *pv = V;
		for (int y = 0; y < h; y++)
		{
			const double a = now();
			vfgs_add_grain_line(py, pu, pv, y, w);
			if (y < 3) first3[y] = now() - a;
			py += w;
			if (y & 1) { pu += w / 2; pv += w / 2; }
		}
		const double dt = now() - t0;
		printf("%dx%d walk %d: %8.1f ms = %6.1f us per line (first three calls: %.1f / %.3f / %.3f ms)\n", w, h, walk, dt * 1e3, dt / h * 1e6,
		       first3[0] * 1e3, first3[1] * 1e3, first3[2] * 1e3);
	}
	const double t1 = now();
	vfgs_hip_shutdown();
	printf("shutdown: %.1f ms; whole process so far: %.1f ms\n", (now() - t1) * 1e3, (now() - t00) * 1e3);
	return 0;
}
