/* Developer probe: LD_PRELOAD shim around vfgs_add_grain_line for an unchanged binary (the reference CLI): time spent inside the
 * library per frame walk, and between walks (the program's own file I/O).
 * gcc -O2 -shared -fPIC tools/dev/line_time_shim.c -ldl -o tools/bin/line_time_shim.so */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void (*real)(void*, void*, void*, int, int);
static double t_proc0, t_in, t_walk0, t_prev_end, t_first;
static int walk, last_y = -1;
static double t_head[6], t_max; static int y_max;      /* the first calls of a walk one by one, and its slowest call */

void vfgs_add_grain_line(void* Y, void* U, void* V, int y, int width)
{
	if (!real) { real = (void (*)(void*, void*, void*, int, int))dlsym(RTLD_NEXT, "vfgs_add_grain_line"); t_proc0 = now(); t_prev_end = t_proc0; }
	const double a = now();
	if (y == 0)
	{
		if (last_y >= 0)
			fprintf(stderr, "shim: walk %d: %d lines, %.1f ms inside the library (first call %.1f ms), %.1f ms outside before it; calls 0..5: %.2f %.2f %.2f %.2f %.2f %.2f ms, slowest: line %d %.2f ms\n", walk++, last_y + 1, t_in * 1e3, t_first * 1e3,
			        (t_walk0 - t_prev_end) * 1e3, t_head[0] * 1e3, t_head[1] * 1e3, t_head[2] * 1e3, t_head[3] * 1e3, t_head[4] * 1e3, t_head[5] * 1e3, y_max, t_max * 1e3), t_prev_end = a;
		t_in = 0; t_walk0 = a; t_max = 0;
	}
	real(Y, U, V, y, width);
	const double d = now() - a;
	if (y == 0) t_first = d;
	if (y >= 0 && y < 6) t_head[y] = d;
	if (d > t_max) { t_max = d; y_max = y; }
	t_in += d;
	last_y = y;
}

__attribute__((destructor)) static void fin(void)
{
	if (last_y >= 0) fprintf(stderr, "shim: walk %d: %d lines, %.1f ms inside the library (first call %.1f ms); calls 0..5: %.2f %.2f %.2f %.2f %.2f %.2f ms, slowest: line %d %.2f ms; %.1f ms since the first call\n", walk, last_y + 1, t_in * 1e3, t_first * 1e3,
	                         t_head[0] * 1e3, t_head[1] * 1e3, t_head[2] * 1e3, t_head[3] * 1e3, t_head[4] * 1e3, t_head[5] * 1e3, y_max, t_max * 1e3, (now() - t_proc0) * 1e3);
}
