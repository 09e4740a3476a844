/*
 * TEST INFRASTRUCTURE -- CPU oracle for the VFGS hardware layer (see
 * vfgs_oracle.h for what may load this and how it is pinned to the reference).
 *
 * Written from scratch; every function cites the lines of
 * /root/reference/src/vfgs_hw.c whose *behaviour* it restates.  Where the
 * reference walks a line block by block through a static software pipeline
 * (vfgs_hw.c:140-284), this file computes a whole line in three flat passes
 * (raw grain -> edge filter -> scale/clip), which is the same arithmetic in a
 * different order and is the bridge to the closed form below.
 */
#include "vfgs_oracle.h"

#include <stdlib.h>
#include <string.h>

#define SLOTS 8      /* usable pattern slots, vfgs_hw.h:49; the 9th slot of vfgs_hw.c:49 is never read (PATTERN_INTERPOLATION 0) */
#define PAT   64     /* pattern bank edge */

struct vfgs_oracle {
	int8_t   bank[2][SLOTS + 1][PAT][PAT]; /* [0]=luma, [1]=chroma; slot 8 stays zero  vfgs_hw.c:49 */
	uint8_t  scale_lut[3][256];         /*                               vfgs_hw.c:50 */
	uint8_t  slot_lut[3][256];          /* raw pLUT bytes (slot = v>>4)  vfgs_hw.c:51 */
	uint32_t rnd, rnd_up, line_rnd, line_rnd_up; /*                      vfgs_hw.c:52-55 */
	int      scale_shift;               /*                               vfgs_hw.c:56 */
	int      bs;                        /* depth - 8                     vfgs_hw.c:57 */
	int      lo[2], hi[2];              /* [0]=luma [1]=chroma clip, 8-bit units  vfgs_hw.c:58-61 */
	int      csubx, csuby;              /*                               vfgs_hw.c:62-63 */
	/* scratch for one component line */
	int      cap;
	int32_t* grain;
	uint8_t* gain;
};

/* ---- small helpers -------------------------------------------------------- */

static int rshift_round(int v, int s)            /* vfgs_hw.c:43 */
{
	return (v + (1 << (s - 1))) >> s;
}

uint32_t vfgs_oracle_lfsr_step(uint32_t reg)     /* vfgs_hw.c:74-79 */
{
	uint32_t fb = ((reg >> 1) ^ (reg >> 29)) & 1u;
	return (reg >> 1) | (fb << 31);
}

typedef struct { int sign, ox, oy; } block_offs;

/* vfgs_hw.c:99-138: which bits of the register feed sign / x / y per component. */
static block_offs block_offsets(uint32_t v, int c, int csubx, int csuby)
{
	block_offs r;
	uint32_t fx, fy;
	int sbit, mx, my;

	if (c == 0)      { sbit = 31; fx = v & 0x3ff;         fy = (v >> 14) & 0x3ff;                          mx = 4;         my = 4; }
	else if (c == 1) { sbit = 2;  fx = (v >> 10) & 0x3ff; fy = ((v >> 24) & 0xff) | ((v << 8) & 0x300);     mx = 4 / csubx; my = 4 / csuby; }
	else             { sbit = 15; fx = (v >> 20) & 0x3ff; fy = (v >> 4) & 0x3ff;                           mx = 4 / csubx; my = 4 / csuby; }

	r.sign = ((v >> sbit) & 1) ? -1 : 1;
	r.ox = (int)((fx * 13) >> 10) * mx;
	r.oy = (int)((fy * 12) >> 10) * my;
	return r;
}

/* vfgs_hw.c:173-188: blend weights (current, upper) for luma line y. */
static void overlap_weights(int y, int suby, int* w_cur, int* w_up)
{
	int j = y & 15;
	*w_cur = *w_up = 0;
	if (y > 15 && j == 0) { *w_cur = suby > 1 ? 20 : 12; *w_up = suby > 1 ? 20 : 24; }
	else if (y > 15 && j == 1) { *w_cur = 24; *w_up = 12; }
}

static int read_sample(const void* p, int bs, int i)
{
	return bs ? ((const uint16_t*)p)[i] : ((const uint8_t*)p)[i];
}

static void write_sample(void* p, int bs, int i, int v)
{
	if (bs) ((uint16_t*)p)[i] = (uint16_t)v; else ((uint8_t*)p)[i] = (uint8_t)v;
}

/* Raw (signed, overlap-blended, not yet edge-filtered) grain of one sample:
 * vfgs_hw.c:211-229 for sample i of a block with offsets `cur` (and `up`). */
static int raw_grain(const vfgs_oracle* o, int c, int intensity, int i, int row, int row_up,
                     block_offs cur, block_offs up, int w_cur, int w_up)
{
	int slot = o->slot_lut[c][intensity] >> 4;
	const int8_t (*P)[PAT];
	if (slot > SLOTS)   /* the reference would index past its bank here (undefined); the firmware never writes > 7 */
		abort();
	P = o->bank[c ? 1 : 0][slot];
	int g = P[cur.oy + row][cur.ox + i] * cur.sign;
	if (w_cur)
		g = rshift_round(g * w_cur + P[up.oy + row_up][up.ox + i] * w_up * up.sign, 5);
	return g;
}

/* Edge filter across the boundary of two neighbouring blocks: vfgs_hw.c:250-259.
 * g points at the first sample of the right-hand block. */
static void filter_block_edge(int32_t* g)
{
	int l1 = g[-2], l0 = g[-1], r0 = g[0], r1 = g[1];
	g[-1] = (int16_t)rshift_round(l1 + 3 * l0 + r0, 2);
	g[0]  = (int16_t)rshift_round(l0 + 3 * r0 + r1, 2);
}

/* Scale, add, clip one sample: vfgs_hw.c:263-267. */
static int blend_sample(const vfgs_oracle* o, int c, int in, int grain, int gain)
{
	int g = rshift_round(gain * (int16_t)grain, o->scale_shift);
	int lo = o->lo[c ? 1 : 0] << o->bs, hi = o->hi[c ? 1 : 0] << o->bs;
	int v = in + g;
	if (v > hi) v = hi;
	if (v < lo) v = lo;
	return v;
}

static void need_scratch(vfgs_oracle* o, int n)
{
	if (n > o->cap)
	{
		o->grain = (int32_t*)realloc(o->grain, sizeof(int32_t) * n);
		o->gain = (uint8_t*)realloc(o->gain, n);
		o->cap = n;
	}
}

/* ---- lifecycle and setters ------------------------------------------------ */

vfgs_oracle* vfgs_oracle_create(void)
{
	vfgs_oracle* o = (vfgs_oracle*)calloc(1, sizeof(*o));
	o->rnd = o->rnd_up = o->line_rnd = o->line_rnd_up = 0xdeadbeefu; /* vfgs_hw.c:52-55 */
	o->scale_shift = 5 + 6;                                           /* vfgs_hw.c:56 */
	o->bs = 0;
	o->lo[0] = o->lo[1] = 0;
	o->hi[0] = o->hi[1] = 255;
	o->csubx = o->csuby = 2;
	return o;
}

void vfgs_oracle_destroy(vfgs_oracle* o)
{
	if (!o) return;
	free(o->grain);
	free(o->gain);
	free(o);
}

void vfgs_oracle_set_luma_pattern(vfgs_oracle* o, int index, const int8_t* P)   /* vfgs_hw.c:314-318 */
{
	memcpy(o->bank[0][index], P, PAT * PAT);
}

void vfgs_oracle_set_chroma_pattern(vfgs_oracle* o, int index, const int8_t* P) /* vfgs_hw.c:320-325 */
{
	/* rows and source pitch both come from csuby, row length from csubx */
	int rows = PAT / o->csuby, pitch = PAT / o->csuby, len = PAT / o->csubx;
	for (int r = 0; r < rows; r++)
		memcpy(o->bank[1][index][r], P + pitch * r, len);
}

void vfgs_oracle_set_scale_lut(vfgs_oracle* o, int c, const uint8_t* lut)       /* vfgs_hw.c:327-331 */
{
	memcpy(o->scale_lut[c], lut, 256);
}

void vfgs_oracle_set_pattern_lut(vfgs_oracle* o, int c, const uint8_t* lut)     /* vfgs_hw.c:333-337 */
{
	memcpy(o->slot_lut[c], lut, 256);
}

void vfgs_oracle_set_seed(vfgs_oracle* o, uint32_t seed)                        /* vfgs_hw.c:339-344 */
{
	o->rnd = o->rnd_up = o->line_rnd = o->line_rnd_up = seed << 1;
}

void vfgs_oracle_set_scale_shift(vfgs_oracle* o, int shift)                     /* vfgs_hw.c:346-350 */
{
	o->scale_shift = shift + 6 - o->bs;
}

void vfgs_oracle_set_depth(vfgs_oracle* o, int depth)                           /* vfgs_hw.c:352-362 */
{
	int nbs = depth - 8;
	o->scale_shift += o->bs - nbs;   /* -2 going 8->10, +2 going 10->8, 0 otherwise */
	o->bs = nbs;
}

void vfgs_oracle_set_legal_range(vfgs_oracle* o, int legal)                     /* vfgs_hw.c:364-380 */
{
	o->lo[0] = o->lo[1] = legal ? 16 : 0;
	o->hi[0] = legal ? 235 : 255;
	o->hi[1] = legal ? 240 : 255;
}

void vfgs_oracle_set_chroma_subsampling(vfgs_oracle* o, int subx, int suby)     /* vfgs_hw.c:382-388 */
{
	o->csubx = subx;
	o->csuby = suby;
}

void vfgs_oracle_get_seed_state(const vfgs_oracle* o, uint32_t out[4])
{
	out[0] = o->rnd; out[1] = o->rnd_up; out[2] = o->line_rnd; out[3] = o->line_rnd_up;
}

/* ---- form 1: line-API state machine --------------------------------------- */

/* One component of one line.  The per-block registers are re-derived from the
 * two line-start registers by stepping once per block (vfgs_hw.c:301-311). */
static void component_line(vfgs_oracle* o, void* I, int c, int y, int width, uint32_t r_cur, uint32_t r_up)
{
	int subx = c ? o->csubx : 1, suby = c ? o->csuby : 1;
	int n = 16 / subx;                    /* samples per block */
	int nblk = (width + 15) / 16;         /* loop of vfgs_hw.c:301 */
	int total = nblk * n;
	int w_cur, w_up, j = y & 15;

	if ((y & 1) && suby > 1)              /* vfgs_hw.c:164-165 */
		return;

	overlap_weights(y, suby, &w_cur, &w_up);
	need_scratch(o, total);

	/* pass 1: raw grain + gain for every sample of the line (vfgs_hw.c:190-240) */
	for (int b = 0; b < nblk; b++)
	{
		block_offs cur = block_offsets(r_cur, c, o->csubx, o->csuby);
		block_offs up = block_offsets(r_up, c, o->csubx, o->csuby);
		for (int i = 0; i < n; i++)
		{
			int x = b * n + i;
			int intensity = (read_sample(I, o->bs, x) >> o->bs) & 0xff; /* uint8 intensity, vfgs_hw.c:157,211 */
			o->grain[x] = (int16_t)raw_grain(o, c, intensity, i, j / suby, (16 + j) / suby, cur, up, w_cur, w_up);
			o->gain[x] = o->scale_lut[c][intensity];
		}
		r_cur = vfgs_oracle_lfsr_step(r_cur);
		r_up = vfgs_oracle_lfsr_step(r_up);
	}

	/* pass 2: every interior block boundary (vfgs_hw.c:245-259; first block has
	 * no left filter, last block no right filter) */
	for (int b = 1; b < nblk; b++)
		filter_block_edge(o->grain + b * n);

	/* pass 3: scale, add, clip, store (vfgs_hw.c:260-268 and the flush :278-282) */
	for (int x = 0; x < total; x++)
		write_sample(I, o->bs, x, blend_sample(o, c, read_sample(I, o->bs, x), o->grain[x], o->gain[x]));
}

void vfgs_oracle_add_grain_line(vfgs_oracle* o, void* Y, void* U, void* V, int y, int width) /* vfgs_hw.c:288-312 */
{
	int nblk = (width + 15) / 16;

	if (y && (y & 15) == 0)               /* vfgs_hw.c:291-296 */
	{
		o->line_rnd_up = o->line_rnd;
		o->line_rnd = o->rnd;
	}
	o->rnd_up = o->line_rnd_up;           /* vfgs_hw.c:297-298 */
	o->rnd = o->line_rnd;

	component_line(o, Y, 0, y, width, o->rnd, o->rnd_up);
	component_line(o, U, 1, y, width, o->rnd, o->rnd_up);
	component_line(o, V, 2, y, width, o->rnd, o->rnd_up);

	for (int b = 0; b < nblk; b++)        /* vfgs_hw.c:309-310 */
	{
		o->rnd = vfgs_oracle_lfsr_step(o->rnd);
		o->rnd_up = vfgs_oracle_lfsr_step(o->rnd_up);
	}
}

void vfgs_oracle_add_grain_frame(vfgs_oracle* o, void* Y, void* U, void* V,
                                 int width, int height, int stride, int cstride) /* vfgs_main.c:664-682 */
{
	int sz = o->bs ? 2 : 1;
	uint8_t *py = (uint8_t*)Y, *pu = (uint8_t*)U, *pv = (uint8_t*)V;
	for (int y = 0; y < height; y++)
	{
		vfgs_oracle_add_grain_line(o, py, pu, pv, y, width);
		py += (size_t)stride * sz;
		if ((y & 1) || o->csuby == 1)
		{
			pu += (size_t)cstride * sz;
			pv += (size_t)cstride * sz;
		}
	}
}

/* ---- LFSR as a random-access bit stream ----------------------------------- */

/* Bit m of the stream is t[m]; t[0..31] are the register bits and
 * t[m] = t[m-31] ^ t[m-3] for m >= 32 (vfgs_hw.c:74-79 shifts right and feeds
 * bit1^bit29 into bit 31).  For the sequence u[i] = t[i+1] the characteristic
 * polynomial can be raised to the 32nd power over GF(2), giving the same
 * recurrence at 32-bit word granularity, words[n] = words[n-31] ^ words[n-3],
 * valid once every referenced bit index is >= 1, i.e. from word 32 on. */
void vfgs_oracle_lfsr_stream(uint32_t reg, uint32_t* words, uint64_t nwords)
{
	uint64_t n;
	for (n = 0; n < nwords && n < 32; n++)
	{
		words[n] = reg;
		for (int k = 0; k < 32; k++)
			reg = vfgs_oracle_lfsr_step(reg);
	}
	for (; n < nwords; n++)
		words[n] = words[n - 31] ^ words[n - 3];
}

static uint32_t stream_window(const uint32_t* words, uint64_t bit)
{
	uint64_t w = bit >> 5;
	unsigned s = (unsigned)(bit & 31);
	return s ? (words[w] >> s) | (words[w + 1] << (32 - s)) : words[w];
}

/* ---- form 2: closed form, random access ----------------------------------- */

typedef struct {
	const vfgs_oracle* o;
	const uint32_t* cur_stream;   /* stream whose window 0 is line_rnd at entry        */
	const uint32_t* up0_stream;   /* stream whose window 0 is line_rnd_up at entry     */
	int nblk;
} cf_ctx;

/* Raw grain of component-sample (xc, yc), straight from the frame buffer. */
static int cf_raw(const cf_ctx* k, const void* plane, int pitch, int c, int xc, int yc)
{
	const vfgs_oracle* o = k->o;
	int subx = c ? o->csubx : 1, suby = c ? o->csuby : 1;
	int n = 16 / subx;
	int y = yc * suby, r = y >> 4, j = y & 15;
	int b = xc / n, i = xc % n;
	int w_cur, w_up;
	uint32_t v_cur = stream_window(k->cur_stream, (uint64_t)r * k->nblk + b);
	uint32_t v_up = r ? stream_window(k->cur_stream, (uint64_t)(r - 1) * k->nblk + b)
	                  : stream_window(k->up0_stream, (uint64_t)b);
	const uint8_t* row = (const uint8_t*)plane + (size_t)yc * pitch * (o->bs ? 2 : 1);
	int intensity = (read_sample(row, o->bs, xc) >> o->bs) & 0xff;

	overlap_weights(y, suby, &w_cur, &w_up);
	return (int16_t)raw_grain(o, c, intensity, i, j / suby, (16 + j) / suby,
	                          block_offsets(v_cur, c, o->csubx, o->csuby),
	                          block_offsets(v_up, c, o->csubx, o->csuby), w_cur, w_up);
}

static void cf_plane(const cf_ctx* k, const void* src, void* dst, int pitch, int c, int height)
{
	const vfgs_oracle* o = k->o;
	int subx = c ? o->csubx : 1, suby = c ? o->csuby : 1;
	int n = 16 / subx, total = k->nblk * n, rows = (height + suby - 1) / suby;

	for (int yc = 0; yc < rows; yc++)
	{
		const uint8_t* in = (const uint8_t*)src + (size_t)yc * pitch * (o->bs ? 2 : 1);
		uint8_t* out = (uint8_t*)dst + (size_t)yc * pitch * (o->bs ? 2 : 1);
		for (int xc = 0; xc < total; xc++)
		{
			int b = xc / n, i = xc % n;
			int g = cf_raw(k, src, pitch, c, xc, yc);
			int sample = read_sample(in, o->bs, xc);
			/* 3-tap edge filter on *raw* neighbours, only next to an interior boundary */
			if ((i == 0 && b > 0) || (i == n - 1 && b < k->nblk - 1))
				g = (int16_t)rshift_round(cf_raw(k, src, pitch, c, xc - 1, yc) + 3 * g + cf_raw(k, src, pitch, c, xc + 1, yc), 2);
			write_sample(out, o->bs, xc, blend_sample(o, c, sample, g, o->scale_lut[c][(sample >> o->bs) & 0xff]));
		}
	}
}

void vfgs_oracle_add_grain_frame_closed_form(vfgs_oracle* o, void* Y, void* U, void* V,
                                             int width, int height, int stride, int cstride)
{
	int nblk = (width + 15) / 16;
	int nbr = (height + 15) / 16;
	int sz = o->bs ? 2 : 1;
	int crows = (height + o->csuby - 1) / o->csuby;
	uint64_t nbits = (uint64_t)(nbr + 1) * nblk + 64;
	uint64_t nwords = (nbits >> 5) + 2;
	uint32_t* cur = (uint32_t*)malloc(nwords * 4);
	uint32_t* up0 = (uint32_t*)malloc(((uint64_t)nblk / 32 + 4) * 4);
	size_t ysize = (size_t)stride * height * sz, csize = (size_t)cstride * crows * sz;
	uint8_t* ycopy = (uint8_t*)malloc(ysize);
	uint8_t* ucopy = (uint8_t*)malloc(csize);
	uint8_t* vcopy = (uint8_t*)malloc(csize);
	cf_ctx k;

	vfgs_oracle_lfsr_stream(o->line_rnd, cur, nwords);
	vfgs_oracle_lfsr_stream(o->line_rnd_up, up0, (uint64_t)nblk / 32 + 4);
	k.o = o; k.cur_stream = cur; k.up0_stream = up0; k.nblk = nblk;

	/* in-place semantics: every sample is a function of the *input* frame */
	memcpy(ycopy, Y, ysize);
	memcpy(ucopy, U, csize);
	memcpy(vcopy, V, csize);
	cf_plane(&k, ycopy, Y, stride, 0, height);
	cf_plane(&k, ucopy, U, cstride, 1, height);
	cf_plane(&k, vcopy, V, cstride, 2, height);

	/* registers after the last line (derivation in DESIGN.md "seed registers") */
	if (height > 0)
	{
		uint32_t lu = o->line_rnd_up;
		o->line_rnd = stream_window(cur, (uint64_t)(nbr - 1) * nblk);
		o->rnd = stream_window(cur, (uint64_t)nbr * nblk);
		if (nbr >= 2)
		{
			o->line_rnd_up = stream_window(cur, (uint64_t)(nbr - 2) * nblk);
			o->rnd_up = stream_window(cur, (uint64_t)(nbr - 1) * nblk);
		}
		else
		{
			o->line_rnd_up = lu;
			o->rnd_up = stream_window(up0, (uint64_t)nblk);
		}
	}

	free(cur); free(up0); free(ycopy); free(ucopy); free(vcopy);
}

/* ---- driving a foreign line function (the real reference) ------------------ */

void vfgs_oracle_drive_lines(vfgs_line_fn fn, void* Y, void* U, void* V, int width, int height,
                             int stride, int cstride, int sz, int csuby)   /* vfgs_main.c:664-682 */
{
	uint8_t *py = (uint8_t*)Y, *pu = (uint8_t*)U, *pv = (uint8_t*)V;
	for (int y = 0; y < height; y++)
	{
		fn(py, pu, pv, y, width);
		py += (size_t)stride * sz;
		if ((y & 1) || csuby == 1)
		{
			pu += (size_t)cstride * sz;
			pv += (size_t)cstride * sz;
		}
	}
}

/* ---- synthetic input ------------------------------------------------------ */

uint32_t vfgs_oracle_lcg_fill(uint32_t x, void* dst, uint64_t nsamples, int depth) /* SURVEY.md Appendix B */
{
	uint32_t mask = (1u << depth) - 1;
	if (depth > 8)
	{
		uint16_t* p = (uint16_t*)dst;
		for (uint64_t i = 0; i < nsamples; i++) { x = x * 1664525u + 1013904223u; p[i] = (uint16_t)((x >> 16) & mask); }
	}
	else
	{
		uint8_t* p = (uint8_t*)dst;
		for (uint64_t i = 0; i < nsamples; i++) { x = x * 1664525u + 1013904223u; p[i] = (uint8_t)((x >> 16) & mask); }
	}
	return x;
}
