// Developer bench (round 6): variants of the word-recurrence extension of an LFSR segment (vfgs_host.cpp lfsr_extend): plain loops, explicit chunks,
// SSE2 chunks of four (store-forwarding stalls), chunks as long as the lag, and the shipped form with the previous chunk in registers.
//   g++ -O3 -std=c++17 tools/dev/lfsr_extend_bench.cpp -o /tmp/lfsr_extend_bench && /tmp/lfsr_extend_bench
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <algorithm>
#include <emmintrin.h>
static uint32_t lfsr_step(uint32_t r) { return (r >> 1) | ((((r >> 1) ^ (r >> 29)) & 1u) << 31); }
__attribute__((noinline)) void ext_a(uint32_t* __restrict w, size_t n)
{
	size_t i = 32;
	for (; i < n && i < 63; i++) w[i] = w[i - 31] ^ w[i - 3];
	for (; i < n && i < 125; i++) w[i] = w[i - 62] ^ w[i - 6];
	for (; i < n && i < 249; i++) w[i] = w[i - 124] ^ w[i - 12];
	for (; i + 24 <= n; i += 24)
	{
		uint32_t* __restrict d = w + i; const uint32_t* __restrict a = w + i - 248; const uint32_t* __restrict b = w + i - 24;
		for (int k = 0; k < 24; k++) d[k] = a[k] ^ b[k];
	}
	for (; i < n; i++) w[i] = w[i - 248] ^ w[i - 24];
}
template <int LA, int LB, int CH> static inline size_t phase(uint32_t* w, size_t i, size_t end)
{
	for (; i + CH <= end; i += CH)
	{
		uint32_t t[CH];
		for (int k = 0; k < CH; k++) t[k] = w[i - LA + k] ^ w[i - LB + k];
		for (int k = 0; k < CH; k++) w[i + k] = t[k];
	}
	for (; i < end; i++) w[i] = w[i - LA] ^ w[i - LB];
	return i;
}
__attribute__((noinline)) void ext_b(uint32_t* w, size_t n)
{
	size_t i = 32;
	i = phase<31, 3, 3>(w, i, std::min<size_t>(n, 63));
	i = phase<62, 6, 4>(w, i, std::min<size_t>(n, 125));
	i = phase<124, 12, 12>(w, i, std::min<size_t>(n, 249));
	i = phase<248, 24, 24>(w, i, n);
}
template <int LA, int LB> static inline size_t phase4(uint32_t* w, size_t i, size_t end)
{
	static_assert(LB >= 4, "four words per step");
	for (; i + 4 <= end; i += 4)
		_mm_storeu_si128((__m128i*)(w + i), _mm_xor_si128(_mm_loadu_si128((const __m128i*)(w + i - LA)), _mm_loadu_si128((const __m128i*)(w + i - LB))));
	for (; i < end; i++) w[i] = w[i - LA] ^ w[i - LB];
	return i;
}
__attribute__((noinline)) void ext_c(uint32_t* w, size_t n)
{
	size_t i = 32;
	for (; i < n && i < 63; i++) w[i] = w[i - 31] ^ w[i - 3];
	i = phase4<62, 6>(w, i, std::min<size_t>(n, 125));
	i = phase4<124, 12>(w, i, std::min<size_t>(n, 249));
	i = phase4<248, 24>(w, i, n);
}
// chunks as long as the short lag, moved in pieces whose loads meet exactly the stores of the chunk before (store forwarding)
template <int LA, int LB> static inline size_t phase_sse(uint32_t* w, size_t i, size_t end)
{
	static_assert(LB % 4 == 0, "whole 16-byte pieces");
	for (; i + LB <= end; i += LB)
		for (int k = 0; k < LB; k += 4)
			_mm_storeu_si128((__m128i*)(w + i + k), _mm_xor_si128(_mm_loadu_si128((const __m128i*)(w + i + k - LA)), _mm_loadu_si128((const __m128i*)(w + i + k - LB))));
	for (; i < end; i++) w[i] = w[i - LA] ^ w[i - LB];
	return i;
}
static inline uint64_t ld64(const uint32_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st64(uint32_t* p, uint64_t v) { memcpy(p, &v, 8); }
__attribute__((noinline)) void ext_d(uint32_t* w, size_t n)
{
	size_t i = 32;
	for (; i < n && i < 63; i++) w[i] = w[i - 31] ^ w[i - 3];
	const size_t e6 = std::min<size_t>(n, 125);
	for (; i + 6 <= e6; i += 6)
		for (int k = 0; k < 6; k += 2) st64(w + i + k, ld64(w + i + k - 62) ^ ld64(w + i + k - 6));
	for (; i < e6; i++) w[i] = w[i - 62] ^ w[i - 6];
	i = phase_sse<124, 12>(w, i, std::min<size_t>(n, 249));
	i = phase_sse<248, 24>(w, i, n);
}
// the chunk before stays in registers: the chain of dependent steps costs an XOR per step instead of a store -> load round trip
__attribute__((noinline)) void ext_e(uint32_t* w, size_t n)
{
	size_t i = 32;
	{
		uint32_t p0 = w[29], p1 = w[30], p2 = w[31];
		for (; i + 3 <= n && i + 3 <= 63 + 2; i += 3)      // (words 32 .. 64: the base recurrence is valid from word 32 on, also beyond 62)
		{
			p0 ^= w[i - 31]; p1 ^= w[i - 30]; p2 ^= w[i - 29];
			w[i] = p0; w[i + 1] = p1; w[i + 2] = p2;
		}
	}
	for (; i < n && i < 65; i++) w[i] = w[i - 31] ^ w[i - 3];
	if (i >= n) return;
	{
		// i == 65: lag (62, 6), valid from word 63 on
		uint64_t q0 = ld64(w + i - 6), q1 = ld64(w + i - 4), q2 = ld64(w + i - 2);
		for (; i + 6 <= n && i + 6 <= 125 + 6; i += 6)
		{
			q0 ^= ld64(w + i - 62); q1 ^= ld64(w + i - 60); q2 ^= ld64(w + i - 58);
			st64(w + i, q0); st64(w + i + 2, q1); st64(w + i + 4, q2);
		}
	}
	if (i + 12 <= n)
	{
		// lag (124, 12), valid from word 125 on
		__m128i r0 = _mm_loadu_si128((const __m128i*)(w + i - 12)), r1 = _mm_loadu_si128((const __m128i*)(w + i - 8)), r2 = _mm_loadu_si128((const __m128i*)(w + i - 4));
		for (; i + 12 <= n && i + 12 <= 249 + 12; i += 12)
		{
			r0 = _mm_xor_si128(r0, _mm_loadu_si128((const __m128i*)(w + i - 124)));
			r1 = _mm_xor_si128(r1, _mm_loadu_si128((const __m128i*)(w + i - 120)));
			r2 = _mm_xor_si128(r2, _mm_loadu_si128((const __m128i*)(w + i - 116)));
			_mm_storeu_si128((__m128i*)(w + i), r0); _mm_storeu_si128((__m128i*)(w + i + 4), r1); _mm_storeu_si128((__m128i*)(w + i + 8), r2);
		}
	}
	if (i + 24 <= n)
	{
		__m128i r[6];
		for (int k = 0; k < 6; k++) r[k] = _mm_loadu_si128((const __m128i*)(w + i - 24 + 4 * k));
		for (; i + 24 <= n; i += 24)
			for (int k = 0; k < 6; k++)
			{
				r[k] = _mm_xor_si128(r[k], _mm_loadu_si128((const __m128i*)(w + i - 248 + 4 * k)));
				_mm_storeu_si128((__m128i*)(w + i + 4 * k), r[k]);
			}
	}
	for (; i < n; i++) w[i] = i >= 249 ? w[i - 248] ^ w[i - 24] : i >= 125 ? w[i - 124] ^ w[i - 12] : i >= 63 ? w[i - 62] ^ w[i - 6] : w[i - 31] ^ w[i - 3];
}
int main()
{
	const int nseg = 64, nw = 531;
	static uint32_t buf[64 * 531 + 64], t[4][256];
	for (int k = 0; k < 4; k++) for (int v = 0; v < 256; v++) t[k][v] = (uint32_t)(v * 2654435761u + k * 40503u);
	uint32_t head[32], reg = 24690;
	for (int i = 0; i < 32; i++) { head[i] = reg; for (int k = 0; k < 32; k++) reg = lfsr_step(reg); }
	{
		static uint32_t x[700], y[700];
		for (int n : {33, 40, 63, 64, 65, 66, 100, 125, 126, 131, 137, 249, 250, 260, 273, 300, 531, 600, 699})
		{
			memcpy(x, head, 128); memcpy(y, head, 128);
			for (int i = 32; i < n; i++) x[i] = x[i - 31] ^ x[i - 3];
			ext_e(y, n);
			if (memcmp(x, y, n * 4)) { printf("ext_e WRONG at n = %d\n", n); return 1; }
		}
		printf("ext_e == the word recurrence for every length tried\n");
	}
	for (int variant = 0; variant < 5; variant++)
	{
		auto t0 = std::chrono::steady_clock::now();
		uint32_t acc = 0;
		const int reps = 20000;
		for (int r = 0; r < reps; r++)
		{
			uint32_t h[32];
			memcpy(h, head, sizeof h);
			for (int f = 0; f < nseg; f++)
			{
				uint32_t* w = buf + (size_t)f * nw;
				memcpy(w, h, sizeof h);
				if (variant == 0) ext_a(w, nw); else if (variant == 1) ext_b(w, nw); else if (variant == 2) ext_c(w, nw); else if (variant == 3) ext_d(w, nw); else ext_e(w, nw);
				if (variant < 2 || true)
					for (int i = 0; i < 32; i++) { const uint32_t x = h[i]; h[i] = t[0][x & 255] ^ t[1][(x >> 8) & 255] ^ t[2][(x >> 16) & 255] ^ t[3][x >> 24]; }
			}
			acc ^= buf[r % (nseg * nw)];
		}
		const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
		printf("variant %d (%s): %.2f us per 64-segment image, %.0f ns per segment (%u)\n", variant, variant == 0 ? "shipped loops" : variant == 1 ? "explicit chunks" : variant == 2 ? "SSE2 chunks of four" : variant == 3 ? "chunks = lag, forwardable" : "previous chunk in registers", us, us * 1000 / nseg, acc);
	}
	return 0;
}
