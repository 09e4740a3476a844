// Shared between the host state machine (vfgs_host.cpp) and the gfx950 kernels
// (vfgs_kernel.hip): the LDS image of the pattern banks + LUTs and the launch record.
#pragma once
#include <stdint.h>

namespace vfgs {

constexpr int kSlots = 8;        // pattern slots per component, vfgs_hw.h:49
constexpr int kMaxUnits = 64;    // 8-sample units per work item row (one per lane)
// Tuning knobs (defaults are the shipped configuration; tools/ablate.py overrides them).
#ifndef VFGS_WAVES
#define VFGS_WAVES 12
#endif
#ifndef VFGS_WG_PER_CU
#define VFGS_WG_PER_CU 2
#endif
#ifndef VFGS_LDAUX
#define VFGS_LDAUX 0      // cache policy bits of the sample loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef VFGS_STAUX
#define VFGS_STAUX 0      // cache policy bits of the sample stores
#endif
#ifndef VFGS_PIPE
#define VFGS_PIPE 0       // 1: issue the next item's loads before computing the current item (two register sets)
#endif
#ifndef VFGS_ABLATE
#define VFGS_ABLATE 0   // 0 = product.  >0: timing-only variants with WRONG output (tools/ablate.py):
                        //   1 copy only, 5 no stores, 6 no table staging, 7 block parameters of segment 0 reused,
                        //   8 no LUT gather, 9 no pattern fetch, 10 = 8 + 9, 11 = 1 + 6, 12 = 1 without LFSR loads, 13 = 11 + 12
#endif
constexpr int kWavesPerWG = VFGS_WAVES;     // waves per workgroup, one LDS image each
constexpr int kWGPerCU = VFGS_WG_PER_CU;    // resident workgroups per CU the grid is sized for
constexpr int kBlock = 16;       // luma samples per grain block

// LDS / device image of everything the kernel looks up.
//
// Pattern banks are stored "slot-interleaved": for every (row, column) position the
// eight slots' int8 values sit in 8 consecutive bytes, so the LDS address of a sample's
// pattern data does NOT depend on the sample's intensity (the slot is picked afterwards
// in registers with v_perm_b32).  A lane's 8 samples are then 4 ds_read_b128.
// Each bank row is padded by one 16-byte slot so consecutive rows rotate through the
// sixteen 16-byte LDS slots of a 256-byte bank row.
//
// LUT entry (one dword per 8-bit intensity; per component TWO tables of 256 entries, the
// first with +scale, the second with -scale, so that a block's random sign can be applied by
// choosing the table instead of multiplying every sample):
//   bits 31:24 byte selector for v_perm_b32: slot 0..7, or 0x0c (constant zero) for slot 8
//              (the reference's never-written 9th slot, vfgs_hw.c:49)
//   bits 15:0  signed scale factor (+-sLUT)
template <int CSUBX, int CSUBY>
struct TableLayout {
	static constexpr int LRS = 64 * kSlots + 16;        // luma bank row stride, bytes
	static constexpr int CW = 64 / CSUBX;               // chroma bank columns actually addressable
	static constexpr int CH = 64 / CSUBY;               // chroma bank rows
	static constexpr int CRS = CW * kSlots + 16;        // chroma bank row stride, bytes
	static constexpr int LUMA_OFF = 0;
	static constexpr int CHROMA_OFF = 64 * LRS;
	static constexpr int LUT_OFF = CHROMA_OFF + CH * CRS;
	static constexpr int BYTES = LUT_OFF + 3 * 2 * 256 * 4;
	static_assert(BYTES % 16 == 0, "image is copied in 16-byte pieces");
};

// One launch = nframes x (Y rows + Cb rows + Cr rows of the stripe) x tiles per row work items;
// one item = one row of one plane x 4 segments of <= 64 lanes x 8 samples, owned by ONE wavefront.
struct KernelArgs {
	const uint8_t* Y;         // source: line `y0` of frame 0 (device)
	const uint8_t* U;         // source: chroma row y0/csuby of frame 0
	const uint8_t* V;
	uint8_t* dY;              // destination planes, same geometry (== source: in place)
	uint8_t* dU;
	uint8_t* dV;
	uint32_t y_extent;        // bytes of one frame's luma stripe (rows * pitch), < 2^31: buffer range check
	uint32_t c_extent;        // bytes of one frame's chroma stripe, per plane
	uint64_t y_frame_pitch;   // bytes from frame f to frame f+1 (batched launches)
	uint64_t c_frame_pitch;
	// destination geometry; identical to the source's unless the output is narrowed to 8 bit
	uint32_t dy_extent, dc_extent;
	uint64_t dy_frame_pitch, dc_frame_pitch;
	int dstride, dcstride;    // samples
	const uint32_t* stream;   // LFSR bit stream cache (device), bit m = word[m>>5] >> (m&31)
	const uint8_t* tables;    // TableLayout image (device)
	uint32_t cur_bit0;        // stream bit of the register of block 0, first block row of the stripe, frame 0
	uint32_t up_bit0;         // same for the "upper" register of that first block row
	uint32_t frame_bit_step;  // stream bits between consecutive frames of a batch
	int y0;                   // absolute luma line of the first line of the stripe
	int nlines;               // luma lines in the stripe
	int nblk;                 // 16-sample blocks per line = ceil(width/16), vfgs_hw.c:301
	// work items (vfgs_kernel.hip "Work items"): one row of one plane x 4 segments
	int upt_y, segs_y, tiles_y;   // planes with 16-sample blocks: units per segment (even, <= 64), segments and tiles per row
	int upt_c, segs_c, tiles_c;   // chroma planes (8-sample blocks: edges per segment; else copies of the *_y values)
	int crow_first, ncrows;       // chroma rows of the stripe: first absolute row, count
	int items_y, items_c;         // nlines * tiles_y, ncrows * tiles_c
	int nitems;                   // nframes * (items_y + 2 * items_c)
	uint32_t chroma_off, lut_off; // TableLayout offsets of the chroma bank and the LUTs
	int stride, cstride;      // samples
	int nframes;
	int scale_shift;          // vfgs_hw.c:56 (already includes +6-bs)
	int ylo, yhi, clo, chi;   // clip bounds in sample units (I_min<<bs ...), vfgs_hw.c:264-267
};

}  // namespace vfgs
