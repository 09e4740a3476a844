// Shared between the host state machine (vfgs_host.cpp) and the gfx950 kernels
// (vfgs_kernel.hip): the LDS image of the pattern banks + LUTs and the launch record.
#pragma once
#include <stdint.h>

namespace vfgs {

constexpr int kSlots = 8;        // pattern slots per component, vfgs_hw.h:49
constexpr int kMaxUnits = 64;    // 16-byte units per position (one per lane)
constexpr int kBlock = 16;       // luma samples per grain block

// Tuning knobs (defaults are the shipped configuration; tools/dev/build_variant.sh overrides them).  They change HOW the
// kernels run, never WHAT they compute: the timing-only probes with wrong output that rounds 2 and 3 kept in the kernel source
// (their results: profiles/r02_variants*.log, r03_ab*.log) were removed in round 4, and so were the knobs of
// round 2's tiled kernels when the row walk became the only kernel family.
#ifndef VFGS_WAVES
#define VFGS_WAVES 4          // waves per workgroup (power of two); they share one LDS image
#endif
#ifndef VFGS_WG_PER_CU
#define VFGS_WG_PER_CU 4      // resident workgroups per CU the register allocation is sized for
#endif
#ifndef VFGS_SCHED_FENCE
#define VFGS_SCHED_FENCE 1    // 1: a scheduling fence after every position keeps its store and refill in program order
#endif
#ifndef VFGS_LDAUX_ALIGNED
#define VFGS_LDAUX_ALIGNED 2  // cache policy bits of the sample loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1): nontemporal
#endif
#ifndef VFGS_STAUX_ALIGNED
#define VFGS_STAUX_ALIGNED 2  // ... of the sample stores
#endif
// Register sets of a wave's ring = positions (1 KiB wave accesses) it has in flight: a position is stored that many positions after its load
// was issued.  Pure streams at the kernels' occupancy prefer a SHORT distance between the load and the store of the same bytes (tools/walk_probe.hip
// depth: one set 0.775, two 0.744, four 0.727 of 8 TB/s at 16-20 waves per CU, profiles/r06_walk_ring_depth.log), the kernels need the loads of the
// next positions in flight while they compute this one: four for the general forms and at 10 bit (two: -1 % at the headline, -3 % for the 8-bit general form),
// two for the packed 16-bit planes of the 8-bit one-pattern kernels (+1.3 .. 3 %, and 8 VGPRs fewer: six waves per SIMD without a spill, +1.4 .. 3.4 %
// together; profiles/r06_ab3 .. r06_ab5).
#ifndef VFGS_RING
#define VFGS_RING 4           // general-form planes
#endif
#ifndef VFGS_RING_ONE10
#define VFGS_RING_ONE10 4     // one-pattern planes at 10 bit
#endif
#ifndef VFGS_RING_PK
#define VFGS_RING_PK 2        // packed 16-bit planes (8-bit one-pattern form)
#endif
#ifndef VFGS_RING_NARROW10
#define VFGS_RING_NARROW10 2  // one-pattern planes at 10 bit whose rows are one or two positions (1080p chroma): 1080p fgs_sei +2.4 % at 8, +1.1 % at 32 frames per launch, ff_test1 +0 .. 0.3 % (profiles/r06_ab8)
#endif
// Resident workgroups per CU of the all-one-pattern kernels, held there by unused LDS behind their tables (lds_allocation below; 0 = as many as fit).
// Their 15 KB of LDS and ~80 registers would let SIX in: 24 waves with four 1 KiB loads each in flight per CU, and the memory system answers a CU
// that asks for less at once better (tools/walk_probe.hip depth: 20-24 KiB of loads in flight per CU stream at 0.79-0.80 of 8 TB/s, 64-96 KiB at
// 0.73-0.75).  At 10 bit FOUR are worth +1.5 .. 4 % from 720p to 4320p (three: up to +5.7 % at 1080p / 2160p, -5 % at 720p;
// profiles/r06_ab13_workgroups_per_cu.log, r06_ab14_workgroups_per_cu_sizes.log); the 8-bit kernels (twice the arithmetic per byte) lose 2-3 % at five
// and four and keep their six.  The general-form kernels are at four by their 40 KB image, and three cost them 2.5 % (r06_ab11).
#ifndef VFGS_ONE10_WG_PER_CU
#define VFGS_ONE10_WG_PER_CU 4
#endif
#ifndef VFGS_ONE8_WG_PER_CU
#define VFGS_ONE8_WG_PER_CU 0
#endif
#ifndef VFGS_RW_CONSEC
#define VFGS_RW_CONSEC 0      // 1 = a wave's rows are consecutive, 0 = the waves of a workgroup take every kWavesPerWG-th row
#endif

// The product is built with every knob at its default (versatilefilmgrain_amd/build.py passes none).  The developer tools that
// time variants (tools/dev/build_variant.sh, tools/gpu_variants.sh) define VFGS_DEV_BUILD; without it any
// other value is a build error, so a stray -D cannot produce a library that silently runs something else -- and a
// developer build says so at run time (vfgs_hip_dev_build(), refused by versatilefilmgrain_amd.hw unless asked for).
#if !defined(VFGS_DEV_BUILD)
#if VFGS_WAVES != 4 || VFGS_WG_PER_CU != 4 || VFGS_RING != 4 || VFGS_RING_ONE10 != 4 || VFGS_RING_PK != 2 || VFGS_RING_NARROW10 != 2 || VFGS_ONE10_WG_PER_CU != 4 || VFGS_ONE8_WG_PER_CU != 0 || VFGS_SCHED_FENCE != 1 || VFGS_LDAUX_ALIGNED != 2 || VFGS_STAUX_ALIGNED != 2 || VFGS_RW_CONSEC != 0 || \
    defined(VFGS_NO_FRONTS) || defined(VFGS_NO_LOOKAHEAD) || defined(VFGS_NO_ONE_PATTERN) || defined(VFGS_NO_PK16) || defined(VFGS_PK_NO_READ2) || defined(VFGS_PK_WAVES) || defined(VFGS_ONE10_WAVES) ||  defined(VFGS_RW_WG_BYTES) || defined(VFGS_RW_MIN_FILL_PCT) || defined(VFGS_PERSIST_MIN_TASKS) || defined(VFGS_PERSIST_MAX_WG_KB)
#error "libvfgs_hip: a tuning knob differs from the shipped configuration; developer variants must define VFGS_DEV_BUILD"
#endif
#endif

constexpr int kWavesPerWG = VFGS_WAVES;
#ifdef VFGS_NO_PK16               // developer builds only: the 8-bit one-pattern components keep round 5's byte bank + dword LUT (same-box A/Bs)
constexpr bool kPk16 = false;
#else
constexpr bool kPk16 = true;      // 8-bit one-pattern components use the packed 16-bit form (below)
#endif

// The kernels (vfgs_kernel.hip "Row walk"): a wave streams whole rows, the workgroup's block parameters live in LDS
// behind the table image: two tables (this block row's registers, the block row above's) of one dword per grain block +
// one block in front.  A table holds kTileBlocks blocks: a row of more blocks is walked in parts, the table refilled between them.
constexpr int kTileBlocks = 512;
constexpr int kParamEntries = kTileBlocks + 4;   // entry e = block e - 1; the lanes behind a row's end read up to block nblk + 2 (clamped values)
constexpr int kParamTableBytes = (kParamEntries * 4 + 15) & ~15;
constexpr int kParamBytes = 2 * kParamTableBytes;

// LDS a grain kernel ALLOCATES: what it uses (table image + block parameters), padded where its class is held at fewer resident
// workgroups per CU than would fit (VFGS_ONE10_WG_PER_CU / VFGS_ONE8_WG_PER_CU above): a size with which exactly that many are resident.
// The kernels with one-pattern luma at 10 bit (rows walked in parts included: 92-95 registers, five would be resident -- 16384-wide AFGS1 +2.5 % at two
// frames per launch, +1 % at four: profiles/r06_ab18) need 15 KB and allocate 40; the kernels with a general-form plane are at four by their 40 KB image.
constexpr int kLdsPerCU = 163840;
constexpr int lds_allocation(const bool depth10, const bool one_y, const bool one_c, const bool wide, const int need)
{
	// (10 bit: every kernel whose LUMA is one-pattern -- over general-form chroma at 4:2:0 / 4:2:2 it needs 15-23 KB and five would be resident: +0.9 .. 1.3 %,
	// profiles/r06_ab23; over general-form chroma at 4:4:4 the 40 KB image decides anyway)
	const int cap = (depth10 && one_y) ? VFGS_ONE10_WG_PER_CU : ((!depth10 && one_y && one_c && !wide) ? VFGS_ONE8_WG_PER_CU : 0);
	if (cap <= 0) return need;
	const int target = (kLdsPerCU / cap) & ~2047;        // exactly `cap` workgroups resident (well inside any allocation granule) ...
	const int next = (kLdsPerCU / (cap + 1)) & ~2047;    // ... and a size with which cap + 1 would be
	return need > next ? need : (target < 65536 ? target : 65536);
}

// Device image of everything the kernel looks up: one sub-image per plane type (luma; chroma) -- or per chroma
// component -- and a workgroup (which works on ONE plane) copies the sub-image of its plane to LDS offset 0.
//
//   luma image  : [LUT Y : 2 x 256 dwords] [luma bank]          (one-pattern form: 1 x 256 dwords, the +scale table, then the bank and its negated copy)
//   Cb image    : [LUT Cb: 2 x 256 dwords] [chroma bank]        (general form: the chroma bank is stored twice in the device image, so that a
//   Cr image    : [LUT Cr: 2 x 256 dwords] [chroma bank]         workgroup -- which serves ONE component -- stages one LUT pair, not two: with both,
//                                                                 the 4:4:4 chroma image was 2 KB larger than the luma image and a CU held three
//                                                                 workgroups where the kernels are built for four)
//             or: [LUT Cb] [bank of Cb's pattern]   and   [LUT Cr] [bank of Cr's pattern]          (one-pattern form)
//
// GENERAL form of a bank ("slot-interleaved"): for every (row, column) position the eight slots' int8 values sit in 8
// consecutive bytes, so the LDS address of a sample's pattern data does NOT depend on the sample's intensity (the slot
// is picked afterwards in registers with v_perm_b32); four samples are two ds_read_b128.  Each bank row is padded by one
// 16-byte slot so consecutive rows rotate through the sixteen 16-byte LDS slots of a 256-byte bank row.
//
// ONE-PATTERN form: when a component's pattern LUT selects the same slot for every intensity (all AFGS1 models, the
// single-pattern SEI models, the chroma of the default SEI model) only that pattern is stored, one byte per sample, rows
// padded by 16 bytes; four samples are one dword.  An eighth of the LDS traffic and of the LDS footprint.  The bank is
// followed by its NEGATED copy (y_neg / c_neg bytes further on): a block's random sign then is a choice of bank, made once
// per block and row in the address, the values that come out of LDS are the true signed grain (plain edge filter, no
// relative signs) and every sample uses the +scale table at LDS offset 0 (no table select in the gather address).  The host
// only chooses this form for patterns without the value -128, which has no negation in a byte (the firmware's generators
// clip to +-127, vfgs_fw.c:321-323,494).
//
// ONE-PATTERN form at 8 BIT ("packed 16-bit form", round 6): the bank holds the pattern as int16 values (rows of 2 x columns + 16
// bytes, followed by the negated copy) and the LUT is the plain table of the 256 scale BYTES (vfgs_hw.c:50).  Two samples then
// share every vector instruction behind the gathers: a pair's pattern values arrive packed out of LDS (four samples = one
// ds_read_b64), v_pk_mad_i16 (pattern x scale + 2^(shift-1)), v_pk_ashrrev_i16, and the packed add / max / min of the clip --
// exact because |P| <= 127 and the host only chooses the form when max(scale) * 127 + 2^(shift-1) <= 32767 (image_form in
// vfgs_host.cpp; otherwise the general form serves the component).  The two samples at a block edge (after the 3-tap filter
// up to +-159) and the two overlap lines of a block row (after the blend up to +-199) keep 32-bit products.
//
// LUT (one dword per 8-bit intensity; per component TWO tables of 256 entries, the first with
// +scale, the second with -scale, so that a block's random sign is applied by choosing the table
// instead of multiplying every sample).  A component's pair of tables starts at a multiple of
// 2048 bytes below 64 KiB: the LDS address of a sample's entry is (4 * intensity) | base | sign << 10,
// two samples per v_and_or_b32.  Entry:
//   bits 31:24 byte selector for v_perm_b32: slot 0..7, or 0x0c (constant zero) for slot 8
//              (the reference's never-written 9th slot, vfgs_hw.c:49); unused in the one-pattern form
//   bits 23:0  signed scale factor (+-sLUT), pre-shifted: scale << (16 - scale_shift)
struct ImageLayout {
	int lut_bytes;              // one component: +scale table, -scale table
	int y_rs, c_rs;             // bank row strides, bytes
	int y_bank, c_bank;         // offsets of the banks inside their sub-images (= LDS offsets)
	int y_neg, c_neg;           // one-pattern form: bytes from a bank to its negated copy (0 in the general form)
	int y_bytes, c_bytes;       // sizes of the sub-images (chroma: of ONE chroma sub-image)
	int y_off, c_off[2];        // offsets of the sub-images of Y, Cb, Cr in the device image
	int c_lut[2];               // LDS offset of Cb's / Cr's LUT pair inside its sub-image
	int bytes;                  // whole device image
	int lds_bytes;              // LDS a workgroup needs
	int cw, ch;                 // chroma bank columns actually addressable, rows
};

constexpr ImageLayout image_layout(int csubx, int csuby, bool one_y, bool one_c, bool depth8)
{
	ImageLayout L{};
	L.lut_bytes = 2 * 256 * 4;
	L.cw = 64 / csubx;
	L.ch = 64 / csuby;
	depth8 = depth8 && kPk16;
	const int ob = depth8 ? 2 : 1;     // bytes per sample of a one-pattern bank (8 bit: the packed 16-bit form)
	L.y_rs = one_y ? 64 * ob + 16 : 64 * kSlots + 16;
	L.c_rs = one_c ? L.cw * ob + 16 : L.cw * kSlots + 16;
	// (the one-pattern form never reads the -scale table: its sign is a choice of bank; only the +scale half is stored -- at 8 bit
	// as the 256 scale bytes)
	L.y_bank = one_y ? (depth8 ? 256 : L.lut_bytes / 2) : L.lut_bytes;
	L.c_bank = one_c ? (depth8 ? 256 : L.lut_bytes / 2) : L.lut_bytes;
	L.y_neg = one_y ? 64 * L.y_rs : 0;
	L.c_neg = one_c ? L.ch * L.c_rs : 0;
	L.y_bytes = L.y_bank + 64 * L.y_rs + L.y_neg;
	L.c_bytes = L.c_bank + L.ch * L.c_rs + L.c_neg;
	L.y_off = 0;
	L.c_off[0] = L.y_bytes;
	L.c_off[1] = L.y_bytes + L.c_bytes;
	L.c_lut[0] = 0;
	L.c_lut[1] = 0;
	L.bytes = L.y_bytes + 2 * L.c_bytes;
	L.lds_bytes = L.y_bytes > L.c_bytes ? L.y_bytes : L.c_bytes;
	return L;
}

// Geometry of one plane type (0 = luma, 1 = the two chroma planes) for one launch.
//
// A row of the plane is cut into 16-byte "units" (one per lane and access), 64 consecutive units are a "position" (one wave
// access, 1 KiB); the lanes compute bytes shifted left against the units so that block edges lie inside lanes / lane pairs
// (vfgs_kernel.hip "Lanes"), which is why a row has one position more than its units fill.  One workgroup = kWavesPerWG waves
// x rw_rpw rows each (wave w: rows w, w + kWavesPerWG, ... of the workgroup's part of ONE block row), rw_splits workgroups per
// block row.
struct PlaneDesc {
	uint32_t pitch, dpitch;       // row pitch of source / destination, bytes
	uint64_t fpitch, dfpitch;     // bytes from frame f to frame f+1
	uint32_t rowbytes, drowbytes; // bytes of a row the reference touches (whole blocks)
	int nrows;                    // rows of the stripe
	int wgs;                      // workgroups per frame for ONE plane of this type
	int rw_segs;                  // positions per row
	int rw_rpw, rw_splits, rw_lsplits;
};

// Frames of a batch that do NOT lie at a constant pitch (a decoder's pool of separately allocated frames, vfgs_hip_add_grain_frame_list_*):
// their plane pointers travel in the kernel arguments themselves -- no device table to upload and keep alive, no copy on the
// stream in front of the launch -- and a workgroup fetches its frame's pointers with one scalar load.  kListFrames frames per
// launch (longer lists: several launches); 2 x 3 x 8 x kListFrames bytes of the 4 KB argument segment.
constexpr int kListFrames = 32;
struct FrameTable {
	const uint8_t* src[3][kListFrames];   // [component][frame of the launch]: first line of the stripe
	uint8_t* dst[3][kListFrames];
};

// One launch = nframes x (luma workgroups + 2 x chroma workgroups); workgroups are numbered in memory
// order (frame, plane, block row group, split, column group) and are NOT persistent: the hardware
// dispatcher hands them out as CUs free up.
struct KernelArgs {
	const uint8_t* src[3];    // source planes: first line of the stripe, frame 0 (device)
	uint8_t* dst[3];          // destination planes, same geometry (== source: in place)
	PlaneDesc pd[2];
	const uint32_t* stream;   // LFSR bit stream cache (device), bit m = word[m>>5] >> (m&31)
	uint32_t stream_bytes;
	const uint8_t* tables;    // device image (image_layout)
	uint32_t cur_bit0;        // stream bit of the register of block 0, first block row of the stripe, frame 0
	uint32_t up_bit0;         // same for the "upper" register of that first block row
	uint32_t frame_bit_step;  // stream bits between consecutive frames of a batch
	int y0;                   // absolute luma line of the first line of the stripe (multiple of 16 unless the stripe is one block row)
	int nblk;                 // 16-sample blocks per line = ceil(width/16), vfgs_hw.c:301
	int nbrows;               // block rows the stripe touches
	int nframes;
	int listed;               // 1: the planes of frame f are FrameTable::src / dst [.][f] (src / dst above and the frame pitches are unused)
	int lfronts;              // log2 of the frames of a batch that are swept at the same time (their workgroups are dealt out in turn)
	int persist_wgs;          // PERSIST kernels: P luma workgroups share the launch's nframes x pd[0].wgs luma tasks (task t -> workgroup t % P) ...
	int persist_step_f, persist_step_r;   // ... and P = persist_step_f * pd[0].wgs + persist_step_r: what a workgroup advances by
	int pk_shift;             // 8-bit one-pattern forms (packed 16-bit form, above): the scale shift of vfgs_hw.c:263, 8..13
	uint32_t lo2[2], hi2[2];  // clip bounds in sample units (I_min<<bs ...) in both halves of a dword, per plane type (vfgs_hw.c:264-267)
};

}  // namespace vfgs
