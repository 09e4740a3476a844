/*
 * libvfgs_hip -- MI355X (gfx950) film grain synthesis hardware layer.
 *
 * C ABI.  Two groups of entry points:
 *
 * (1) DROP-IN: the ten functions of the reference hardware-layer interface,
 *     /root/reference/src/vfgs_hw.h:51-62, with identical names, argument
 *     meaning and (absence of) error reporting, so that the reference's
 *     vfgs_fw.c (callers: vfgs_fw.c:585,594,637,638,643,672-704) and
 *     vfgs_main.c (callers: vfgs_main.c:674,750,751,760) link against this
 *     library unchanged.  The reference header spells its types with macros
 *     (`int8`, `uint8`, `uint32`, vfgs_hw.h:40-47); the plain C types below
 *     are the same types.  State is a process-global singleton exactly like
 *     the reference's file-scope statics (vfgs_hw.c:49-68); all pointer
 *     arguments are borrowed for the duration of the call only.
 *
 * (2) EXTENSIONS (vfgs_add_grain_stripe, vfgs_hip_*): the throughput path.
 *     The reference has no stripe/frame entry; these are defined here with
 *     the contract "same samples and same seed registers afterwards as
 *     calling vfgs_add_grain_line for every line of the stripe, in order".
 *
 * There is no CPU fallback: every processing call runs HIP kernels and fails
 * loudly (message on stderr + abort for the void drop-in calls, nonzero return
 * for the vfgs_hip_* calls) when no gfx950 device/runtime is usable.
 */
#ifndef VFGS_HIP_H
#define VFGS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFGS_HIP_MAX_PATTERNS 8   /* vfgs_hw.h:49 */

/* ---- (1) drop-in hardware-layer interface -------------------------------- */

/* vfgs_hw.h:51 / vfgs_hw.c:314-318 -- P: 64x64 int8, row-major. index 0..7. */
void vfgs_set_luma_pattern(int index, signed char* P);
/* vfgs_hw.h:52 / vfgs_hw.c:320-325 -- copies 64/csuby rows of 64/csubx bytes, source
 * pitch 64/csuby (sic): depends on the subsampling set EARLIER. index 0..7. */
void vfgs_set_chroma_pattern(int index, signed char* P);
/* vfgs_hw.h:53 / vfgs_hw.c:327-331 -- 256 scale factors for component c (0=Y,1=Cb,2=Cr). */
void vfgs_set_scale_lut(int c, unsigned char lut[]);
/* vfgs_hw.h:54 / vfgs_hw.c:333-337 -- 256 pattern selectors; slot = lut[i] >> 4 (must be <= 8). */
void vfgs_set_pattern_lut(int c, unsigned char lut[]);
/* vfgs_hw.h:56 / vfgs_hw.c:339-344 -- loads seed<<1 into all four LFSR registers. */
void vfgs_set_seed(unsigned int seed);
/* vfgs_hw.h:57 / vfgs_hw.c:346-350 -- shift in 2..7. */
void vfgs_set_scale_shift(int shift);
/* vfgs_hw.h:58 / vfgs_hw.c:352-362 -- 8 or 10. */
void vfgs_set_depth(int depth);
/* vfgs_hw.h:59 / vfgs_hw.c:364-380 */
void vfgs_set_legal_range(int legal);
/* vfgs_hw.h:60 / vfgs_hw.c:382-388 -- each 1 or 2. */
void vfgs_set_chroma_subsampling(int subx, int suby);
/* vfgs_hw.h:62 / vfgs_hw.c:288-312 -- Y/U/V: HOST pointers to the start of line y in each
 * plane (U,V: chroma row y/csuby).  In place; complete when the call returns. */
void vfgs_add_grain_line(void* Y, void* U, void* V, int y, int width);

/* ---- (2) extensions ------------------------------------------------------- */

/* vfgs_add_grain_line and lines the caller has not handed over yet.  One drop-in call is one line in host memory and
 * must be complete on return (~90 us: copy in, launch, copy out).  The library therefore works ahead: on a miss it
 * starts computing the lines FOLLOWING the requested one as well -- the rest of the frame, in stripes of about 2 MB whose
 * upload, kernel and download are pipelined on three streams while the caller walks through the stripes already back --
 * keeps those results, and serves the next calls from them when they are exactly the predicted lines with unchanged
 * input bytes.  Lines not yet handed over are only
 * ever READ, and only inside rows the caller has proven to own: either this same buffer (same line-0 pointers and
 * width) has been walked top to bottom once before, or the caller has said so with vfgs_hip_declare_frame().  A first
 * frame, or a frame in a buffer the previous walk did not go through, is computed line by line.
 * vfgs_hip_line_lookahead(0) switches the behaviour off (so does the environment variable VFGS_HIP_LINE_LOOKAHEAD=0);
 * vfgs_hip_declare_frame(Y, U, V, width, height, stride, cstride) (host pointers to line 0, strides in samples)
 * promises that the three planes hold `height` lines at those pitches, which enables working ahead from the first
 * line of the first frame on; it stays valid for walks that start at these pointers; all-NULL revokes it.
 * For a binary that cannot be rebuilt, the environment variable VFGS_HIP_FRAME_HEIGHT=<lines> makes the same promise for
 * every walk that starts at line 0 (planes of at least that many lines at the pitches the first lines show): the first
 * frame then costs three line round trips instead of one per line (4320p: 0.36 s -> 12 ms inside the library; a short run of
 * the unchanged CLI, promised and plain runs interleaved on one box: never slower, 0.1 s faster at 4320p x 3 frames --
 * profiles/r06_cli_short_runs.log; round 5's "slower with the promise" was the order of its runs). */
void vfgs_hip_line_lookahead(int enable);
int vfgs_hip_declare_frame(const void* Y, const void* U, const void* V, unsigned width, unsigned height,
                           unsigned stride, unsigned cstride);

/* Host-memory stripe: lines y .. y+height-1, strides in samples.  Equivalent to `height`
 * consecutive vfgs_add_grain_line calls (one H2D + kernel + D2H instead of `height`). */
void vfgs_add_grain_stripe(void* Y, void* U, void* V, unsigned y, unsigned width,
                           unsigned height, unsigned stride, unsigned cstride);

/* Select the HIP device (default: the current device at first use) and create the
 * device-side state.  Returns 0 on success. */
int vfgs_hip_init(int device);
void vfgs_hip_shutdown(void);

/* Back to the reference's power-on state (vfgs_hw.c:49-63): zero banks and LUTs, seeds
 * 0xdeadbeef, 8-bit, 4:2:0, full range.  The reference has no such call because its state
 * only ever lives for one process; a long-lived library needs one. */
void vfgs_hip_reset_state(void);

/* Device-resident stripe / frame: dY/dU/dV are DEVICE pointers to the first line of the
 * stripe (dU/dV: chroma row y/csuby); 16-byte aligned, pitch*bytes_per_sample % 16 == 0,
 * stride >= 16*ceil(width/16) (the reference writes whole 16-sample blocks, SURVEY 8a
 * quirk 7).  Asynchronous on `stream` (a hipStream_t, may be NULL).  Returns 0 on success. */
int vfgs_hip_add_grain_stripe_dev(void* dY, void* dU, void* dV, unsigned y, unsigned width,
                                  unsigned height, unsigned stride, unsigned cstride, void* stream);
int vfgs_hip_add_grain_frame_dev(void* dY, void* dU, void* dV, unsigned width, unsigned height,
                                 unsigned stride, unsigned cstride, void* stream);

/* Multi-GPU stripe split (no collective): process only lines [part_y, part_y+part_height)
 * of a frame of `frame_height` lines -- pointers address line part_y -- but advance the
 * seed registers as if the whole frame had been processed, so that every rank stays in
 * lock step.  part_y must be a multiple of 16. */
int vfgs_hip_add_grain_frame_part_dev(void* dY, void* dU, void* dV, unsigned width,
                                      unsigned frame_height, unsigned part_y, unsigned part_height,
                                      unsigned stride, unsigned cstride, void* stream);

/* Batch: `nframes` equally shaped device-resident frames in ONE launch, processed as
 * consecutive frames (frame f+1 continues frame f's seed state).  Plane pointers of frame
 * f are dY + f*y_frame_pitch_bytes etc. */
int vfgs_hip_add_grain_frames_dev(void* dY, void* dU, void* dV, unsigned width, unsigned height,
                                  unsigned stride, unsigned cstride, unsigned nframes,
                                  uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes,
                                  void* stream);

/* Batch + stripe split combined: lines [part_y, part_y+part_height) of each of `nframes`
 * consecutive frames in ONE launch (what one rank of an N-GPU stripe split runs per step);
 * pointers address line part_y of frame 0; seeds advance as for whole frames. */
int vfgs_hip_add_grain_frames_part_dev(void* dY, void* dU, void* dV, unsigned width,
                                       unsigned frame_height, unsigned part_y, unsigned part_height,
                                       unsigned stride, unsigned cstride, unsigned nframes,
                                       uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes,
                                       void* stream);

/* Out-of-place form of the call above: reads sY/sU/sV, writes dY/dU/dV (same geometry and
 * pitches; src == dst is allowed and is what the in-place entry points pass).  This is how a
 * decoder uses film grain -- the clean picture stays a reference frame, the grained copy goes
 * to the display queue.  Same kernels, same speed within 2-3 % (profiles/r04_config_matrix*.jsonl:
 * 4320p 0.733 out of place / 0.716 in place of the 8 TB/s peak; as PURE streams the order is the
 * other way round on MI355X -- in-place nontemporal 6.6 TB/s, src->dst copy 5.0-5.3 TB/s,
 * BENCH_r03 / bench.py copy_ceilings_gbs). */
int vfgs_hip_add_grain_copy_dev(const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV,
                                unsigned width, unsigned frame_height, unsigned part_y, unsigned part_height,
                                unsigned stride, unsigned cstride, unsigned nframes,
                                uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes, void* stream);

/* The same with the output narrowed to 8 bit in the store: dst sample = (uint8)((v + 2) >> 2),
 * i.e. the reference CLI's `--outdepth 8` step yuv_to_8bit (yuv.c:216-258, called at
 * vfgs_main.c:787-788) fused into the kernel.  Needs vfgs_set_depth(10).  Source planes hold
 * uint16 samples (strides in samples), destination planes uint8 samples with their own strides
 * and frame pitches (multiples of 16).  Halves the write traffic of the path. */
int vfgs_hip_add_grain_copy8_dev(const void* sY, const void* sU, const void* sV, void* dY, void* dU, void* dV,
                                 unsigned width, unsigned frame_height, unsigned part_y, unsigned part_height,
                                 unsigned stride, unsigned cstride, unsigned dst_stride, unsigned dst_cstride,
                                 unsigned nframes, uint64_t y_frame_pitch_bytes, uint64_t c_frame_pitch_bytes,
                                 uint64_t dst_y_frame_pitch_bytes, uint64_t dst_c_frame_pitch_bytes, void* stream);

/* Batch of frames that are NOT equally spaced in memory (a decoder's pool of separately allocated frames; the reference's own
 * loop hands over one frame per call, vfgs_main.c:771-790): `nframes` equally shaped device-resident frames, frame f's planes at
 * frames[f].Y / .U / .V (device pointers to line 0, 16-byte aligned), processed as consecutive frames in ONE launch per 32
 * frames -- results and seed registers are those of `nframes` vfgs_hip_add_grain_frame_dev calls in list order.  The list itself
 * is host memory, read during the call only (the pointers travel in the kernel arguments: nothing to keep alive, no copy queued
 * in front of the launch).  A single frame per launch runs at 0.23 (1080p) / 0.49 (2160p) / 0.61 (4320p) of the HBM peak, the
 * same frames handed over 32 / 16 / 8 at a time at 0.70 / 0.76 / 0.72 (a launch costs 5 us of fill and drain).  All frames of
 * a call are in flight together, so the call refuses (error 18, nothing changed) destination planes that share bytes -- the
 * same plane listed twice, or two whose rows overlap -- and, in the _copy forms, a source plane that shares bytes with the
 * destination of ANOTHER frame (consecutive calls would read it grained or not, deterministically; one launch would not).
 * _copy: out of place, src[f] -> dst[f], same geometry (a plane of src[f] may BE the plane of dst[f]: in place);
 * _copy8: 10-bit source, 8-bit destination as vfgs_hip_add_grain_copy8_dev.  A refused call changes nothing. */
typedef struct vfgs_hip_frame_ptrs { void* Y; void* U; void* V; } vfgs_hip_frame_ptrs;
int vfgs_hip_add_grain_frame_list_dev(const vfgs_hip_frame_ptrs* frames, unsigned nframes, unsigned width, unsigned height,
                                      unsigned stride, unsigned cstride, void* stream);
/* ... lines [part_y, part_y + part_height) of every listed frame only (what one rank of a stripe split runs on its stripes of a
 * pool of frames): frames[f].Y / .U / .V address line part_y (chroma row part_y / csuby) of frame f, part_y a multiple of 16;
 * the seed registers advance as for whole frames of `frame_height` lines (vfgs_hip_add_grain_frames_part_dev). */
int vfgs_hip_add_grain_frame_list_part_dev(const vfgs_hip_frame_ptrs* frames, unsigned nframes, unsigned width,
                                           unsigned frame_height, unsigned part_y, unsigned part_height,
                                           unsigned stride, unsigned cstride, void* stream);
int vfgs_hip_add_grain_frame_list_copy_dev(const vfgs_hip_frame_ptrs* src, const vfgs_hip_frame_ptrs* dst, unsigned nframes,
                                           unsigned width, unsigned height, unsigned stride, unsigned cstride, void* stream);
int vfgs_hip_add_grain_frame_list_copy8_dev(const vfgs_hip_frame_ptrs* src, const vfgs_hip_frame_ptrs* dst, unsigned nframes,
                                            unsigned width, unsigned height, unsigned stride, unsigned cstride,
                                            unsigned dst_stride, unsigned dst_cstride, void* stream);

/* {rnd, rnd_up, line_rnd, line_rnd_up} as the reference would hold them (vfgs_hw.c:52-55). */
void vfgs_hip_get_seed_state(uint32_t out[4]);

/* The rest of the programmed state, for tests and debugging (host only, no GPU needed):
 * component c's scale and pattern LUTs as vfgs_hw.c:50-51 holds them (either pointer may be NULL), and
 * out[8] = { scale_shift as stored (vfgs_hw.c:56: shift + 6 - bs), bs, Y min, Y max, C min, C max, csubx, csuby }. */
int vfgs_hip_get_luts(int c, unsigned char scale[256], unsigned char pattern[256]);
void vfgs_hip_get_params(int out[8]);
/* ... and of the device copy of the LFSR bit stream (the reference steps its registers, vfgs_hw.c:74-79,309-310; here a window of the
 * stream lives in device memory): out[4] = { windows uploaded in a caller's stream (a bubble between two of its kernels), windows
 * built ahead on the library's copy stream, switches to a window built ahead, 32-bit words of the current window }. */
void vfgs_hip_get_stream_stats(uint64_t out[4]);
/* A batch of stripes (vfgs_hip_add_grain_frames_part_dev, vfgs_hip_add_grain_frame_list_part_dev: what one rank of a stripe split runs
 * per step) that covers at most about two thirds of its frames reads short runs of that stream, a whole frame's steps apart
 * (vfgs_hw.c:291-298,309-310: (ceil(height / 16) - 1) x ceil(width / 16) steps per frame).  The library builds those runs alone, each
 * from the one before by a jump -- the step of vfgs_hw.c:74-79 is linear over GF(2), n steps are one 32 x 32 bit matrix -- and
 * uploads the NEXT call's runs while this call's kernel works: out[4] = { images built in a caller's stream, images built ahead on
 * the copy stream, switches to an image built ahead, 1 if the most recent launch read such an image }.  The environment variable
 * VFGS_HIP_STRIPE_JUMP=0 keeps the contiguous window for those calls too (results are the same either way). */
void vfgs_hip_get_stripe_stream_stats(uint64_t out[4]);
/* The generator behind it, host only (no GPU needed): out[f * seg_words + k], f < nseg, k < seg_words = the 32-bit register
 * (vfgs_hw.c:74-79) after first_bit + f * step_bits + 32 * k steps from `reg` -- what `prng` would return after stepping
 * that far, without the stepping.  0 or an error code. */
int vfgs_hip_lfsr_segments(unsigned int reg, uint64_t first_bit, uint64_t step_bits, unsigned nseg, unsigned seg_words, uint32_t* out);

/* Last error of a vfgs_hip_* call (0 = none) and its text. */
int vfgs_hip_last_error(void);
const char* vfgs_hip_last_error_string(void);

/* Timing helper for harnesses: average device time in microseconds of the grain kernels
 * launched between begin/end on `stream` (hipEvent pair on that stream). */
int vfgs_hip_timer_begin(void* stream);
int vfgs_hip_timer_end(void* stream, float* elapsed_ms);

/* Overlap region for device-resident frames that arrive ONE PER CALL (the reference's call pattern, vfgs_main.c:771-790, with the
 * frames already in HBM): a kernel launch pays its fill and drain -- a 4320p frame runs at 0.60 of the HBM peak alone and at 0.72
 * when the next frame's launch overlaps its tail.  Between _begin(stream) and _end(stream) the caller promises that the
 * device-pointer calls it makes on `stream` are independent of each other (distinct frames) and of anything else queued on
 * `stream` after _begin; the library runs them alternately on two internal streams forked from `stream` at _begin and joined
 * back into it at _end (events only there, nothing between the launches).  Work queued on `stream` before _begin is complete
 * before the first call starts; work queued after _end sees every result.  Results and seed registers are those of the same
 * calls without the region.  One region at a time.  0 or an error code.  (The first region of a process creates the two
 * internal streams and their hardware queues: about 10 ms, once.) */
int vfgs_hip_overlap_begin(void* stream);
int vfgs_hip_overlap_end(void* stream);

/* Several devices in ONE process, for frames that live in host memory (SURVEY 8e x 8f row f3; the reference's frame loop and
 * file I/O, vfgs_main.c:664-682 / yuv.c:162-214, are one process and one thread).  After this call vfgs_add_grain_stripe and
 * vfgs_hip_add_grain_frames_host give every listed device a stripe of whole 16-line block rows of each frame and run the
 * devices concurrently (one PCIe link each, no exchange between them); results and seed registers are those of the
 * single-device call.  devices[0] is the library's device for everything else (the device-pointer entry points, the line
 * call).  A device may be listed more than once (testing on a one-GPU machine).  n = 1 returns to one device.
 * 0 or an error code. */
int vfgs_hip_init_devices(const int* devices, int n);

/* 1 if this library was built by a developer tool with tuning or ablation knobs (its output may be wrong by design),
 * 0 for the product build.  versatilefilmgrain_amd/build.py only ever builds the latter. */
int vfgs_hip_dev_build(void);

/* SURVEY 8f row f3 -- the data path around the reference's yuv_read / yuv_write (yuv.c:162-214): nframes frames that live
 * in HOST memory, processed in place and pipelined through a ring of three device frames: the upload of frame i, the
 * kernel of frame i-1 and the download of frame i-2 are queued on three streams.  Y/U/V: arrays of nframes plane pointers
 * (any host memory; frames need not be contiguous), all frames with the same geometry, strides in samples as in
 * vfgs_add_grain_stripe.  Only the bytes the reference touches (whole 16-sample blocks of every row) travel; stride
 * padding is never written.  Pinned memory (vfgs_hip_host_alloc) makes the copies asynchronous; with pageable memory the
 * call is correct but the runtime stages every copy and the calling thread blocks for it.  Returns when all frames are
 * back in host memory; the seed registers have advanced as after nframes whole-frame calls.  0 or an error code. */
int vfgs_hip_add_grain_frames_host(void* const* Y, void* const* U, void* const* V, unsigned nframes, unsigned width,
                                   unsigned height, unsigned stride, unsigned cstride);
void* vfgs_hip_host_alloc(uint64_t bytes);   /* pinned host memory (hipHostMalloc); NULL on failure */
void vfgs_hip_host_free(void* p);

/* Introspection for benchmarks/tests. */
int vfgs_hip_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* name, int name_len);

/* What the most recent grain launch of this process (primary device) actually dispatched -- so that a benchmark labels
 * its numbers with the kernel that ran instead of the kernel it expects.  EVERY grain launch is recorded and counted in
 * `launches`, also those the host-memory entry points make on the library's own staging buffers (`internal` = 1: a line of
 * vfgs_add_grain_line, the stripes it computes ahead -- including ones that are dropped later --, vfgs_add_grain_stripe,
 * vfgs_hip_add_grain_frames_host): after a walk through the line call the record describes such a stripe launch.  `kernel` is the instantiation's name as the
 * profiler prints it (e.g. "grain_rw_kernel<10,2,2,false,false,true,false,false>": depth, chroma subsampling
 * x / y, 8-bit destination of a 10-bit path, luma one-pattern form, chroma one-pattern form, rows walked in parts,
 * persistent luma workgroups).  Returns 0, or -1 when nothing has been launched yet. */
typedef struct vfgs_hip_launch_info {
	int depth, csubx, csuby;          /* sample depth and chroma format the launch was compiled for */
	int out8;                         /* 1: 10-bit source narrowed to 8 bit in the store */
	int one_y, one_c;                 /* 1: one-pattern form of the luma / chroma table image */
	int in_place;                     /* 1: destination planes == source planes */
	int nframes;                      /* frames of the launch */
	int workgroups_per_frame;         /* luma + 2 x chroma */
	int frames_per_front;             /* frames swept at the same time (1 or 2) */
	int rows_per_wave[2];             /* luma, chroma: rows of one block row a wave walks */
	int positions_per_row[2];         /* luma, chroma: 1 KiB wave accesses per row */
	int parts_per_row;                /* passes over the block-parameter table a row needs (1 up to 512 blocks = 8192 samples per row) */
	int persistent_luma_workgroups;   /* 0, or the number of luma workgroups that share the launch's luma tasks (general-form luma of small pictures) */
	int waves_per_workgroup;
	int lds_bytes_per_workgroup;      /* LDS the kernel allocates: table image + block parameters (the 10-bit all-one-pattern kernels: padded to 40,960, four workgroups per CU) */
	unsigned long long launches;      /* grain launches of this process so far */
	char kernel[96];
	int listed;                       /* 1: the frames' plane pointers came as a list (vfgs_hip_add_grain_frame_list_*) */
	int internal;                     /* 1: launched by a host-memory entry point on the library's staging buffers, not on a caller's device planes */
} vfgs_hip_launch_info;
int vfgs_hip_last_launch_info(vfgs_hip_launch_info* out);

#ifdef __cplusplus
}
#endif

#endif /* VFGS_HIP_H */
