/*
 * libvfgs_hip -- firmware layer on the GPU.
 *
 * Drop-in for the reference's firmware interface, /root/reference/src/vfgs_fw.h:49-92: the
 * two parameter structures (same members, same order, same C layout) and the two entry points
 *
 *     void vfgs_init_sei(fgs_sei* cfg);        vfgs_fw.h:91, vfgs_fw.c:517-644
 *     void vfgs_init_afgs1(fgs_afgs1* cfg);    vfgs_fw.h:92, vfgs_fw.c:663-708
 *
 * Like the reference's, they program the hardware layer (vfgs_hip.h) through its setters; the
 * difference is where the grain patterns are made.  The reference builds every pattern on the
 * CPU (64x64 / 32x32 integer inverse DCT of band-limited Gaussian noise, vfgs_fw.c:297-408, or
 * a causal auto-regressive filter over 82x73 / 44x38 samples, vfgs_fw.c:410-502) and copies it
 * into the hardware layer.  Here the host only derives the small tables (scale / pattern LUTs,
 * shifts, seed) and queues one kernel launch that generates all patterns of the configuration
 * directly in the device-resident pattern banks: no pattern byte crosses PCIe, and a
 * per-frame configuration switch (`-c <poc>:file`, vfgs_main.c:773-781) does not stall the
 * grain kernels that are still queued.
 *
 * A program that also links the reference's vfgs_fw.c keeps using that one (symbols of the
 * executable win); a program that links only vfgs_main.c + yuv.c gets these.
 *
 * Errors: none returned, as in the reference; violations that the reference asserts on
 * (vfgs_fw.c:459 unsupported coefficient count, :656 non-increasing scaling points) abort with
 * a message, HIP failures abort too.  vfgs_hip_last_error_string() has the text.
 */
#ifndef VFGS_HIP_FW_H_
#define VFGS_HIP_FW_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEI_MAX_MODEL_VALUES 6   /* vfgs_fw.h:49 */

/* FGC SEI message (H.274), vfgs_fw.h:51-60 */
typedef struct fgs_sei_s {
	uint8_t model_id;                               /* 0: frequency filtering, 1: auto-regressive */
	uint8_t log2_scale_factor;
	uint8_t comp_model_present_flag[3];
	uint16_t num_intensity_intervals[3];
	uint8_t num_model_values[3];
	uint8_t intensity_interval_lower_bound[3][256];
	uint8_t intensity_interval_upper_bound[3][256];
	int16_t comp_model_value[3][256][SEI_MAX_MODEL_VALUES];
} fgs_sei;

/* AOM film grain metadata (AFGS1), vfgs_fw.h:62-89 */
typedef struct fgs_afgs1_s {
	uint16_t grain_seed;
	uint8_t num_y_points;            /* 0..14 */
	uint8_t point_y_values[14];
	uint8_t point_y_scaling[14];
	uint8_t chroma_scaling_from_luma;
	uint8_t num_cb_points;           /* 0..10 */
	uint8_t point_cb_values[10];
	uint8_t point_cb_scaling[10];
	uint8_t num_cr_points;           /* 0..10 */
	uint8_t point_cr_values[10];
	uint8_t point_cr_scaling[10];
	uint8_t grain_scaling;           /* 8..11 */
	uint8_t ar_coeff_lag;            /* 0..3 */
	int16_t ar_coeffs_y[24];
	int16_t ar_coeffs_cb[25];
	int16_t ar_coeffs_cr[25];
	uint8_t ar_coeff_shift;          /* 6..9 */
	uint8_t grain_scale_shift;       /* 0..3 */
	uint8_t cb_mult;
	uint8_t cb_luma_mult;
	uint16_t cb_offset;
	uint8_t cr_mult;
	uint8_t cr_luma_mult;
	uint16_t cr_offset;
	uint8_t overlap_flag;
	uint8_t clip_to_restricted_range;
} fgs_afgs1;

/* ---- drop-in entry points (vfgs_fw.h:91-92) ------------------------------------------- */
void vfgs_init_sei(fgs_sei* cfg);
void vfgs_init_afgs1(fgs_afgs1* cfg);

/* ---- extensions ---------------------------------------------------------------------- */

/* One grain pattern to be generated on the device into pattern slot `index`. */
typedef struct vfgs_hip_pattern_job {
	int32_t kind;       /* 0: frequency filtered (vfgs_fw.c:362-408), 1: auto-regressive (vfgs_fw.c:410-502) */
	int32_t chroma;     /* 0: 64x64 luma pattern, noise seed Seed_LUT[0]; 1: 32x32 chroma pattern */
	int32_t index;      /* pattern slot 0..7, as vfgs_set_luma_pattern / vfgs_set_chroma_pattern */
	int32_t seed_index; /* which entry of the seed table starts the noise generator (vfgs_fw.c:369,392,620,...) */
	int32_t fh, fv;     /* kind 0: horizontal / vertical cut-off, comp_model_value[1], [2] (2..14) */
	int32_t scale;      /* kind 1: right shift of the filter sum (log2_scale_factor or ar_coeff_shift), 1..15 */
	int32_t shift;      /* kind 1: right shift of the Gaussian sample, 1..7 */
	int16_t coef[28];   /* kind 1: the 4x7 causal filter taps exactly as vfgs_fw.c:412,424-462 lays them out */
} vfgs_hip_pattern_job;

/* Queue the generation of `n` (<= 16) patterns.  Asynchronous; ordered before every later
 * vfgs_add_grain_* / vfgs_hip_add_grain_* call.  Luma jobs come first: a chroma job on a
 * non-4:2:0 layout inherits the tail of the last luma pattern of the same call, as the
 * reference's shared 64x64 scratch buffer does (vfgs_fw.c:519,603-623 with vfgs_hw.c:320-325).
 * Returns 0 or an error code. */
int vfgs_hip_generate_patterns(const vfgs_hip_pattern_job* jobs, int n);

/* Read back pattern slot `index` of bank `chroma` (0 luma, 1 chroma) as the hardware layer
 * holds it ([64][64], vfgs_hw.c:49).  Synchronises; for tests and debugging. */
int vfgs_hip_get_pattern(int chroma, int index, signed char out[64 * 64]);

/* ---- configuration files ----------------------------------------
 * What the reference CLI does between `-c <file>` and vfgs_init_* (vfgs_main.c:126-195 value
 * readers, :208-232 chroma adjustment, :234-303 checks, :309-434 AFGS1 "grain table" syntax,
 * :436-559 cfg and SEI-dump syntaxes, :561-593 gain), as library calls, so that a host without
 * the reference's vfgs_main.c can go from a configuration file to a programmed device.  Host
 * only; no GPU needed until vfgs_hip_cfg_program().
 *
 * The state is the pair of parameter sets the CLI keeps (vfgs_main.c:69-124).  Reading a file
 * updates it IN PLACE -- fields a file does not mention keep their previous values, exactly as
 * in the CLI, where every file is read on top of the defaults / the previous configuration.
 * AFGS1 is active when afgs1.num_y_points != 0 (vfgs_main.c:297-303). */
typedef struct vfgs_hip_cfg {
	fgs_sei sei;
	fgs_afgs1 afgs1;
} vfgs_hip_cfg;

/* the CLI's built-in SEI (vfgs_main.c:69-120), AFGS1 off */
void vfgs_hip_cfg_defaults(vfgs_hip_cfg* cfg);
/* read one file (any of the three syntaxes); 0, or 1 with the reference's message in vfgs_hip_last_error_string() */
int vfgs_hip_cfg_read(vfgs_hip_cfg* cfg, const char* filename);
/* the CLI's acceptance checks for picture format 420 / 422 / 444 and bit depth 8 / 10; 0 or 1 + message */
int vfgs_hip_cfg_check(const vfgs_hip_cfg* cfg, int format, int depth);
/* SEI frequency cut-offs and scale of the chroma components for subsampled formats (vfgs_main.c:208-232) */
void vfgs_hip_cfg_adjust_chroma(vfgs_hip_cfg* cfg, int format);
/* global strength in percent (vfgs_main.c:561-593); 100 = unchanged */
void vfgs_hip_cfg_apply_gain(vfgs_hip_cfg* cfg, unsigned gain);
/* vfgs_init_afgs1(&cfg->afgs1) or vfgs_init_sei(&cfg->sei), whichever is active (vfgs_main.c:757-760) */
void vfgs_hip_cfg_program(vfgs_hip_cfg* cfg);

#ifdef __cplusplus
}
#endif

#endif /* VFGS_HIP_FW_H_ */
