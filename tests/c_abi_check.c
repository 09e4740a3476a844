/* Compiled and linked (not run) by tests/test_library_cpu.py with a plain C compiler: proves that
 * include/vfgs_hip.h is valid C and that every declared entry point resolves against
 * libvfgs_hip.so with C linkage -- the way the reference's vfgs_fw.c / vfgs_main.c use it. */
#include "vfgs_hip.h"
#include "vfgs_hip_fw.h"

#include <stdio.h>

int main(void)
{
	/* take the address of every entry point so the linker must resolve it */
	void (*fn[])(void) = {
		(void (*)(void))vfgs_set_luma_pattern, (void (*)(void))vfgs_set_chroma_pattern, (void (*)(void))vfgs_set_scale_lut, (void (*)(void))vfgs_set_pattern_lut,
		(void (*)(void))vfgs_set_seed, (void (*)(void))vfgs_set_scale_shift, (void (*)(void))vfgs_set_depth, (void (*)(void))vfgs_set_legal_range,
		(void (*)(void))vfgs_set_chroma_subsampling, (void (*)(void))vfgs_add_grain_line, (void (*)(void))vfgs_add_grain_stripe,
		(void (*)(void))vfgs_hip_init, (void (*)(void))vfgs_hip_shutdown, (void (*)(void))vfgs_hip_reset_state,
		(void (*)(void))vfgs_hip_add_grain_stripe_dev, (void (*)(void))vfgs_hip_add_grain_frame_dev, (void (*)(void))vfgs_hip_add_grain_frame_part_dev,
		(void (*)(void))vfgs_hip_add_grain_frames_dev, (void (*)(void))vfgs_hip_add_grain_frames_part_dev, (void (*)(void))vfgs_hip_add_grain_copy_dev,
		(void (*)(void))vfgs_hip_add_grain_copy8_dev, (void (*)(void))vfgs_hip_add_grain_frame_list_dev, (void (*)(void))vfgs_hip_add_grain_frame_list_part_dev, (void (*)(void))vfgs_hip_add_grain_frame_list_copy_dev,
		(void (*)(void))vfgs_hip_add_grain_frame_list_copy8_dev, (void (*)(void))vfgs_hip_get_seed_state, (void (*)(void))vfgs_hip_get_luts, (void (*)(void))vfgs_hip_get_params, (void (*)(void))vfgs_hip_last_error,
		(void (*)(void))vfgs_hip_last_error_string, (void (*)(void))vfgs_hip_timer_begin, (void (*)(void))vfgs_hip_timer_end, (void (*)(void))vfgs_hip_device_info,
		(void (*)(void))vfgs_hip_dev_build, (void (*)(void))vfgs_hip_init_devices, (void (*)(void))vfgs_hip_overlap_begin, (void (*)(void))vfgs_hip_overlap_end, (void (*)(void))vfgs_hip_get_stream_stats, (void (*)(void))vfgs_hip_get_stripe_stream_stats, (void (*)(void))vfgs_hip_lfsr_segments, (void (*)(void))vfgs_hip_line_lookahead, (void (*)(void))vfgs_hip_declare_frame,
		(void (*)(void))vfgs_hip_add_grain_frames_host, (void (*)(void))vfgs_hip_host_alloc, (void (*)(void))vfgs_hip_host_free, (void (*)(void))vfgs_hip_last_launch_info,
		/* firmware interface, vfgs_fw.h:91-92, and its extensions */
		(void (*)(void))vfgs_init_sei, (void (*)(void))vfgs_init_afgs1, (void (*)(void))vfgs_hip_generate_patterns, (void (*)(void))vfgs_hip_get_pattern,
		/* configuration files */
		(void (*)(void))vfgs_hip_cfg_defaults, (void (*)(void))vfgs_hip_cfg_read, (void (*)(void))vfgs_hip_cfg_check,
		(void (*)(void))vfgs_hip_cfg_adjust_chroma, (void (*)(void))vfgs_hip_cfg_apply_gain, (void (*)(void))vfgs_hip_cfg_program,
	};
	unsigned char lut[256] = {0};
	signed char pat[64 * 64] = {0};
	unsigned seeds[4];
	/* host-only calls work without a GPU: exactly the firmware's call pattern (vfgs_fw.c:585-643) */
	vfgs_set_depth(10);
	vfgs_set_chroma_subsampling(2, 2);
	vfgs_set_luma_pattern(0, pat);
	vfgs_set_chroma_pattern(0, pat);
	vfgs_set_scale_lut(0, lut);
	vfgs_set_pattern_lut(0, lut);
	vfgs_set_scale_shift(5);
	vfgs_set_legal_range(0);
	vfgs_set_seed(12345u);
	vfgs_hip_get_seed_state(seeds);
	printf("%u entry points, seed register 0x%08x\n", (unsigned)(sizeof fn / sizeof fn[0]), seeds[0]);
	return seeds[0] == (12345u << 1) ? 0 : 1;
}
