// Firmware layer, host part: turns an FGC SEI message or an AFGS1 parameter set into what the
// hardware layer needs -- the two entry points of the reference's vfgs_fw.h:91-92.
//
// Only the small tables are computed here (scale / pattern LUTs, shifts, seed, the list of
// distinct patterns); the patterns themselves are generated on the GPU by
// vfgs_hip_generate_patterns (vfgs_fw_kernel.hip).  Like the reference's firmware this file is a
// pure client of the hardware-layer interface; it keeps no state of its own.
//
// The derivations follow the reference line by line where the result depends on it, including
// its oddities (each is marked QUIRK), because parity is judged on the bytes the hardware layer
// ends up with (tests/golden/traces).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/vfgs_hip.h"
#include "../../include/vfgs_hip_fw.h"

namespace {

constexpr int kMaxPatterns = 8;   // VFGS_MAX_PATTERNS, vfgs_hw.h:49

[[noreturn]] void fw_die(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	fprintf(stderr, "libvfgs_hip: fatal: ");
	vfprintf(stderr, fmt, ap);
	fprintf(stderr, "\n");
	va_end(ap);
	abort();   // the reference asserts (vfgs_fw.c:459, :656)
}

void generate(const vfgs_hip_pattern_job* jobs, int n)
{
	if (n && vfgs_hip_generate_patterns(jobs, n))
		fw_die("pattern generation failed: %s", vfgs_hip_last_error_string());
}

// ---- SEI -------------------------------------------------------------------------------

// The list of distinct patterns of one bank.  An entry is the position of an interval's model
// values in comp_model_value viewed as one flat int16 array (vfgs_fw.c:545: 6*(k + 256*c)).
struct PatternList {
	int n = 0;
	bool used[kMaxPatterns] = {};
	int flat[kMaxPatterns] = {};
	unsigned char lower[kMaxPatterns] = {};   // lower intensity bound the entry was created from (sort key)
};

// Values 1..5 of an entry, i.e. everything but the scale factor (vfgs_fw.c:504-515).
// QUIRK: the reference initialises unused entries to ~0 and still compares against them; with
// its pointer arithmetic an unused entry reads elements -1+1 .. -1+5 of the flat array, that is
// the FIRST five numbers of the luma component's first interval.  Reproduced.
const int16_t* entry_tail(const fgs_sei* cfg, const PatternList& pl, int i)
{
	const int16_t* base = &cfg->comp_model_value[0][0][0];
	return pl.used[i] ? base + pl.flat[i] + 1 : base;
}

int find_pattern(const fgs_sei* cfg, const PatternList& pl, int flat)
{
	const int16_t* want = &cfg->comp_model_value[0][0][0] + flat + 1;
	for (int i = 0; i < kMaxPatterns; i++)
		if (!memcmp(entry_tail(cfg, pl, i), want, (SEI_MAX_MODEL_VALUES - 1) * sizeof(int16_t)))
			return i;
	return kMaxPatterns;
}

void collect_patterns(const fgs_sei* cfg, int c, PatternList& pl)
{
	// vfgs_fw.c:540-572: new parameter sets join the list, which stays sorted by the lower bound
	for (int k = 0; k < cfg->num_intensity_intervals[c]; k++)
	{
		const int flat = SEI_MAX_MODEL_VALUES * (k + 256 * c);
		if (find_pattern(cfg, pl, flat) != kMaxPatterns || pl.n >= kMaxPatterns)
			continue;
		const unsigned char a = cfg->intensity_interval_lower_bound[c][k];
		int at = pl.n;
		while (at > 0 && pl.lower[at - 1] > a)
		{
			pl.lower[at] = pl.lower[at - 1];
			pl.flat[at] = pl.flat[at - 1];
			at--;
		}
		pl.lower[at] = a;
		pl.flat[at] = flat;
		pl.used[pl.n] = true;   // entries 0..n are the used ones; the sort only moves values
		pl.n++;
	}
}

// SEI auto-regressive model: the six model values become a 4x7 tap matrix (vfgs_fw.c:427-436)
void sei_ar_taps(const int16_t* v, int scale, int16_t coef[28])
{
	memset(coef, 0, 28 * sizeof(int16_t));
	auto at = [&](int row, int col) -> int16_t& { return coef[row * 7 + col]; };
	at(3, 2) = v[1];                                           // left
	at(2, 3) = (int16_t)((v[1] * v[4]) >> scale);              // top
	at(2, 2) = at(2, 4) = (int16_t)((v[3] * v[4]) >> scale);   // top-left, top-right
	at(3, 1) = v[5];                                           // two to the left
	at(1, 3) = (int16_t)((int32_t)((uint32_t)v[5] * (uint32_t)v[4] * (uint32_t)v[4]) >> (2 * scale));   // two up
}

void sei_jobs(const fgs_sei* cfg, const PatternList& pl, int chroma, vfgs_hip_pattern_job* jobs, int& n)
{
	for (int i = 0; i < pl.n; i++)
	{
		const int16_t* v = &cfg->comp_model_value[0][0][0] + pl.flat[i];
		vfgs_hip_pattern_job& j = jobs[n++];
		memset(&j, 0, sizeof j);
		j.chroma = chroma;
		j.index = i;
		j.seed_index = chroma;                 // vfgs_fw.c:369, :392, :606, :613
		if (cfg->model_id)
		{
			j.kind = 1;
			j.scale = cfg->log2_scale_factor;  // vfgs_fw.c:606: shift 1, scale log2_scale_factor
			j.shift = 1;
			sei_ar_taps(v, j.scale, j.coef);
		}
		else
		{
			j.kind = 0;
			j.fh = v[1];
			j.fv = v[2];
		}
	}
}

void sei_luts(const fgs_sei* cfg, const PatternList& pl, int c, unsigned char slut[256])
{
	unsigned char plut[256];
	if (cfg->comp_model_present_flag[c])
	{
		memset(plut, 255, sizeof plut);
		for (int k = 0; k < cfg->num_intensity_intervals[c]; k++)
		{
			const int a = cfg->intensity_interval_lower_bound[c][k];
			const int b = cfg->intensity_interval_upper_bound[c][k];
			const int i = find_pattern(cfg, pl, SEI_MAX_MODEL_VALUES * (k + 256 * c));
			for (int l = a; l <= b; l++)
			{
				slut[l] = (unsigned char)cfg->comp_model_value[c][k][0];
				if (i < kMaxPatterns)
					plut[l] = (unsigned char)(i << 4);
			}
		}
		// intensities outside every interval repeat the pattern below them (vfgs_fw.c:625-633)
		unsigned char last = 0;
		for (int k = 0; k < 256; k++)
		{
			if (plut[k] == 255) plut[k] = last;
			else last = plut[k];
		}
	}
	else
		memset(plut, 0, sizeof plut);
	vfgs_set_scale_lut(c, slut);
	vfgs_set_pattern_lut(c, plut);
}

// ---- AFGS1 -----------------------------------------------------------------------------

// AFGS1 scaling function -> scale LUT: zero outside the points, between two points the rounded interpolation of
// vfgs_fw.c:649-661 (integer division that truncates towards zero, also for falling segments).  Segments are half-open
// [x[s], x[s+1]): the reference writes every segment including its end point and the next segment then rewrites that
// entry with its own start value -- the same bytes; only the last segment keeps its end point.
void scaling_lut(unsigned char lut[256], const unsigned char* x, const unsigned char* y, int npoints)
{
	memset(lut, 0, 256);
	for (int s = 0; s + 1 < npoints; s++)
	{
		const int x0 = x[s], span = x[s + 1] - x0, rise = (int)y[s + 1] - (int)y[s];
		if (span <= 0)
			fw_die("AFGS1 scaling points must be in increasing order (vfgs_fw.c:656)");
		const int stop = x0 + span + (s + 2 == npoints ? 1 : 0);
		for (int t = x0; t < stop; t++)
			lut[t] = (unsigned char)(y[s] + (rise * (t - x0) + span / 2) / span);
	}
}

// AV1 order of the causal neighbourhood -> 4x7 tap matrix (vfgs_fw.c:461-464)
void afgs1_taps(const int16_t* ar, int lag, int16_t coef[28])
{
	memset(coef, 0, 28 * sizeof(int16_t));
	int k = 0;
	for (int j = -lag; j <= 0; j++)
		for (int i = -lag; i <= lag && (i < 0 || j < 0); i++)
			coef[(3 + j) * 7 + 3 + i] = ar[k++];
}

}  // namespace

extern "C" {

void vfgs_init_sei(fgs_sei* cfg)
{
	vfgs_hip_pattern_job jobs[2 * kMaxPatterns];
	int njobs = 0;
	PatternList luma, chroma;

	if (cfg->comp_model_present_flag[0])
		collect_patterns(cfg, 0, luma);
	sei_jobs(cfg, luma, 0, jobs, njobs);
	{
		unsigned char slut[256] = {0};
		sei_luts(cfg, luma, 0, slut);
	}
	// Cb and Cr share one bank and one list (vfgs_fw.c:533-538: the list is reset for c = 0 and 1 only)
	for (int c = 1; c < 3; c++)
		if (cfg->comp_model_present_flag[c])
			collect_patterns(cfg, c, chroma);
	sei_jobs(cfg, chroma, 1, jobs, njobs);
	{
		// QUIRK: one scale table is cleared before Cb and NOT between Cb and Cr (vfgs_fw.c:530,
		// :598-643), so Cr inherits Cb's scale wherever its own intervals leave a gap
		unsigned char slut[256] = {0};
		sei_luts(cfg, chroma, 1, slut);
		sei_luts(cfg, chroma, 2, slut);
	}
	generate(jobs, njobs);
	vfgs_set_scale_shift(cfg->log2_scale_factor - (cfg->model_id ? 1 : 0));   // vfgs_fw.c:644
}

void vfgs_init_afgs1(fgs_afgs1* cfg)
{
	unsigned char lut[256];
	vfgs_hip_pattern_job jobs[3];
	const int lag = cfg->ar_coeff_lag;

	vfgs_set_seed(cfg->grain_seed | ((uint32_t)cfg->grain_seed << 16));   // vfgs_fw.c:672

	scaling_lut(lut, cfg->point_y_values, cfg->point_y_scaling, cfg->num_y_points);
	vfgs_set_scale_lut(0, lut);
	if (!cfg->chroma_scaling_from_luma)
		scaling_lut(lut, cfg->point_cb_values, cfg->point_cb_scaling, cfg->num_cb_points);
	vfgs_set_scale_lut(1, lut);
	if (!cfg->chroma_scaling_from_luma)
		scaling_lut(lut, cfg->point_cr_values, cfg->point_cr_scaling, cfg->num_cr_points);
	vfgs_set_scale_lut(2, lut);

	if (lag < 1 || lag > 3)
		fw_die("AFGS1 ar_coeff_lag %d: the reference supports 1..3 (vfgs_fw.c:438-459)", lag);
	// QUIRK: all three filters take 2*lag*(lag+1) taps (vfgs_fw.c:687-697); the extra chroma
	// coefficient (luma injection) is never used
	const int16_t* ar[3] = {cfg->ar_coeffs_y, cfg->ar_coeffs_cb, cfg->ar_coeffs_cr};
	for (int c = 0; c < 3; c++)
	{
		vfgs_hip_pattern_job& j = jobs[c];
		memset(&j, 0, sizeof j);
		j.kind = 1;
		j.chroma = c > 0;
		j.index = c == 2 ? 1 : 0;              // Cb -> chroma slot 0, Cr -> chroma slot 1
		j.seed_index = c;                      // vfgs_fw.c:689, :693, :697
		j.scale = cfg->ar_coeff_shift;
		j.shift = cfg->grain_scale_shift + 1;  // vfgs_fw.c:683-686
		afgs1_taps(ar[c], lag, j.coef);
	}
	generate(jobs, 3);

	memset(lut, 0, sizeof lut);
	vfgs_set_pattern_lut(0, lut);
	vfgs_set_pattern_lut(1, lut);
	// QUIRK: the reference fills Cr's pattern table with 1, and the hardware layer takes the slot
	// from bits 7:4 (vfgs_hw.c:212): Cr therefore uses chroma slot 0, Cb's pattern
	memset(lut, 1, sizeof lut);
	vfgs_set_pattern_lut(2, lut);

	vfgs_set_scale_shift(cfg->grain_scaling - 6);
	vfgs_set_legal_range(cfg->clip_to_restricted_range);
}

}  // extern "C"
