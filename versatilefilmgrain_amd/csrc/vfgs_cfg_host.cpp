// Configuration files -> parameter sets (include/vfgs_hip_fw.h, "configuration files").
//
// Restates what the reference CLI does between `-c <file>` and vfgs_init_* : the value readers
// (vfgs_main.c:126-195), the three file syntaxes (:309-434 AFGS1 grain table, :436-559 encoder
// cfg and SEI dump), the acceptance checks (:234-303), the chroma adjustment (:208-232) and the
// gain (:561-593).  Host only.  The result must be byte-identical to the structures the CLI
// hands to its firmware (tests/golden/fwcfg), so number parsing deliberately goes through the
// same libc calls (atoi, isblank, strcasecmp) and the same integer types.
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <string>

#include "../../include/vfgs_hip.h"
#include "../../include/vfgs_hip_fw.h"

namespace vfgs {
int set_error(int code, const char* msg);
}

namespace {

constexpr int kDefaultFreq = 8;   // DEFAULT_FREQ, vfgs_main.c:52

int reject(const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	vfgs::set_error(1, buf);
	return 1;   // the reference's CHECK (vfgs_main.c:53) prints "Error: <message>" and returns 1
}

bool numeric(char c, bool with_sign) { return isdigit((unsigned char)c) || (with_sign && (c == '-' || c == '+')); }

// one number off the front of *s, then past its characters and the blanks after them
int take(const char*& s, bool with_sign)
{
	const int v = atoi(s);
	while (numeric(*s, with_sign)) s++;
	while (isblank((unsigned char)*s)) s++;
	return v;
}

// vfgs_main.c:143-154 / :130-141; `cap` keeps a long line inside the member (the reference does not look)
void list_u8(uint8_t* x, int cap, const char* s)
{
	for (int n = 0; isdigit((unsigned char)*s) && n < cap; n++) x[n] = (uint8_t)take(s, false);
}

void list_i16(int16_t* x, int cap, const char* s)
{
	for (int n = 0; numeric(*s, true) && n < cap; n++) x[n] = (int16_t)take(s, true);
}

// model values a message leaves out (vfgs_main.c:156-165): x[0..n) are given
void complete_model(int16_t* x, int n, const fgs_sei& sei)
{
	const int ar = sei.model_id;
	int p = n;   // next unspecified value (for n = 0 the defaults start at the scale factor, as in the reference)
	if (n < 2) x[p++] = ar ? 0 : kDefaultFreq;                       // horizontal cut-off / first AR coefficient
	if (n < 3) { x[p] = ar ? 0 : x[p - 1]; p++; }                    // vertical cut-off = horizontal
	if (n < 4) x[p++] = 0;
	if (n < 5) x[p++] = (int16_t)(ar << sei.log2_scale_factor);
	if (n < 6) x[p++] = 0;
}

// "v v v  v v v ..." : n given values per intensity interval (vfgs_main.c:167-189)
void list_models(int16_t (*x)[SEI_MAX_MODEL_VALUES], const char* s, const fgs_sei& sei, int c)
{
	const int n = sei.num_model_values[c];
	for (int k = 0; numeric(*s, true) && k < 256; k++)
	{
		for (int i = 0; i < n; i++)
		{
			const int v = take(s, true);
			if (i < SEI_MAX_MODEL_VALUES) x[k][i] = (int16_t)v;
		}
		if (n >= 0 && n <= SEI_MAX_MODEL_VALUES) complete_model(x[k], n, sei);
	}
}

// ---- AFGS1 grain table (vfgs_main.c:309-434): six or eight whitespace separated lines ------

struct Words {
	char line[1024];
	char* next = nullptr;
	bool read(FILE* f)
	{
		line[0] = 0;
		if (!fgets(line, sizeof line, f)) line[0] = 0;
		next = line;
		return true;
	}
	const char* word()
	{
		char* w = strtok(next, " \t");
		next = nullptr;
		return w;
	}
};

#define NEED(cond, ...) do { if (!(cond)) return reject(__VA_ARGS__); } while (0)

int read_grain_table(FILE* f, fgs_afgs1& a)
{
	Words w;
	const char* s;
	auto number = [&](const char* what, int& v) -> int {
		s = w.word();
		if (!s) return reject("AFGS1 table entry: missing %s", what);
		v = atoi(s);
		return 0;
	};
	int v = 0;

	w.read(f);
	s = w.word();
	NEED(s && !strcmp(s, "E"), "AFGS1 table entry: expecting header (E)");
	w.word(); w.word(); w.word();          // start time, end time, apply_grain: not used
	if (number("grain_seed", v)) return 1;
	a.grain_seed = (uint16_t)v;

	w.read(f);
	s = w.word();
	NEED(s && !strcmp(s, "p"), "AFGS1 table entry: expecting parameters (p)");
	if (number("ar_coeff_lag", v)) return 1;
	a.ar_coeff_lag = (uint8_t)v;             NEED(a.ar_coeff_lag <= 3, "ar_coeff_lag higher than 3");
	if (number("ar_coeff_shift", v)) return 1;
	a.ar_coeff_shift = (uint8_t)v;           NEED(a.ar_coeff_shift >= 6 && a.ar_coeff_shift <= 9, "ar_coeff_shift out of 6..9 range");
	if (number("grain_scale_shift", v)) return 1;
	a.grain_scale_shift = (uint8_t)v;        NEED(a.grain_scale_shift <= 3, "grain_scale_shift higher than 3");
	if (number("grain_scaling", v)) return 1;
	a.grain_scaling = (uint8_t)v;            NEED(a.grain_scaling >= 8 && a.grain_scaling <= 11, "grain_scaling out of 8..11 range");
	if (number("chroma_scaling_from_luma", v)) return 1;
	a.chroma_scaling_from_luma = (uint8_t)v;
	if (number("overlap_flag", v)) return 1;
	a.overlap_flag = (uint8_t)v;
	if (number("cb_mult", v)) return 1;
	a.cb_mult = (uint8_t)v;
	if (number("cb_luma_mult", v)) return 1;
	a.cb_luma_mult = (uint8_t)v;
	if (number("cb_offset", v)) return 1;
	a.cb_offset = (uint16_t)v;
	if (number("cr_mult", v)) return 1;
	a.cr_mult = (uint8_t)v;
	if (number("cr_luma_mult", v)) return 1;
	a.cr_luma_mult = (uint8_t)v;
	if (number("cr_offset", v)) return 1;
	a.cr_offset = (uint16_t)v;

	struct Curve { const char* tag; const char* name; uint8_t* count; uint8_t* values; uint8_t* scaling; int max; };
	const Curve curves[3] = {
		{"sY", "luma", &a.num_y_points, a.point_y_values, a.point_y_scaling, 14},
		{"sCb", "Cb", &a.num_cb_points, a.point_cb_values, a.point_cb_scaling, 10},
		{"sCr", "Cr", &a.num_cr_points, a.point_cr_values, a.point_cr_scaling, 10},
	};
	for (const Curve& c : curves)
	{
		w.read(f);
		s = w.word();
		NEED(s && !strcmp(s, c.tag), "AFGS1 table entry: expecting %s scaling function (%s)", c.name, c.tag);
		if (number("number of scaling points", v)) return 1;
		*c.count = (uint8_t)v;
		NEED(*c.count <= c.max, "number of %s scaling points higher than %d", c.name, c.max);
		for (int k = 0; k < *c.count; k++)
		{
			if (number("scaling point (value)", v)) return 1;
			c.values[k] = (uint8_t)v;
			if (number("scaling point (scale)", v)) return 1;
			c.scaling[k] = (uint8_t)v;
		}
	}

	struct Taps { const char* tag; int16_t* coef; int extra; };
	const Taps taps[3] = {{"cY", a.ar_coeffs_y, 0}, {"cCb", a.ar_coeffs_cb, 1}, {"cCr", a.ar_coeffs_cr, 1}};
	for (const Taps& t : taps)
	{
		w.read(f);
		s = w.word();
		NEED(s && !strcmp(s, t.tag), "AFGS1 table entry: expecting %s coefficients", t.tag);
		const int n = 2 * a.ar_coeff_lag * (a.ar_coeff_lag + 1) + t.extra;   // chroma: + the luma injection tap
		for (int k = 0; k < n; k++)
		{
			if (number("AR coefficient", v)) return 1;
			t.coef[k] = (int16_t)v;
		}
	}
	return 0;
}

// ---- "name : value" syntaxes (vfgs_main.c:436-559) ----------------------------------------

struct DumpCursor { int c = 0, i = 0, j = 0; };   // position inside an SEI dump (vfgs_main.c:441)

enum Act { STOP = -1, OK = 0 };

// returns 1 if the name is known, 0 if not, -1 to stop reading, 2 on a rejected value (message set)
int assign(vfgs_hip_cfg& st, DumpCursor& d, const char* name, const char* v)
{
	fgs_sei& sei = st.sei;
	fgs_afgs1& a = st.afgs1;
	auto is = [&](const char* k) { return !strcasecmp(name, k); };
	const int n = atoi(v);

	// encoder configuration style, one key per component
	if (is("SEIFGCModelId")) { sei.model_id = (uint8_t)n; return 1; }
	if (is("SEIFGCLog2ScaleFactor")) { sei.log2_scale_factor = (uint8_t)n; return 1; }
	if (!strncasecmp(name, "SEIFGC", 6))
	{
		const size_t len = strlen(name);
		const int c = len ? name[len - 1] - '0' : -1;
		if (c >= 0 && c <= 2 && len > 5 && !strncasecmp(name + len - 5, "Comp", 4))
		{
			const std::string stem(name + 6, len - 6 - 5);
			auto stem_is = [&](const char* k) { return !strcasecmp(stem.c_str(), k); };
			if (stem_is("CompModelPresent")) { sei.comp_model_present_flag[c] = (uint8_t)n; return 1; }
			if (stem_is("NumIntensityIntervalMinus1")) { sei.num_intensity_intervals[c] = (uint16_t)(n + 1); return 1; }
			if (stem_is("NumModelValuesMinus1")) { sei.num_model_values[c] = (uint8_t)(n + 1); return 1; }
			if (stem_is("IntensityIntervalLowerBound")) { list_u8(sei.intensity_interval_lower_bound[c], 256, v); return 1; }
			if (stem_is("IntensityIntervalUpperBound")) { list_u8(sei.intensity_interval_upper_bound[c], 256, v); return 1; }
			if (stem_is("CompModelValues")) { list_models(sei.comp_model_value[c], v, sei, c); return 1; }
		}
		return 0;
	}

	// SEI dump style: the component / interval position is implied by the order of the lines
	if (!strncasecmp(name, "fg_", 3))
	{
		if (is("fg_model_id")) { sei.model_id = (uint8_t)n; return 1; }
		if (is("fg_log2_scale_factor")) { sei.log2_scale_factor = (uint8_t)n; return 1; }
		if (is("fg_characteristics_persistence_flag")) return STOP;   // end of the first message
		if (d.c > 2) return is("fg_comp_model_present_flag[c]") || is("fg_num_intensity_intervals_minus1[c]") || is("fg_num_model_values_minus1[c]") ||
		                    is("fg_intensity_interval_lower_bound[c][i]") || is("fg_intensity_interval_upper_bound[c][i]") || is("fg_comp_model_value[c][i]");
		if (is("fg_comp_model_present_flag[c]")) { sei.comp_model_present_flag[d.c] = (uint8_t)n; d.c = d.c < 2 ? d.c + 1 : 0; return 1; }
		if (is("fg_num_intensity_intervals_minus1[c]")) { sei.num_intensity_intervals[d.c] = (uint16_t)(n + 1); return 1; }
		if (is("fg_num_model_values_minus1[c]")) { sei.num_model_values[d.c] = (uint8_t)(n + 1); return 1; }
		if (is("fg_intensity_interval_lower_bound[c][i]")) { sei.intensity_interval_lower_bound[d.c][d.i & 255] = (uint8_t)n; return 1; }
		if (is("fg_intensity_interval_upper_bound[c][i]")) { sei.intensity_interval_upper_bound[d.c][d.i & 255] = (uint8_t)n; return 1; }
		if (is("fg_comp_model_value[c][i]"))
		{
			int16_t* x = sei.comp_model_value[d.c][d.i & 255];
			if (d.j < SEI_MAX_MODEL_VALUES) x[d.j] = (int16_t)n;
			if (++d.j == sei.num_model_values[d.c])
			{
				if (d.j <= SEI_MAX_MODEL_VALUES) complete_model(x, d.j, sei);
				d.j = 0;
				if (++d.i == sei.num_intensity_intervals[d.c]) { d.c++; d.i = 0; }
			}
			return 1;
		}
		return 0;
	}

	// AFGS1, one key per field
	if (strncasecmp(name, "AFGS1", 5)) return 0;
	const char* k = name + 5;
	auto k_is = [&](const char* x) { return !strcasecmp(k, x); };
#define BOUNDED(field, cond, msg) do { a.field = (uint8_t)n; if (!(cond)) { reject(msg); return 2; } return 1; } while (0)
	if (k_is("GrainSeed")) { a.grain_seed = (uint16_t)n; return 1; }
	if (k_is("NumYPoints")) BOUNDED(num_y_points, a.num_y_points <= 14, "AFGS1NumYPoints higher than 14");
	if (k_is("PointYValues")) { list_u8(a.point_y_values, 14, v); return 1; }
	if (k_is("PointYScaling")) { list_u8(a.point_y_scaling, 14, v); return 1; }
	if (k_is("ChromaScalingFromLuma")) { a.chroma_scaling_from_luma = (uint8_t)n; return 1; }
	if (k_is("NumCbPoints")) BOUNDED(num_cb_points, a.num_cb_points <= 10, "AFGS1NumCbPoints higher than 10");
	if (k_is("PointCbValues")) { list_u8(a.point_cb_values, 10, v); return 1; }
	if (k_is("PointCbScaling")) { list_u8(a.point_cb_scaling, 10, v); return 1; }
	if (k_is("NumCrPoints")) BOUNDED(num_cr_points, a.num_cr_points <= 10, "AFGS1NumCrPoints higher than 10");
	if (k_is("PointCrValues")) { list_u8(a.point_cr_values, 10, v); return 1; }
	if (k_is("PointCrScaling")) { list_u8(a.point_cr_scaling, 10, v); return 1; }
	if (k_is("GrainScaling")) BOUNDED(grain_scaling, a.grain_scaling >= 8 && a.grain_scaling <= 11, "AFGS1GrainScaling out of 8..11 range");
	if (k_is("ARCoeffLag")) BOUNDED(ar_coeff_lag, a.ar_coeff_lag <= 3, "AFGS1ARCoeffLag higher than 3");
	if (k_is("ARCoeffsY")) { list_i16(a.ar_coeffs_y, 24, v); return 1; }
	if (k_is("ARCoeffsCb")) { list_i16(a.ar_coeffs_cb, 25, v); return 1; }
	if (k_is("ARCoeffsCr")) { list_i16(a.ar_coeffs_cr, 25, v); return 1; }
	if (k_is("ARCoeffShift")) BOUNDED(ar_coeff_shift, a.ar_coeff_shift >= 6 && a.ar_coeff_shift <= 9, "AFGS1ARCoeffShift out of 6..9 range");
	if (k_is("GrainScaleShift")) BOUNDED(grain_scale_shift, a.grain_scale_shift <= 3, "AFGS1GrainScaleShift higher than 3");
	if (k_is("CbMult")) { a.cb_mult = (uint8_t)n; return 1; }
	if (k_is("CbLumaMult")) { a.cb_luma_mult = (uint8_t)n; return 1; }
	if (k_is("CbOffset")) { a.cb_offset = (uint16_t)n; return 1; }
	if (k_is("CrMult")) { a.cr_mult = (uint8_t)n; return 1; }
	if (k_is("CrLumaMult")) { a.cr_luma_mult = (uint8_t)n; return 1; }
	if (k_is("CrOffset")) { a.cr_offset = (uint16_t)n; return 1; }
	if (k_is("OverlapFlag")) { a.overlap_flag = (uint8_t)n; return 1; }
	if (k_is("ClipToRestrictedRange")) { a.clip_to_restricted_range = (uint8_t)n; return 1; }
#undef BOUNDED
	return 0;
}

int read_file(vfgs_hip_cfg& st, const char* filename)
{
	FILE* f = fopen(filename, "rt");
	if (!f) return reject("Can not open file %s", filename);
	struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};

	st.afgs1.num_y_points = st.afgs1.num_cb_points = st.afgs1.num_cr_points = 0;   // which model is active is decided anew (vfgs_main.c:452-454)
	DumpCursor d;
	int named = 0, unknown = 0;
	char line[1024];
	while (fgets(line, sizeof line, f))
	{
		if (line[0] == '#') continue;
		char* s = strtok(line, "#");                 // comment to the end of the line
		if (!s) continue;
		while (isblank((unsigned char)*s)) s++;
		s = strtok(s, ":");
		if (!s) continue;
		char* v = strtok(nullptr, ":");
		if (!v)
		{
			if (!strncasecmp(s, "filmgrn1", 8)) return read_grain_table(f, st.afgs1);   // AOM grain table: the rest of the file
			continue;
		}
		while (isblank((unsigned char)*v)) v++;
		char* e = s;
		while (*e && !isblank((unsigned char)*e)) e++;
		*e = 0;
		named++;
		const int r = assign(st, d, s, v);
		if (r == STOP) break;
		if (r == 2) return 1;
		if (r == 0) unknown++;
	}
	if (!(named > unknown)) return reject("could not ready anything from configuration file");   // sic, vfgs_main.c:556
	return 0;
}

// ---- checks (vfgs_main.c:234-303) ---------------------------------------------------------

int check_sei(const fgs_sei& sei, int format, int depth)
{
	const bool colour = sei.comp_model_present_flag[1] || sei.comp_model_present_flag[2];
	NEED(format == 420 || !colour, "color grain currently not supported on yuv422 and yuv444 formats");
	NEED(sei.model_id == 0 || !colour, "color grain currently not supported in SEI.AR mode");
	NEED(sei.model_id <= 1, "SEIFGCModelId shall be 0 or 1");
	const int rng = 1 << depth;
	for (int c = 0; c < 3; c++)
	{
		if (!sei.comp_model_present_flag[c]) continue;
		NEED(sei.num_model_values[c] >= 1 && sei.num_model_values[c] <= 6, "SEIFGCNumModelValuesMinus1Comp%d out of 0..5 range", c);
		for (int i = 0; i < sei.num_intensity_intervals[c]; i++)
		{
			const int16_t* v = sei.comp_model_value[c][i];
			NEED(sei.intensity_interval_lower_bound[c][i] <= sei.intensity_interval_upper_bound[c][i],
			     "inconsistent interval %d for component %d: upper bound should be larger or equal than lower bound", i, c);
			NEED(v[0] < rng, "scaling factor for component %d and interval %d is too large", c, i);
			if (sei.model_id == 0)
			{
				NEED(v[1] >= 2 && v[1] <= 14, "horizontal cutoff frequency for component %d and interval %d out of 2..14 range", c, i);
				// QUIRK (vfgs_main.c:253): the lower limit of the VERTICAL cut-off is tested on the horizontal one
				NEED(v[1] >= 2 && v[2] <= 14, "vertical cutoff frequency for component %d and interval %d out of 2..14 range", c, i);
			}
			else
			{
				NEED(v[1] >= -rng / 2 && v[1] < rng / 2, "first AR coefficient for component %d and interval %d is out of range", c, i);
				NEED(v[3] >= -rng / 2 && v[3] < rng / 2, "second AR coefficient for component %d and interval %d is out of range", c, i);
				NEED(v[5] >= -rng / 2 && v[5] < rng / 2, "third AR coefficient for component %d and interval %d is out of range", c, i);
			}
		}
	}
	return 0;
}

int check_afgs1(const fgs_afgs1& a, int format)
{
	NEED(format == 420 || (!a.num_cb_points && !a.num_cr_points), "color grain currently not supported on yuv422 and yuv444 formats");
	const struct { const uint8_t* v; int n; const char* name; } curves[3] = {
		{a.point_y_values, a.num_y_points, "point_y_values"}, {a.point_cb_values, a.num_cb_points, "point_cb_values"},
		{a.point_cr_values, a.num_cr_points, "point_cr_values"}};
	for (const auto& c : curves)
		for (int i = 1; i < c.n; i++)
			NEED(c.v[i] > c.v[i - 1], "afgs1.%s shall be in increasing order", c.name);
	return 0;
}

}  // namespace

extern "C" {

void vfgs_hip_cfg_defaults(vfgs_hip_cfg* cfg)
{
	// the CLI's built-in film grain characteristics (vfgs_main.c:69-120): eight intensity intervals per
	// component, {scale, horizontal cut-off, vertical cut-off}
	static const uint8_t lower[2][8] = {{0, 40, 60, 80, 100, 120, 140, 160}, {0, 64, 96, 112, 128, 144, 160, 192}};
	static const uint8_t upper[2][8] = {{39, 59, 79, 99, 119, 139, 159, 255}, {63, 95, 111, 127, 143, 159, 191, 255}};
	static const int16_t luma_scale[8] = {100, 100, 100, 110, 120, 135, 145, 180};
	static const int16_t chroma_scale[8] = {128, 96, 64, 64, 64, 64, 96, 128};
	memset(cfg, 0, sizeof *cfg);
	fgs_sei& s = cfg->sei;
	s.model_id = 0;
	s.log2_scale_factor = 5;
	for (int c = 0; c < 3; c++)
	{
		s.comp_model_present_flag[c] = 1;
		s.num_intensity_intervals[c] = 8;
		s.num_model_values[c] = 3;
		for (int k = 0; k < 8; k++)
		{
			s.intensity_interval_lower_bound[c][k] = lower[c ? 1 : 0][k];
			s.intensity_interval_upper_bound[c][k] = upper[c ? 1 : 0][k];
			s.comp_model_value[c][k][0] = c ? chroma_scale[k] : luma_scale[k];
			s.comp_model_value[c][k][1] = s.comp_model_value[c][k][2] = (int16_t)(c ? 8 : 7 + k);
		}
	}
}

int vfgs_hip_cfg_read(vfgs_hip_cfg* cfg, const char* filename) { return read_file(*cfg, filename); }

int vfgs_hip_cfg_check(const vfgs_hip_cfg* cfg, int format, int depth)
{
	if (format != 420 && format != 422 && format != 444) return reject("format %d: 420, 422 or 444", format);
	return cfg->afgs1.num_y_points ? check_afgs1(cfg->afgs1, format) : check_sei(cfg->sei, format, depth);   // vfgs_main.c:297-303
}

void vfgs_hip_cfg_adjust_chroma(vfgs_hip_cfg* cfg, int format)
{
	// The SEI describes grain at luma resolution; for subsampled chroma the cut-offs double and the
	// strength drops (vfgs_main.c:208-232).  Frequency-filtering model only.
	fgs_sei& s = cfg->sei;
	if (s.model_id != 0) return;
	auto twice = [](int16_t f) { const int t = f << 1; return (int16_t)(t > 14 ? 14 : (t < 2 ? 2 : t)); };
	for (int c = 1; c < 3; c++)
	{
		if (!s.comp_model_present_flag[c]) continue;
		for (int k = 0; k < s.num_intensity_intervals[c] && k < 256; k++)
		{
			int16_t* v = s.comp_model_value[c][k];
			if (format != 444) v[1] = twice(v[1]);
			if (format == 420) v[2] = twice(v[2]);
			if (format == 420) v[0] >>= 1;
			else if (format == 422) v[0] = (int16_t)((v[0] * 181 + 128) >> 8);   // 1/sqrt(2)
		}
	}
}

void vfgs_hip_cfg_apply_gain(vfgs_hip_cfg* cfg, unsigned gain)
{
	if (gain == 100) return;
	// powers of two go into the shift, the rest into the scale values (vfgs_main.c:561-593)
	if (cfg->afgs1.num_y_points)
	{
		fgs_afgs1& a = cfg->afgs1;
		for (; gain > 100; gain /= 2) a.grain_scaling--;
		for (; gain && gain < 50; gain *= 2) a.grain_scaling++;
		auto scale = [&](uint8_t* p, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)((int)p[i] * gain / 100); };
		scale(a.point_y_scaling, a.num_y_points);
		scale(a.point_cb_scaling, a.num_cb_points);
		scale(a.point_cr_scaling, a.num_cr_points);
	}
	else
	{
		fgs_sei& s = cfg->sei;
		for (; gain > 100; gain /= 2) s.log2_scale_factor--;
		for (; gain && gain < 50; gain *= 2) s.log2_scale_factor++;
		for (int c = 0; c < 3; c++)
			for (int i = 0; s.comp_model_present_flag[c] && i < s.num_intensity_intervals[c] && i < 256; i++)
				s.comp_model_value[c][i][0] = (int16_t)((int)s.comp_model_value[c][i][0] * gain / 100);
	}
}

void vfgs_hip_cfg_program(vfgs_hip_cfg* cfg)
{
	if (cfg->afgs1.num_y_points) vfgs_init_afgs1(&cfg->afgs1);
	else vfgs_init_sei(&cfg->sei);
}

}  // extern "C"
