/*
 * TEST INFRASTRUCTURE -- driver around the reference CLI translation unit.
 *
 * Compiled only by oracle/Makefile, only when /root/reference is present, with
 * -I/root/reference/src.  It pulls the reference's vfgs_main.c in *by path at
 * build time* (nothing is copied into this repo) so that its static helpers
 * (read_cfg, adjust_chroma_cfg, apply_gain, vfgs_add_grain; vfgs_main.c:208,
 * :436, :561, :664) can be driven in two ways the stock main() cannot:
 *
 *   --program-only   run only the hw-programming part of main()
 *                    (vfgs_main.c:750-760 and :773-781), no frame I/O.  Linked
 *                    against oracle/trace_shim.c this is the trace recorder.
 *   --no-check       skip check_cfg(): the stock main() rejects 4:2:2 / 4:4:4
 *                    with chroma grain before any -c file is read
 *                    (vfgs_main.c:235, :739), although the hardware layer
 *                    itself handles csubx = csuby = 1.  BASELINE config #4
 *                    (2160p 8-bit 4:4:4) needs this.
 *
 *   --dump-cfg FILE  append, before every vfgs_init_sei / vfgs_init_afgs1 call, the
 *                    parameter structure the reference hands to its firmware layer
 *                    (vfgs_fw.h:51-89) as { u32 kind (0 SEI, 1 AFGS1), u32 nbytes, bytes }.
 *                    These records are the INPUT fixtures of the device firmware tests
 *                    (tests/golden/fwcfg/): data, not code.
 *
 * Without any of these flags it forwards to the reference main() untouched.
 */
#define main vfgs_reference_main
#include "vfgs_main.c"
#undef main

static const char* dump_name;

static void dump_cfg(void)
{
	FILE* f;
	unsigned kind = afgs1.num_y_points ? 1 : 0;
	unsigned n = kind ? sizeof(afgs1) : sizeof(sei);
	if (!dump_name)
		return;
	f = fopen(dump_name, "ab");
	if (!f) { perror(dump_name); exit(2); }
	fwrite(&kind, 4, 1, f);
	fwrite(&n, 4, 1, f);
	fwrite(kind ? (void*)&afgs1 : (void*)&sei, 1, n, f);
	fclose(f);
}

static int arg_is(const char* a, const char* s, const char* l)
{
	return (s && !strcasecmp(a, s)) || (l && !strcasecmp(a, l));
}

int main(int argc, const char** argv)
{
	int program_only = 0, no_check = 0;
	unsigned seed = 0, gain = 100;
	const char* in_name = NULL;
	const char* out_name = NULL;
	yuv frame;

	for (int i = 1; i < argc; i++)
	{
		if (!strcmp(argv[i], "--program-only")) program_only = 1;
		if (!strcmp(argv[i], "--no-check")) no_check = 1;
		if (!strcmp(argv[i], "--dump-cfg") && i + 1 < argc) dump_name = argv[i + 1];
	}
	if (!program_only && !no_check && !dump_name)
		return vfgs_reference_main(argc, argv);

	for (int i = 1; i < argc; i++)
	{
		const char* a = argv[i];
		if (!strcmp(a, "--program-only") || !strcmp(a, "--no-check")) continue;
		else if (!strcmp(a, "--dump-cfg")) { i++; continue; }
		else if (arg_is(a, "-w", "--width") && i + 1 < argc) width = atoi(argv[++i]);
		else if (arg_is(a, "-h", "--height") && i + 1 < argc) height = atoi(argv[++i]);
		else if (arg_is(a, "-b", "--bitdepth") && i + 1 < argc) depth = atoi(argv[++i]);
		else if (arg_is(a, "-f", "--format") && i + 1 < argc) format = read_format(argv[++i]);
		else if (arg_is(a, "-n", "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
		else if (arg_is(a, "-r", "--seed") && i + 1 < argc) seed = atoi(argv[++i]);
		else if (arg_is(a, "-g", "--gain") && i + 1 < argc) gain = atoi(argv[++i]);
		else if (arg_is(a, "-c", "--cfg") && i + 1 < argc) { if (push_cfg(argv[++i])) return 1; }
		else if (a[0] != '-' && !in_name) in_name = a;
		else if (a[0] != '-' && !out_name) out_name = a;
		else { fprintf(stderr, "ref_harness: bad argument %s\n", a); return 1; }
	}

	if (!no_check && check_cfg())
		return 1;

	/* same order as vfgs_main.c:750-760 */
	vfgs_set_depth(depth);
	vfgs_set_chroma_subsampling((format < YUV_444) ? 2 : 1, (format < YUV_422) ? 2 : 1);
	adjust_chroma_cfg();
	apply_gain(gain);
	dump_cfg();
	if (afgs1.num_y_points)
		vfgs_init_afgs1(&afgs1);
	else
		vfgs_init_sei(&sei);
	if (seed)
		vfgs_set_seed(seed);

	if (!program_only)
	{
		if (!in_name || !out_name) { fprintf(stderr, "ref_harness: need <in> <out>\n"); return 1; }
		fsrc = fopen(in_name, "rb");
		fdst = fopen(out_name, "wb");
		if (!fsrc || !fdst) { fprintf(stderr, "ref_harness: cannot open files\n"); return 1; }
		yuv_alloc(width, height, depth, format, &frame);
	}

	for (int n = 0; program_only ? (n < 1) : (((frames == 0) || (n < frames)) && !ferror(fsrc)); n++)
	{
		/* same as vfgs_main.c:773-781, with the per-cfg check optional */
		while (icfg < ncfg && (unsigned)n >= config[icfg].poc)
		{
			if (no_check)
			{
				if (read_cfg(config[icfg].filename)) return 1;
				adjust_chroma_cfg();
				apply_gain(gain);
				icfg++;
			}
			else if (pop_cfg(gain))
				break;
			dump_cfg();
			if (afgs1.num_y_points)
				vfgs_init_afgs1(&afgs1);
			else
				vfgs_init_sei(&sei);
		}
		if (program_only)
			break;
		yuv_read(&frame, fsrc);
		if (feof(fsrc))
			break;
		vfgs_add_grain(&frame);
		yuv_write(&frame, fdst);
	}
	return 0;
}
