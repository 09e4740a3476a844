// Shared between vfgs_host.cpp and vfgs_fw_kernel.hip: device-side constants and the launch
// record of the pattern generators (firmware layer on the GPU, include/vfgs_hip_fw.h).
#pragma once
#include <stdint.h>

#include "../../include/vfgs_hip_fw.h"

namespace vfgs {

constexpr int kFwMaxJobs = 16;        // 8 luma + 8 chroma slots
constexpr int kFwSeeds = 3;           // seed table entries the firmware starts its generator from (vfgs_fw.c:369,392,689,693,697)
constexpr int kFwStreamWords = 192;   // 82*73 generator steps + 11 bits, in 32-bit words, rounded up

// Device image of the constants: fw_tables.bin (oracle/dump_fw_tables.c documents the layout)
// followed by the generator's bit stream for each seed.  The firmware's generator is the same
// 31-bit LFSR as the hardware layer's (vfgs_fw.c:284-295), so "register after n steps" is the
// 32-bit window at bit n of one stream per seed; the host computes those streams once.
struct FwConstants {
	int8_t gauss[2048];
	uint32_t seed[256];
	int8_t dct[64][64];
	uint32_t stream[kFwSeeds][kFwStreamWords];
};
static_assert(sizeof(FwConstants) == 7168 + kFwSeeds * kFwStreamWords * 4, "blob layout");

struct FwLaunch {
	const FwConstants* k;     // device
	int8_t* bank;             // device [2][8][64][64]: the pattern banks as vfgs_hw.c:49 holds them
	int8_t* chroma_raw;       // device [8][32*32]: chroma patterns before the bank copy (non-4:2:0 layouts only)
	int njobs;
	int csubx, csuby;         // layout the chroma bank copy follows (vfgs_hw.c:320-325)
	int last_luma;            // slot of the last luma job of this call, -1 if none
	vfgs_hip_pattern_job job[kFwMaxJobs];
};

}  // namespace vfgs
