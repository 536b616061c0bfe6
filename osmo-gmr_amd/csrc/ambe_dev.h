// ambe_dev.h -- device-side types of the AMBE speech decoder (ambe_kernels.hip, capi_ambe.cpp).
//
// Reference: src/codec/{ambe,frame,math,synth,tone}.c behind include/osmocom/gmr1/codec/codec.h:37-45.
// One decoder = one voice channel; its frames depend on each other (magnitude prediction, oscillator phases, noise
// generator, overlap-add), channels do not: the kernel runs one wavefront per channel and walks its frames in order.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "ambe_libm.h"

namespace gmr1 {

constexpr int kAmbeFrameBytes = 10;
constexpr int kAmbeFrameSamples = 160;
constexpr int kAmbeMaxHarm = 56;

// What a decoder carries from frame to frame (reference struct ambe_decoder, src/codec/private.h:84-111), in the
// form the kernel wants it: voicing as bit masks, the previous pitch also as its 7-bit index (128 = no speech
// frame yet) because the fundamental is looked up, not computed (AmbeTab::f0_sf0).
struct AmbeState {
	float tone_ph1, tone_ph2;          // tone.c phase accumulators
	float f0log;                       // previous frame, second subframe
	int32_t pitch_idx;
	float w0;
	int32_t L;
	float gain;
	uint32_t V[2];                     // its per-harmonic voicing, bit l
	float Mlog[kAmbeMaxHarm];
	float M[kAmbeMaxHarm];             // after enhancement (what the voiced synthesiser cross-fades from)
	uint32_t u_last;                   // noise generator (16 bits)
	float uw[121];                     // unvoiced samples of the previous subframe, before overlap-add
	float psi1;
	float phi[kAmbeMaxHarm];
	float SE;
	// decision D9 (oracle/orc_ambe.c): the voicing arrays of the two subframes persist from frame to frame, the
	// way the reference's uncleared locals do in its own program; flags bit 0 = start each frame from zeros instead
	uint32_t slot[2][2];
	int32_t flags;
	int32_t reserved[2];
};
static_assert(sizeof(AmbeState) % 16 == 0, "states are laid out back to back");

// Constant tables in device memory.  Everything the reference gets from libm on values that can be enumerated is
// computed by libm on the host when the library loads (capi_ambe.cpp), so those values are the reference's own.
struct AmbeTab {
	float cosv[1024];                  // cosf(pi i / 512), math.c:40-52
	float win[128];                    // synthesis window ws[121], synth.c:36-54
	float f0_sf1[128];                 // powf(2, -4.312 - 2.1336e-2 pitch), frame.c:300-301
	float log2_L[64];                  // log2f(L)
	int32_t tone_ampl[256];            // (int)(32767 exp2f((a - 255) / 17)), tone.c:146
	uint32_t lcg_mul[128], lcg_add[128];   // noise generator stepped i+1 times at once: x_i = (mul x_0 + add) mod 53125
	float gain[512], prba12[256], prba34[128], prba57[384];
	float hoc[4][512];                 // hoc0 [128][4]; hoc1..3 [64][4]
	float interp[4], perr14[256], perr58[128], rho[56];
	uint16_t vuv[64];
	uint8_t hpg[192];
	float f0_sf0[129 * 128 * 4];       // powf(2, interpolated f0log) by (previous pitch index, pitch, rule), frame.c:303-305
};

// Three more tables, too large for AmbeTab and built on the device itself (k_ambe_noise_table, k_ambe_cs_table, k_ambe_grid_table) with
// the very operations the synthesiser would otherwise repeat for every subframe:
//  * noise_dft[x][bin] = (re, im) of the 128-point DFT of the windowed noise sequence the generator produces from
//    state x (synth.c:127-134, math.c:118-138).  The generator has 53125 states (x -> 171 x + 11213 mod 53125, full
//    period) and nothing else enters that spectrum, so it is a table: 53125 x 65 x 8 bytes = 27.6 MB of HBM.
//  * cs[bin][n] = (cos, sin) table values the inverse DFT multiplies with (math.c:142-163): the table index is a
//    truncated float product of constants, the same for every frame.
constexpr int kAmbeNoiseStates = 53125;
constexpr int kAmbeBins = 65;
//  * grid[Ls - 9][Ld - 9][i] = the position ambe_resample_mag reaches at destination harmonic i when it maps Ls
//    harmonics onto Ld (frame.c:147-166): a running float sum of Ls / Ld, the same for every frame with that pair.
constexpr int kAmbeLs = 48;            // harmonic counts 9 .. 56
struct AmbeBig {
	float2 noise_dft[kAmbeNoiseStates][kAmbeBins];
	float2 cs[kAmbeBins][128];
	float grid[kAmbeLs][kAmbeLs][64];
};

struct AmbeArgs {
	int n_ch, n_frames;
	const uint8_t *frames;             // [n_ch][n_frames][10]
	int16_t *pcm;                      // [n_ch][n_frames][pcm_stride]
	int pcm_stride;                    // >= max(160, tone_n)
	int32_t *rv;                       // optional [n_ch][n_frames]: 0 or -EINVAL (tone.c:197-201)
	AmbeState *state;                  // [n_ch], read and written
	const AmbeTab *tab;
	const AmbeBig *big;
	const ambe_libm::LibmTab *libm;    // glibc's powf tables (ambe_libm.h)
	int tone_n;                        // the N of gmr1_codec_decode_frame: samples a tone frame covers (160 in batches)
	int dbg;                           // timing experiments only (GMR1_HIP_AMBE_DBG): 1 no noise path, 2 no oscillators, 4 no parameter decode
};

hipError_t launch_ambe(const AmbeArgs &a, hipStream_t stream);
hipError_t launch_ambe_init(AmbeState *state, int n_ch, int flags, hipStream_t stream);
hipError_t launch_ambe_big(const AmbeTab *tab, AmbeBig *big, hipStream_t stream);

}  // namespace gmr1
