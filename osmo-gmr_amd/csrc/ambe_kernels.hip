// ambe_kernels.hip -- GMR-1 AMBE speech decoder on gfx950: 10-byte frames -> 160 samples of 8 kHz PCM.
//
// Reference: src/codec/ambe.c (frame dispatch), frame.c (parameter decode), math.c (table cosine, DCT / DFT),
// synth.c (enhancement, unvoiced and voiced synthesis), tone.c (tone frames); API include/osmocom/gmr1/codec/codec.h.
//
// One wavefront per voice channel, frames in order (each frame needs the previous one's magnitudes, phases, noise
// state and overlap samples).  Inside a frame the lanes are, phase by phase: the harmonics (<= 56) for the parameter
// decode, enhancement and oscillator set-up; the DFT bins (64) for the noise spectrum; the output samples for the
// inverse DFT, the oscillator bank and the overlap-add.
//
// The arithmetic is the reference's, operation for operation: its cosine is a 1024-entry table indexed by a truncated
// float product, so a sum added in another order or a fused multiply-add moves table indices and flips PCM bits.
// Hence: -ffp-contract=off, sums over harmonics added lane after lane in index order (seq_sum), the DFT / inverse DFT /
// oscillator sums run serially inside the lane that owns the bin / sample.  What libm computes in the reference is
// either tabulated by the host's libm (cosine table, 2^f0log for every reachable pitch history, log2 L, tone
// amplitudes: exact by construction) or computed the way glibc computes it (ambe_libm.h: powf for 2^Mlog and x^(1/4),
// cosf for the tone frames; checked bit for bit against the host's libm by tests/test_codec_host.py).

#include <cstdlib>

#include <hip/hip_runtime.h>

#include "ambe_dev.h"
#include "profile_env.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

namespace {

constexpr float kPi = 3.141592653589793f;       // private.h:117
constexpr int kEinval = 22;

struct AmbeLds {
	float cosv[1024];
	float win[128];
	float t[128];           // windowed noise, then the new unvoiced samples
	float uw[128];          // unvoiced samples of the previous subframe
	float re[72], im[72], pw[72];
	float mlog_prev[64];    // previous frame, second subframe
	float mlog[2][64];
	float m[3][64];         // [0] previous subframe of the running pair, [1] / [2] alternate
	float scale[72];        // per bin: what its harmonic multiplies it with
	int edge[64];           // edge[l] .. edge[l + 1]: bins of harmonic l
	float phi[64];
	alignas(16) float sum_a[64];
	alignas(16) float sum_b[64];
};

__device__ __forceinline__ float tcos(const float *ct, float a)
{
	const float sc = 512.0f / kPi;
	return ct[(int)(a * sc) & 1023];
}

__device__ __forceinline__ float lane_get(float v, int i)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
}

__device__ __forceinline__ int uni(int v)
{
	return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ float unif(float v)
{
	return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// The values of lanes 0 .. 55 added in that order, starting from 0 (the reference's `sum = 0; for (...) sum += x[i];`).
// Lanes past the end of the reference's loop must hold 0: adding +0 changes nothing.  Through LDS, four per read.
__device__ __forceinline__ float seq_sum(float *buf, float v, int lane)
{
	buf[lane] = v;
	WSYNC();
	float acc = 0.0f;
#pragma unroll
	for (int i = 0; i < kAmbeMaxHarm; i += 4) {
		const float4 x = *reinterpret_cast<const float4 *>(buf + i);
		acc += x.x;
		acc += x.y;
		acc += x.z;
		acc += x.w;
	}
	WSYNC();
	return acc;
}

// `len` bits from bit `pos` of the frame, MSB first (frame.c:40-52); fr = the ten bytes as bits 79..0 of (hi:lo)
__device__ __forceinline__ unsigned field(uint64_t hi16, uint64_t lo64, int pos, int len)
{
	// bit p of the frame (0 = MSB of byte 0) is bit 79 - p of the 80-bit number hi16:lo64
	const int low = 80 - pos - len;          // position of the field's LSB
	uint64_t v;
	if (low >= 64)
		v = hi16 >> (low - 64);
	else if (low == 0)
		v = lo64;
	else
		v = (lo64 >> low) | (hi16 << (64 - low));
	return (unsigned)(v & ((1u << len) - 1u));
}

// log2 of the first subframe's fundamental (frame.c:79-118)
__device__ __forceinline__ float f0log_sf0(float before, float now, int rule)
{
	if (now != before) {
		switch (rule) {
		case 0: return now;
		case 1: return (0.65f * now) + (0.35f * before);
		case 2: return (now + before) / 2.0f;
		default: return before;
		}
	}
	const float step = 4.2672e-2f;
	switch (rule) {
	case 0:
	case 1: return now;
	case 2: return now + step;
	default: return now - step;
	}
}

__device__ __forceinline__ int harmonics(float f0)
{
	const int L = (int)floorf(0.4751f / f0);          // frame.c:125-130
	return L < 9 ? 9 : L > 56 ? 56 : L;
}

// frame.c:140-171 for destination harmonic `lane` (< Ld): src (Ls values, in LDS) seen on a grid of Ld harmonics,
// mean removed.  The grid position is a running float sum in the reference: AmbeBig::grid holds it.
__device__ __forceinline__ float regrid(const AmbeBig *big, float *buf, const float *src, int Ls, int Ld, int lane)
{
	const int a = min(max(Ls, 9), 56) - 9, b = min(max(Ld, 9), 56) - 9;
	const float mine = big->grid[a][b][lane];
	float v = 0.0f;
	if (lane < Ld) {
		const int k = (int)floorf(mine);
		if (k == 0)
			v = src[0];
		else if (k >= Ls)
			v = src[Ls - 1];
		else {
			const float frac = mine - (float)k;
			v = src[k - 1] * (1.0f - frac) + src[k] * frac;
		}
	}
	float mean = seq_sum(buf, v, lane);
	mean /= (float)Ld;
	return v - mean;
}

// float -> int16 the way the reference's x86 build does it: truncate to 32 bits (0x80000000 when out of range
// or NaN), keep the low half
__device__ __forceinline__ int pcm16(float v)
{
	const int x = (fabsf(v) < 2147483648.0f) ? (int)v : 0;
	return (int)(int16_t)x;
}

// glibc's routines work in double precision and want many registers for a short time; kept out of line, they do not
// raise the register count of the frame loop (the tone frames' cosine is on a rare path anyway)
__device__ __noinline__ float tone_cosf(float x)
{
	return ambe_libm::cosf_glibc(x);
}

__device__ __noinline__ float pow2_of_mlog(const ambe_libm::LibmTab *T, float y)
{
	bool ok;
	const float v = ambe_libm::pow2f(*T, y, &ok);              // powf(2.0, Mlog), frame.c:355
	return ok ? v : (float)exp2((double)y);
}

__device__ __noinline__ float fourth_root(const ambe_libm::LibmTab *T, float x)
{
	bool ok;
	const float v = ambe_libm::powf_pos(*T, x, 0.25f, &ok);    // powf(x, 0.25f), synth.c:352-355
	return ok ? v : (float)sqrt(sqrt((double)x));
}

}  // namespace

__global__ __launch_bounds__(64) void k_ambe_init(AmbeState *state, int n_ch, int flags)
{
	// ambe.c:36-46, synth.c:306-311
	const int ch = blockIdx.x;
	if (ch >= n_ch)
		return;
	uint32_t *w = reinterpret_cast<uint32_t *>(state + ch);
	for (unsigned i = threadIdx.x; i < sizeof(AmbeState) / 4; i += 64)
		w[i] = 0;
	WSYNC();
	if (threadIdx.x == 0) {
		AmbeState &s = state[ch];
		s.u_last = 3147;
		s.w0 = 0.09378f;
		s.L = 30;
		s.pitch_idx = 128;
		s.flags = flags;
	}
}

// AmbeBig::noise_dft, one generator state per work-group: the noise sequence (synth.c:98-110), its window
// (synth.c:130-131) and the forward DFT (math.c:118-138), one bin per lane, samples in order
__global__ __launch_bounds__(64) void k_ambe_noise_table(const AmbeTab *tab, AmbeBig *big)
{
	__shared__ float cosv[1024];
	__shared__ float t[128];
	const int lane = (int)threadIdx.x;
	const uint32_t x0 = blockIdx.x;
	const AmbeTab &T = *tab;
	for (int i = lane; i < 1024; i += 64)
		cosv[i] = T.cosv[i];
	for (int i = lane; i < 128; i += 64) {
		const uint32_t u = (T.lcg_mul[i] * x0 + T.lcg_add[i]) % 53125u;
		t[i] = i < 121 ? (float)u * T.win[i] : 0.0f;
	}
	__syncthreads();
	const float c1 = (-2.0f * kPi / 128.0f);
	const float sc = 512.0f / kPi;
	for (int r = 0; r < 2; r++) {
		const int bin = lane + 64 * r;
		if (bin > 64)
			break;
		const float cb = c1 * (float)bin;
		float ar = 0.0f, ai = 0.0f;
		for (int n = 0; n < 121; n++) {
			const float ang = cb * (float)n;
			const int idx = (int)(ang * sc);
			const float x = t[n];
			ar += x * cosv[idx & 1023];
			ai += x * cosv[(idx + 768) & 1023];
		}
		big->noise_dft[x0][bin] = make_float2(ar, ai);
	}
}

// AmbeBig::grid: the running sum of frame.c:147-166 for one (Ls, Ld) pair per work-group
__global__ __launch_bounds__(64) void k_ambe_grid_table(AmbeBig *big)
{
	const int Ls = 9 + (int)blockIdx.x / kAmbeLs, Ld = 9 + (int)blockIdx.x % kAmbeLs;
	const int lane = (int)threadIdx.x;
	const float step = (float)Ls / (float)Ld;
	float at = step, mine = step;
	for (int i = 0; i < kAmbeMaxHarm; i++) {
		if (i == lane)
			mine = at;
		at += step;
	}
	big->grid[Ls - 9][Ld - 9][lane] = mine;
}

// AmbeBig::cs: the table entries cosf_fast / sinf_fast return for the angle (-2 pi / 128) bin n (math.c:152-156)
__global__ __launch_bounds__(128) void k_ambe_cs_table(const AmbeTab *tab, AmbeBig *big)
{
	const int n = (int)threadIdx.x, bin = (int)blockIdx.x;
	const float c1 = (-2.0f * kPi / 128.0f);
	const float sc = 512.0f / kPi;
	const float ang = (c1 * (float)bin) * (float)n;
	const int idx = (int)(ang * sc);
	big->cs[bin][n] = make_float2(tab->cosv[idx & 1023], tab->cosv[(idx + 768) & 1023]);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ambe(AmbeArgs a)
{
	__shared__ AmbeLds s;
	const int lane = (int)threadIdx.x;
	const int ch = (int)blockIdx.x;
	if (ch >= a.n_ch)
		return;
	const AmbeTab &T = *a.tab;
	AmbeState &S = a.state[ch];

	for (int i = lane; i < 1024; i += 64)
		s.cosv[i] = T.cosv[i];
	for (int i = lane; i < 128; i += 64) {
		s.win[i] = T.win[i];
		s.uw[i] = i < 121 ? S.uw[i] : 0.0f;
	}
	s.mlog_prev[lane] = lane < 56 ? S.Mlog[lane] : 0.0f;
	s.m[0][lane] = lane < 56 ? S.M[lane] : 0.0f;
	s.phi[lane] = lane < 56 ? S.phi[lane] : 0.0f;

	float tone_ph1 = S.tone_ph1, tone_ph2 = S.tone_ph2;
	float prev_f0log = S.f0log, prev_w0 = S.w0, prev_gain = S.gain, psi1 = S.psi1, SE = S.SE;
	// (a state the caller damaged must not turn into an out-of-range table index: clamp what is used as one)
	int prev_idx = min(max(S.pitch_idx, 0), 128), prev_L = min(max(S.L, 9), kAmbeMaxHarm);
	uint64_t prev_V = ((uint64_t)S.V[1] << 32) | S.V[0];
	uint32_t u_last = S.u_last;
	uint64_t slot[2] = {((uint64_t)S.slot[0][1] << 32) | S.slot[0][0], ((uint64_t)S.slot[1][1] << 32) | S.slot[1][0]};
	const bool cleared = (S.flags & 1) != 0;
	int m_prev = 0;          // which of s.m[] holds the previous subframe's magnitudes
	WSYNC();

	const float *ct = s.cosv;
	// window weights of this lane's two output samples (i = lane and lane + 64 < 80)
	const int i0 = lane, i1 = lane + 64;

	for (int f = 0; f < a.n_frames; f++) {
		const uint8_t *fr = a.frames + ((size_t)ch * a.n_frames + f) * kAmbeFrameBytes;
		int16_t *out = a.pcm + ((size_t)ch * a.n_frames + f) * a.pcm_stride;
		uint64_t hi16 = 0, lo64 = 0;
		for (int k = 0; k < 2; k++)
			hi16 = (hi16 << 8) | fr[k];
		for (int k = 2; k < 10; k++)
			lo64 = (lo64 << 8) | fr[k];
		hi16 = ((uint64_t)(unsigned)uni((int)(hi16 >> 32)) << 32) | (unsigned)uni((int)hi16);
		lo64 = ((uint64_t)(unsigned)uni((int)(lo64 >> 32)) << 32) | (unsigned)uni((int)lo64);
		const unsigned byte0 = (unsigned)(hi16 >> 8) & 0xff, byte1 = (unsigned)hi16 & 0xff;
		int rv = 0;

		if ((byte0 & 0xfc) == 0xf8) {
			// silence indication: 160 zeros, nothing else changes (ambe.c:115-118)
			for (int i = lane; i < kAmbeFrameSamples; i += 64)
				out[i] = 0;
		} else if ((byte0 & 0xfc) == 0xfc) {
			// ---- tone frame (tone.c:115-204) ----
			const int N = a.tone_n;
			const int sel = byte0 & 3;
			int code = 0;
			for (int bit = 0; bit < 8; bit++) {
				int ones = ((byte0 >> (7 - bit)) & 1) + ((byte1 >> (7 - bit)) & 1);
				for (int j = 0; j < 6; j++)
					ones += (int)(lo64 >> (56 - 8 * j + 7 - bit)) & 1;
				code = (code << 1) | (ones >= 4 ? 1 : 0);
			}
			const int start = (sel & 2) ? 0 : N >> 1;
			const int stop = (sel & 1) ? (N - 1) : ((N >> 1) - 1);
			int f1 = 0, f2 = 0, ampl = T.tone_ampl[byte1];
			bool two = false, any = false;
			if (start >= stop || code == 0xff) {
				// nothing to add
			} else if (code >= 0x80 && code <= 0xa3) {
				const int k = code & 0xf;
				if (code >= 0xa0) {
					f1 = k == 0 ? 440 : k == 1 ? 480 : k == 2 ? 630 : 490;
					f2 = k == 0 ? 350 : k == 1 ? 440 : k == 2 ? 480 : 350;
				} else if (code >= 0x90) {
					const int c = k >> 2, r = k & 3;
					f1 = c == 0 ? 1052 : c == 1 ? 1162 : c == 2 ? 1297 : 1430;
					f2 = r == 0 ? 606 : r == 1 ? 672 : r == 2 ? 743 : 820;
				} else {
					const int c = k >> 2, r = k & 3;
					f1 = c == 0 ? 1209 : c == 1 ? 1336 : c == 2 ? 1477 : 1633;
					f2 = r == 0 ? 697 : r == 1 ? 770 : r == 2 ? 852 : 941;
				}
				ampl >>= 1;
				two = any = true;
			} else if (code < 0x7f) {
				f1 = (code * 125) >> 2;
				any = true;
			} else {
				rv = -kEinval;
			}
			// the phase is a running float sum over the samples (tone.c:101-107): run it, each lane keeps its own
			const int n = any ? stop - start + 1 : 0;
			const float st1 = (2.0f * kPi * (float)f1) / 8000.0f, st2 = (2.0f * kPi * (float)f2) / 8000.0f;
			for (int base = 0; base < N; base += 64) {
				const int i = base + lane;          // sample index in the frame
				float p1 = tone_ph1, p2 = tone_ph2;
				float my1 = 0.0f, my2 = 0.0f;
				const int upto = min(base + 64, start + n) - start;     // samples of the tone generated so far + this block
				const int from = max(base, start) - start;
				// advance from `from` to `upto`, catching this lane's phase
				for (int k = from; k < upto; k++) {
					if (start + k == i) {
						my1 = p1;
						my2 = p2;
					}
					p1 += st1;
					p2 += st2;
				}
				if (upto > from) {
					tone_ph1 = p1;
					if (two)
						tone_ph2 = p2;
				}
				int v = 0;
				if (any && i >= start && i < start + n) {
					v = (int)(int16_t)(int)((float)ampl * tone_cosf(my1));
					if (two)
						v = (int)(int16_t)(v + (int)(int16_t)(int)((float)ampl * tone_cosf(my2)));
				}
				if (i < N)
					out[i] = (int16_t)v;
			}
		} else {
			// ---- speech frame (ambe.c:77-108) ----
			const int pitch = (int)field(hi16, lo64, 0, 7);
			const int pitch_rule = (int)field(hi16, lo64, 48, 2);
			const int gain_i = (int)((field(hi16, lo64, 7, 6) << 2) | field(hi16, lo64, 50, 2));
			const int vuv_i = (int)field(hi16, lo64, 13, 6);
			const int prba12_i = (int)((field(hi16, lo64, 19, 6) << 1) | field(hi16, lo64, 52, 1));
			const int prba34_i = (int)((field(hi16, lo64, 25, 3) << 3) | field(hi16, lo64, 53, 3));
			const int prba57_i = (int)((field(hi16, lo64, 28, 3) << 4) | field(hi16, lo64, 56, 4));
			const int hoc_i[4] = {(int)((field(hi16, lo64, 31, 3) << 4) | field(hi16, lo64, 60, 4)),
			                      (int)((field(hi16, lo64, 34, 3) << 3) | field(hi16, lo64, 64, 3)),
			                      (int)((field(hi16, lo64, 37, 2) << 4) | field(hi16, lo64, 67, 4)),
			                      (int)((field(hi16, lo64, 39, 2) << 3) | field(hi16, lo64, 71, 3))};
			const int mag_rule = (int)field(hi16, lo64, 46, 2);
			const int perr14_i = (int)((field(hi16, lo64, 41, 3) << 3) | field(hi16, lo64, 74, 3));
			const int perr58_i = (int)((field(hi16, lo64, 44, 2) << 3) | field(hi16, lo64, 77, 3));

			// fundamentals (frame.c:299-305): 2^x looked up in the host-built tables
			const float f0log1 = -4.312f - 2.1336e-2f * (float)pitch;
			const float f0_1 = T.f0_sf1[pitch];
			const float f0_0 = T.f0_sf0[(prev_idx * 128 + pitch) * 4 + pitch_rule];
			const float f0s[2] = {unif(f0_0), unif(f0_1)};
			const int Ls[2] = {uni(harmonics(f0s[0])), uni(harmonics(f0s[1]))};
			const unsigned pat = T.vuv[vuv_i];
			float gains[2];
			for (int k = 0; k < 2; k++) {
				gains[k] = (0.5f * prev_gain) + T.gain[gain_i * 2 + k];
				if (gains[k] > 13.0f)
					gains[k] = 13.0f;
			}
			const int Lp = uni(prev_L);

			// -- second subframe's log magnitudes (frame.c:175-242) --
			if (!(a.dbg & 4)) {
				const int L = Ls[1];
				float v = regrid(a.big, s.sum_a, s.mlog_prev, Lp, L, lane) * 0.65f;
				// PRBA vector -> 8 points (lanes 0..7)
				float g[8];
				g[0] = 0.0f;
				g[1] = T.prba12[prba12_i * 2 + 0];
				g[2] = T.prba12[prba12_i * 2 + 1];
				g[3] = T.prba34[prba34_i * 2 + 0];
				g[4] = T.prba34[prba34_i * 2 + 1];
				g[5] = T.prba57[prba57_i * 3 + 0];
				g[6] = T.prba57[prba57_i * 3 + 1];
				g[7] = T.prba57[prba57_i * 3 + 2];
				float R = g[0];
				{
					const float base = kPi / 8.0f;
					const float nh = (float)(lane & 7) + .5f;
					for (int k = 1; k < 8; k++)
						R += 2.0f * g[k] * tcos(ct, base * (float)k * nh);
				}
				// blocks: which one this harmonic is in, and where
				int Lb[4], first[4];
				for (int b = 0, at = 0; b < 4; b++) {
					Lb[b] = T.hpg[(L - 9) * 4 + b];
					first[b] = at;
					at += Lb[b];
				}
				const int b = lane >= first[3] ? 3 : lane >= first[2] ? 2 : lane >= first[1] ? 1 : 0;
				const float half_rsqrt2 = (1.0f / (2.0f * 1.41421356237309504880f));
				float C0[4], C1[4], weighted = 0.0f;
				for (int q = 0; q < 4; q++) {
					const float ra = lane_get(R, 2 * q), rb = lane_get(R, 2 * q + 1);
					C0[q] = (ra + rb) * 0.5f;
					C1[q] = (ra - rb) * half_rsqrt2;
					weighted += C0[q] * (float)Lb[q];
				}
				if (lane < L) {
					const float *hoc = &T.hoc[b][hoc_i[b] * 4];
					const float C[6] = {C0[b], C1[b], hoc[0], hoc[1], hoc[2], hoc[3]};
					const float base = kPi / (float)Lb[b];
					const float nh = (float)(lane - first[b]) + .5f;
					float c = C[0];
					for (int k = 1; k < 6; k++)
						c += 2.0f * C[k] * tcos(ct, base * (float)k * nh);
					v += c;
				}
				const float shift = gains[1] - (0.5f * T.log2_L[L]) - (weighted / (float)L);
				v += shift;
				s.mlog[1][lane] = lane < L ? v : 0.0f;
			}
			WSYNC();
			// -- first subframe's (frame.c:246-286) --
			if (!(a.dbg & 4)) {
				const int L = Ls[0];
				const float from_before = regrid(a.big, s.sum_a, s.mlog_prev, Lp, L, lane);
				const float from_after = regrid(a.big, s.sum_a, s.mlog[1], Ls[1], L, lane);
				const float al = T.interp[mag_rule];
				float e[9];
				e[0] = 0.0f;
				for (int k = 0; k < 4; k++) {
					e[1 + k] = T.perr14[perr14_i * 4 + k];
					e[5 + k] = T.perr58[perr58_i * 4 + k];
				}
				const float base = kPi / (float)L;
				const float nh = (float)lane + .5f;
				float fix = e[0];
				for (int k = 1; k < 9; k++)
					fix += 2.0f * e[k] * tcos(ct, base * (float)k * nh);
				const float level = gains[0] - (0.5f * T.log2_L[L]);
				const float v = level + fix + (al * from_before) + ((1.0f - al) * from_after);
				s.mlog[0][lane] = lane < L ? v : 0.0f;
			}
			WSYNC();

			// -- the two subframes, one after the other --
			for (int sub = 0; sub < 2; sub++) {
				const int L = Ls[sub];
				const float f0 = f0s[sub];
				const int Lq = sub == 0 ? Lp : Ls[0];                 // harmonics of the subframe before
				const float w0 = f0 * (2.0f * kPi);                     // frame.c:342-359
				const float w0q = sub == 0 ? prev_w0 : f0s[0] * (2.0f * kPi);
				const int m_now = m_prev == 0 ? 1 : (m_prev == 1 ? 2 : 1);
				// noise generator: this subframe's 121 values start from u_last, the next one's 80 further on
				// (synth.c:98-110, 127-128); their windowed spectrum is a function of u_last alone: AmbeBig::noise_dft.
				// Asked for here, used after the enhancement.
				const uint32_t x0 = u_last & 0xffffu;
				u_last = (T.lcg_mul[79] * x0 + T.lcg_add[79]) % 53125u;
				const float2 *row = a.big->noise_dft[x0 < (uint32_t)kAmbeNoiseStates ? x0 : 0];
				const float2 spec = row[lane];

				// per-harmonic voicing and linear magnitude
				const unsigned bands = sub == 0 ? (pat & 0xff) : (pat >> 8);       // MSB = lowest band (frame.c:313-318)
				int band = (int)((float)lane * 16.0f * f0);
				if (band > 7)
					band = 7;                                          // decision D10 (oracle/orc_ambe.c)
				const bool voiced_l = lane < L && ((bands >> (7 - band)) & 1);
				const uint64_t below_L = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
				const uint64_t Vmask = (__ballot(voiced_l) & below_L) | (cleared ? 0ull : (slot[sub] & ~below_L));
				slot[sub] = Vmask;
				float M = 0.0f;
				if (lane < L) {
					const float unv = 0.2046f / sqrtf(w0);
					const float two_to = pow2_of_mlog(a.libm, s.mlog[sub][lane]);
					M = two_to / 6.0f;
					if (!voiced_l)
						M *= unv;
				}

				// spectral enhancement (synth.c:314-379)
				const float cw = tcos(ct, w0 * (float)(lane + 1));
				{
					const float p = M * M;
					const float q = p * cw;
					s.sum_a[lane] = p;                     // lanes >= L hold M = 0
					s.sum_b[lane] = q;
					WSYNC();
					float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
					for (int i = 0; i < kAmbeMaxHarm; i += 4) {
						const float4 x = *reinterpret_cast<const float4 *>(s.sum_a + i);
						const float4 y = *reinterpret_cast<const float4 *>(s.sum_b + i);
						r0 += x.x; r0 += x.y; r0 += x.z; r0 += x.w;
						r1 += y.x; r1 += y.y; r1 += y.z; r1 += y.w;
					}
					WSYNC();
					const float k1 = 0.96f * kPi / (w0 * r0 * (r0 * r0 - r1 * r1));
					const float k2 = r0 * r0 + r1 * r1;
					const float k3 = 2.0f * r0 * r1;
					float w = 1.0f;
					if ((lane + 1) * 8 > L) {
						const float x = k1 * (k2 - k3 * cw);
						const float root4 = fourth_root(a.libm, x);
						w = sqrtf(M) * root4;
						if (w > 1.2f)
							w = 1.2f;
						else if (w < 0.5f)
							w = 0.5f;
					}
					M *= w;
					if (lane >= L)
						M = 0.0f;
					const float after = seq_sum(s.sum_a, M * M, lane);
					const float norm = sqrtf(r0 / after);
					M *= norm;
					if (lane >= L)
						M = 0.0f;
					SE = 0.95f * SE + 0.05f * r0;
					if (SE < 1e4f)
						SE = 1e4f;
				}
				s.m[m_now][lane] = M;

				// ---- unvoiced part (synth.c:114-214) ----
				// band edges of this lane's harmonic
				int hi = (int)ceilf(128.0f / (2 * kPi) * ((float)lane + 1.5f) * w0);
				if (hi > 65)
					hi = 65;
				int lo0 = (int)ceilf(128.0f / (2 * kPi) * (.5f) * w0);
				if (lo0 > 65)
					lo0 = 65;
				if (lane == 0)
					s.edge[0] = lo0;
				if (lane < L)
					s.edge[lane + 1] = hi;
				WSYNC();
				const int lo = lane < L ? s.edge[lane] : 0;
				const bool noisy_l = lane < L && !voiced_l && hi > lo;
				const bool any_noise = __ballot(noisy_l) != 0ull;
				const int last_edge = uni(s.edge[L]);

				float nu0 = 0.0f, nu1 = 0.0f;      // new unvoiced samples n = lane, lane + 64
				if (any_noise && !(a.dbg & 1)) {
					const bool need64 = last_edge > 64;        // a band reaches the last bin: decision D10 inputs only
					{
						const float2 v = spec;
						s.re[lane] = v.x;
						s.im[lane] = v.y;
						s.pw[lane] = v.x * v.x + v.y * v.y;
						if (lane == 0) {
							const float2 w = need64 ? row[64] : make_float2(0.0f, 0.0f);
							s.re[64] = w.x;
							s.im[64] = w.y;
							s.pw[64] = w.x * w.x + w.y * w.y;
						}
					}
					WSYNC();
					// per harmonic: energy of its bins -> the factor its bins are multiplied with (zero: voiced band)
					s.scale[lane] = 0.0f;                      // bins no harmonic covers are cleared (synth.c:141-144, 184-187)
					if (lane == 0)
						s.scale[64] = 0.0f;
					WSYNC();
					if (lane < L) {
						float e = 0.0f;
						for (int k = lo; k < hi; k++)
							e += s.pw[k];
						const float f = voiced_l ? 0.0f : 76.89f * M / sqrtf(e / (float)(hi - lo));
						for (int k = lo; k < hi; k++)
							s.scale[k] = f;
					}
					WSYNC();
					uint64_t live = 0;
					{
						const float f = s.scale[lane];
						const float vr = f == 0.0f ? 0.0f : s.re[lane] * f, vi = f == 0.0f ? 0.0f : s.im[lane] * f;
						s.re[lane] = vr;
						s.im[lane] = vi;
						live = __ballot(vr != 0.0f || vi != 0.0f);
						if (lane == 0) {
							const float g = s.scale[64];
							s.re[64] = (need64 && g != 0.0f) ? s.re[64] * g : 0.0f;
							s.im[64] = (need64 && g != 0.0f) ? s.im[64] * g : 0.0f;
						}
					}
					WSYNC();
					const bool live64 = need64 && (s.re[64] != 0.0f || s.im[64] != 0.0f);
					// inverse DFT, samples n = lane and lane + 64 (math.c:142-163); zero bins add nothing
					float a0 = 0.0f, a1 = 0.0f;
					uint64_t todo = live;
					// the table values of the next live bin are requested while the current one is added
					int k = todo ? __builtin_ctzll(todo) : 0;
					float2 ca = a.big->cs[k][lane], cb = a.big->cs[k][lane + 64];
					while (todo) {
						todo &= todo - 1;
						const int kn = todo ? __builtin_ctzll(todo) : k;
						const float2 na = a.big->cs[kn][lane], nb = a.big->cs[kn][lane + 64];
						const float twice = k == 0 ? 1.0f : 2.0f;
						const float br = s.re[k], bi = s.im[k];
						a0 += twice * (br * ca.x + bi * ca.y);
						a1 += twice * (br * cb.x + bi * cb.y);
						k = kn;
						ca = na;
						cb = nb;
					}
					if (live64) {
						const float br = s.re[64], bi = s.im[64];
						const float2 ca = a.big->cs[64][lane], cb = a.big->cs[64][lane + 64];
						a0 += 1.0f * (br * ca.x + bi * ca.y);
						a1 += 1.0f * (br * cb.x + bi * cb.y);
					}
					nu0 = a0 / 128.0f;
					nu1 = a1 / 128.0f;
				}
				WSYNC();
				// weighted overlap-add with the previous subframe's samples (synth.c:196-213)
				float suv0, suv1 = 0.0f;
				{
					s.t[lane] = nu0;
					s.t[lane + 64] = nu1;
					WSYNC();
					auto ola = [&](int i) -> float {
						if (i < 21)
							return s.uw[i + 60];
						if (i < 60) {
							const float wa = s.win[i + 60], wb = s.win[i - 20];
							return (wa * s.uw[i + 60] + wb * s.t[i - 20]) / (wa * wa + wb * wb);
						}
						return s.t[i - 20];
					};
					suv0 = ola(i0);
					if (i1 < 80)
						suv1 = ola(i1);
					WSYNC();
					s.uw[lane] = nu0;
					s.uw[lane + 64] = nu1;
				}

				// ---- voiced part (synth.c:218-302) ----
				const int Lmax = Lq > L ? Lq : L;
				const uint64_t below_max = Lmax >= 64 ? ~0ull : ((1ull << Lmax) - 1ull);
				const int n_unv = __builtin_popcountll(~Vmask & below_max);
				psi1 = remainderf(psi1 + (w0 + w0q) * 40.0f, 2 * kPi);
				const float spread = (float)n_unv / (float)L;
				// oscillator of harmonic `lane`
				const bool v_now = lane < L && ((Vmask >> lane) & 1);
				const bool v_was = lane < Lq && ((prev_V >> lane) & 1);
				const float mg_now = lane < L ? M : 0.0f;
				const float mg_was = lane < Lq ? s.m[m_prev][lane] : 0.0f;
				const float w_now = (float)(lane + 1) * w0;
				const float w_was = (float)(lane + 1) * w0q;
				const float ph_was = s.phi[lane];
				float ph_now = psi1 * (float)(lane + 1);
				if (lane >= (L / 4) || lane >= Lmax)
					ph_now += spread * T.rho[lane < 56 ? lane : 55];
				WSYNC();
				if (lane < 56)
					s.phi[lane] = ph_now;
				const bool smooth = v_now && v_was && (lane < 7) && (fabsf(w_now - w_was) < (.1f * w_now));
				const float dm = (mg_now - mg_was) / 80.0f;
				const float dp = ph_now - ph_was - (w_now + w_was) * 40.0f;
				const float dw = (dp - 2 * kPi * floorf((dp + kPi) / (2 * kPi))) / 80.0f;
				const float ta = w_was + dw;
				const float tb = (w_now - w_was) / 160.0f;
				const uint64_t m_smooth = __ballot(smooth && lane < Lmax);
				const uint64_t m_cur = __ballot(!smooth && v_now && lane < Lmax);
				const uint64_t m_old = __ballot(!smooth && v_was && lane < Lmax);

				float sv0 = 0.0f, sv1 = 0.0f;
				const float fi0 = (float)i0, fi1 = (float)i1;
				const float wc0 = i0 >= 21 ? s.win[i0 - 20] : 0.0f;         // current subframe fades in from sample 21
				const float wc1 = i1 < 80 ? s.win[i1 - 20] : 0.0f;
				const float wo0 = i0 < 60 ? s.win[i0 + 60] : 0.0f;          // previous one fades out until sample 59
				for (int l = 0; l < ((a.dbg & 2) ? 0 : Lmax); l++) {
					const uint64_t bit = 1ull << l;
					if (m_smooth & bit) {
						const float a_m = lane_get(mg_was, l), a_dm = lane_get(dm, l), a_ph = lane_get(ph_was, l);
						const float a_ta = lane_get(ta, l), a_tb = lane_get(tb, l);
						sv0 += (a_m + fi0 * a_dm) * tcos(ct, a_ph + (a_ta + a_tb * fi0) * fi0);
						sv1 += (a_m + fi1 * a_dm) * tcos(ct, a_ph + (a_ta + a_tb * fi1) * fi1);
					}
					if (m_cur & bit) {
						const float a_m = lane_get(mg_now, l), a_ph = lane_get(ph_now, l), a_w = lane_get(w_now, l);
						if (i0 >= 21)
							sv0 += wc0 * a_m * tcos(ct, a_ph + a_w * (float)(i0 - 80));
						sv1 += wc1 * a_m * tcos(ct, a_ph + a_w * (float)(i1 - 80));
					}
					if (m_old & bit) {
						const float a_m = lane_get(mg_was, l), a_ph = lane_get(ph_was, l), a_w = lane_get(w_was, l);
						if (i0 < 60)
							sv0 += wo0 * a_m * tcos(ct, a_ph + a_w * fi0);
					}
				}

				// samples (synth.c:381-395)
				out[sub * 80 + i0] = (int16_t)pcm16((suv0 + 2.0f * sv0) * 4.0f);
				if (i1 < 80)
					out[sub * 80 + i1] = (int16_t)pcm16((suv1 + 2.0f * sv1) * 4.0f);

				// this subframe becomes "the one before"
				prev_V = Vmask;
				m_prev = m_now;
				WSYNC();
			}
			// carry to the next frame (ambe.c:103-104)
			prev_f0log = f0log1;
			prev_idx = pitch;
			prev_w0 = f0s[1] * (2.0f * kPi);
			prev_L = Ls[1];
			prev_gain = gains[1];
			s.mlog_prev[lane] = s.mlog[1][lane];
			WSYNC();
		}
		if (a.rv && lane == 0)
			a.rv[(size_t)ch * a.n_frames + f] = rv;
	}

	// state back
	WSYNC();
	if (lane == 0) {
		S.tone_ph1 = tone_ph1;
		S.tone_ph2 = tone_ph2;
		S.f0log = prev_f0log;
		S.pitch_idx = prev_idx;
		S.w0 = prev_w0;
		S.L = prev_L;
		S.gain = prev_gain;
		S.V[0] = (uint32_t)prev_V;
		S.V[1] = (uint32_t)(prev_V >> 32);
		S.u_last = u_last;
		S.psi1 = psi1;
		S.SE = SE;
		S.slot[0][0] = (uint32_t)slot[0];
		S.slot[0][1] = (uint32_t)(slot[0] >> 32);
		S.slot[1][0] = (uint32_t)slot[1];
		S.slot[1][1] = (uint32_t)(slot[1] >> 32);
	}
	if (lane < 56) {
		S.Mlog[lane] = s.mlog_prev[lane];
		S.M[lane] = s.m[m_prev][lane];
		S.phi[lane] = s.phi[lane];
	}
	for (int i = lane; i < 121; i += 64)
		S.uw[i] = s.uw[i];
}

hipError_t launch_ambe(const AmbeArgs &a, hipStream_t stream)
{
	if (a.n_ch <= 0 || a.n_frames <= 0)
		return hipSuccess;
	static int dbg = -1;
	if (dbg < 0) {
		const char *e = profile_env("GMR1_HIP_AMBE_DBG");
		dbg = e ? atoi(e) : 0;
	}
	AmbeArgs b = a;
	b.dbg = dbg;
	hipLaunchKernelGGL(k_ambe, dim3(a.n_ch), dim3(64), 0, stream, b);
	return hipGetLastError();
}

hipError_t launch_ambe_big(const AmbeTab *tab, AmbeBig *big, hipStream_t stream)
{
	hipLaunchKernelGGL(k_ambe_noise_table, dim3(kAmbeNoiseStates), dim3(64), 0, stream, tab, big);
	hipLaunchKernelGGL(k_ambe_cs_table, dim3(kAmbeBins), dim3(128), 0, stream, tab, big);
	hipLaunchKernelGGL(k_ambe_grid_table, dim3(kAmbeLs * kAmbeLs), dim3(64), 0, stream, big);
	return hipGetLastError();
}

hipError_t launch_ambe_init(AmbeState *state, int n_ch, int flags, hipStream_t stream)
{
	if (n_ch <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_ambe_init, dim3(n_ch), dim3(64), 0, stream, state, n_ch, flags);
	return hipGetLastError();
}

}  // namespace gmr1
