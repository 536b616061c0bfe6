// chan_kernels.hip -- wideband capture -> per-ARFCN streams at sym_rate x sps (gfx950).
//
// What the reference's recorder does with GNU Radio blocks (utils/gmr1_rx_sdr.py:391-602):
// pfb.channelizer_ccf(n_chans, low_pass taps, 2x oversampled) followed, per ARFCN, by
// pfb.arb_resampler_ccf(rate, 32-phase root-raised-cosine bank).  Here as two streaming kernels:
//
//   k_pfb<64>  : Y_k[t] = sum_s x[s] h[t D - s] e^{-j 2 pi k s / 64},  D = 32.
//                One wavefront walks a range of output instants; lane r owns the polyphase branch of
//                the samples s = r (mod 64).  Each new block of 64 samples is ONE coalesced 512-byte
//                load and serves two instants (t even: taps h[64 q - r], t odd: h[64 q + 32 - r]), so
//                the FIR is register-blocked: an 11-deep window of the lane's own samples and its
//                22 taps live in VGPRs, nothing is staged.  The 64-point DFT across the branches is
//                six radix-2 stages across the LANES of the wave (DPP for spans 1..8, bpermute for
//                16 and 32); results leave bit-reversed, are transposed through a wave-private LDS tile
//                and written as 128-byte runs per channel.  Only the channels asked for are stored.
//   k_pfb_any  : the same filterbank for any even channel count (direct DFT of the selected channels).
//   k_resamp   : out[n] = sum_k (b_j[k] + frac d_j[k]) y[i - k],  phase = j0 + n num/den in 1/32 input
//                samples kept in integers (j = phase mod 32, i = phase / 32, frac = remainder / den).
//                One output per lane; taps (b, d) pairs and the input span of the block sit in LDS.
//
// Both are HBM-streaming: 8 B in + 16 B out per wideband sample (all 64 channels kept), then
// 8 B in + 12 B out per channel sample.
#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr float kPif = 3.14159265358979323846f;
static constexpr int kPfbSteps = 64;        // output instants per wavefront (32 new blocks)
static constexpr int kPfbTile = 16;         // instants per LDS transpose tile

template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// value of lane (l ^ X)
template <int X>
__device__ __forceinline__ float lane_xor(float v)
{
	if constexpr (X == 32 || X == 16) return __shfl_xor(v, X, 64);
	else if constexpr (X == 8) return dppf<0x128>(v);               // row_ror:8
	else if constexpr (X == 4) return dppf<0x1B>(dppf<0x141>(v));   // half mirror, then quad reverse
	else if constexpr (X == 2) return dppf<0x4E>(v);
	else return dppf<0xB1>(v);
}

// one decimation-in-frequency stage over the lanes: pairs (l, l ^ SPAN); the lane without the bit keeps a + b, the other
// (a - b) * w, w = e^{-j 2 pi (l mod SPAN) / (2 SPAN)}.  Without selects: every lane forms partner + sg * own (sg = +1 in
// the lane without the bit, -1 in the other: exact, a product with +-1) and multiplies by (wr, wi), which is (1, 0) in the
// lanes without the bit -- for finite samples the same values as choosing between the two forms afterwards (a sum's -0
// can come out +0).  NOT for non-finite ones: dr * 1 - di * 0 turns an Inf or NaN in ONE component of a clipped or
// corrupt sample into NaNs in BOTH (Inf * 0), where a select would have passed the other component through; such a
// sample poisons the output instants its filter taps reach either way (tests/test_gpu_chan.py: the other instants stay
// exact).
template <int SPAN>
__device__ __forceinline__ void dif_stage(float &re, float &im, float sg, float wr, float wi)
{
	const float pr = lane_xor<SPAN>(re), pi = lane_xor<SPAN>(im);
	const float dr = fmaf(re, sg, pr), di = fmaf(im, sg, pi);
	re = dr * wr - di * wi;
	im = dr * wi + di * wr;
}

template <bool ROT>
__global__ __launch_bounds__(64) void k_pfb64(PfbArgs a)
{
	__shared__ float2 tile[64 * (kPfbTile + 1)];
	__shared__ int slot_of[64];                  // output slot of every channel: the write-out loop asks for one per store
	const int r = threadIdx.x;
	slot_of[r] = a.slot[r];
	const long long t0 = (long long)blockIdx.x * kPfbSteps;       // first instant of this wave (even)
	const int NB = a.n_blocks;

	// taps of this branch: even instants h[64 q - r], odd instants h[64 q + 32 - r]
	float he[kPfbMaxBlocks], ho[kPfbMaxBlocks];
#pragma unroll
	for (int q = 0; q < kPfbMaxBlocks; q++) {
		const int ie = 64 * q - r, io = 64 * q + 32 - r;
		he[q] = (q < NB && ie >= 0 && ie < a.ntaps) ? a.taps[ie] : 0.0f;
		ho[q] = (q < NB && io >= 0 && io < a.ntaps) ? a.taps[io] : 0.0f;
	}
	// twiddles of the six stages and the channel this lane ends up holding
	float wr[6], wi[6], hb[6];                   // (hb: -1 in the lanes that hold the second element of the stage's pairs, else +1)
#pragma unroll
	for (int s = 0; s < 6; s++) {
		const int span = 32 >> s;
		const float ang = -kPif * (float)(r & (span - 1)) / (float)span;
		const bool hi = (r & span) != 0;
		wr[s] = hi ? __cosf(ang) : 1.0f;
		wi[s] = hi ? __sinf(ang) : 0.0f;
		hb[s] = hi ? -1.0f : 1.0f;
	}
	const int chan = (int)(__brev((unsigned)r) >> 26);
	const int slot = a.slot[chan];

	// window of the lane's samples: w[q] = x[64 (b - q) + r] for the current block b
	float2 w[kPfbMaxBlocks];
	const long long b0 = t0 / 2;                                   // block of the first instant
	auto load_block = [&](long long b) -> float2 {
		const long long s = 64 * b + r;
		float2 v = make_float2(0.f, 0.f);
		if (b >= 0 && s < a.n_in) {
			v = a.x[s];
			if (ROT) {
				// e^{j rotation s}: the angle reduced in double so that long captures keep their phase
				const double ph = (double)a.rotation * (double)s;
				const float fr = (float)(ph - 6.283185307179586 * rint(ph * 0.15915494309189535));
				float sn, cs;
				__sincosf(fr, &sn, &cs);
				v = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
			}
		}
		return v;
	};
#pragma unroll
	for (int q = 1; q < kPfbMaxBlocks; q++)
		w[q] = load_block(b0 - q);

	float2 nxt = load_block(b0);
	for (int tt = 0; tt < kPfbSteps; tt += kPfbTile) {
#pragma unroll 1
		for (int u = 0; u < kPfbTile; u += 2) {
			const long long t = t0 + tt + u;
			// new block: it serves instants t (even) and t + 1; the one after it is already on its way
			w[0] = nxt;
			nxt = load_block(t / 2 + 1);
			float er = 0.f, ei = 0.f, orr = 0.f, oi = 0.f;
#pragma unroll
			for (int q = 0; q < kPfbMaxBlocks; q++) {
				er = fmaf(he[q], w[q].x, er);
				ei = fmaf(he[q], w[q].y, ei);
				orr = fmaf(ho[q], w[q].x, orr);
				oi = fmaf(ho[q], w[q].y, oi);
			}
#pragma unroll
			for (int q = kPfbMaxBlocks - 1; q > 0; q--)
				w[q] = w[q - 1];
			dif_stage<32>(er, ei, hb[0], wr[0], wi[0]);
			dif_stage<32>(orr, oi, hb[0], wr[0], wi[0]);
			dif_stage<16>(er, ei, hb[1], wr[1], wi[1]);
			dif_stage<16>(orr, oi, hb[1], wr[1], wi[1]);
			dif_stage<8>(er, ei, hb[2], wr[2], wi[2]);
			dif_stage<8>(orr, oi, hb[2], wr[2], wi[2]);
			dif_stage<4>(er, ei, hb[3], wr[3], wi[3]);
			dif_stage<4>(orr, oi, hb[3], wr[3], wi[3]);
			dif_stage<2>(er, ei, hb[4], wr[4], wi[4]);
			dif_stage<2>(orr, oi, hb[4], wr[4], wi[4]);
			dif_stage<1>(er, ei, hb[5], wr[5], wi[5]);
			dif_stage<1>(orr, oi, hb[5], wr[5], wi[5]);
			tile[chan * (kPfbTile + 1) + u] = make_float2(er, ei);
			tile[chan * (kPfbTile + 1) + u + 1] = make_float2(orr, oi);
		}
		WSYNC();
		// write-out: 16 instants x 8 bytes = one 128-byte run per kept channel
		for (int e = r; e < 64 * kPfbTile; e += 64) {
			const int c = e / kPfbTile, u = e % kPfbTile;
			const int sl = slot_of[c];
			const long long t = t0 + tt + u;
			if (sl >= 0 && t < a.T)
				a.y[(long long)sl * a.T + t] = tile[c * (kPfbTile + 1) + u];
		}
		WSYNC();
	}
	(void)slot;
}

// ---------------------------------------------------------------------------
// Any even channel count (the recorder derives n_chans from the sample rate: 32 at 1.0 Msps, 40 at 1.25,
// 80 at 2.5, 128 at 4.0 ...; gmr1_rx_sdr.py:408).  Same definition, two plain stages per tile of 16
// output instants: the polyphase sums v_t[r] = sum_b x[b M + r] h[t D - b M - r] into LDS (reads of x are
// coalesced along r), then a direct DFT for the SELECTED channels only, Y_k[t] = sum_r v_t[r] W^{k r},
// with the M twiddles in LDS.  O(M) per kept channel and instant instead of the lane FFT's O(log M) per
// channel, so k_pfb64 stays the fast path of the 2.0 Msps plan; this one is the general fallback.
// ---------------------------------------------------------------------------
static constexpr int kAnyTile = 16;          // instants per work-group
static constexpr int kAnyThreads = 256;

template <bool ROT>
__global__ __launch_bounds__(kAnyThreads) void k_pfb_any(PfbArgs a)
{
	extern __shared__ float2 lds_any[];
	const int M = a.n_chans, D = M / 2, ld = M + 1;
	float2 *v = lds_any;                       // kAnyTile x (M + 1)
	float2 *tw = lds_any + kAnyTile * ld;      // M twiddles e^{-j 2 pi i / M}
	const int tid = threadIdx.x;
	const long long t0 = (long long)blockIdx.x * kAnyTile;

	for (int i = tid; i < M; i += kAnyThreads) {
		float sn, cs;
		sincospif(-2.0f * (float)i / (float)M, &sn, &cs);
		tw[i] = make_float2(cs, sn);
	}
	for (int e = tid; e < kAnyTile * M; e += kAnyThreads) {
		const int u = e / M, r = e % M;
		const long long t = t0 + u;
		float ar = 0.f, ai = 0.f;
		if (t < a.T) {
			const long long td = t * D;
			// newest sample of branch r at or before t D (floor division: t D - r may be negative)
			long long blk = (td - r >= 0) ? (td - r) / M : -1;
			for (int q = 0; q < a.n_blocks; q++, blk--) {
				const long long sidx = blk * M + r;
				const long long ti = td - sidx;
				if (blk < 0 || ti >= a.ntaps)
					break;
				if (sidx >= a.n_in)
					continue;
				float2 x = a.x[sidx];
				if (ROT) {
					const double ph = (double)a.rotation * (double)sidx;
					const float fr = (float)(ph - 6.283185307179586 * rint(ph * 0.15915494309189535));
					float sn, cs;
					__sincosf(fr, &sn, &cs);
					x = make_float2(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
				}
				const float h = a.taps[ti];
				ar = fmaf(h, x.x, ar);
				ai = fmaf(h, x.y, ai);
			}
		}
		v[u * ld + r] = make_float2(ar, ai);
	}
	__syncthreads();
	for (int e = tid; e < a.n_sel * kAnyTile; e += kAnyThreads) {
		const int c = e / kAnyTile, u = e % kAnyTile;
		const long long t = t0 + u;
		if (t >= a.T)
			continue;
		const int k = a.sel[c];
		const float2 *vr = v + u * ld;
		float yr = 0.f, yi = 0.f;
		int idx = 0;                             // (k r) mod M
		for (int r = 0; r < M; r++) {
			const float2 w = tw[idx], s = vr[r];
			yr = fmaf(s.x, w.x, fmaf(-s.y, w.y, yr));
			yi = fmaf(s.x, w.y, fmaf(s.y, w.x, yi));
			idx += k;
			idx = idx >= M ? idx - M : idx;
		}
		a.y[(long long)c * a.T + t] = make_float2(yr, yi);
	}
}

hipError_t launch_pfb(const PfbArgs &a, hipStream_t stream)
{
	if (a.T <= 0)
		return hipSuccess;
	if (a.n_chans != 64) {
		if (a.n_chans < 2 || a.n_chans > kPfbMaxChans || (a.n_chans & 1) || !a.sel || a.n_sel <= 0)
			return hipErrorInvalidValue;
		const long long grid = (a.T + kAnyTile - 1) / kAnyTile;
		const size_t lds = (size_t)(kAnyTile * (a.n_chans + 1) + a.n_chans) * sizeof(float2);
		if (a.rotation != 0.0f)
			hipLaunchKernelGGL(k_pfb_any<true>, dim3((unsigned)grid), dim3(kAnyThreads), lds, stream, a);
		else
			hipLaunchKernelGGL(k_pfb_any<false>, dim3((unsigned)grid), dim3(kAnyThreads), lds, stream, a);
		return hipGetLastError();
	}
	const long long grid = (a.T + kPfbSteps - 1) / kPfbSteps;
	if (a.rotation != 0.0f)
		hipLaunchKernelGGL(k_pfb64<true>, dim3((unsigned)grid), dim3(64), 0, stream, a);
	else
		hipLaunchKernelGGL(k_pfb64<false>, dim3((unsigned)grid), dim3(64), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// arbitrary resampler
//
// The phase step num/den is rational, so the (filter, fraction) pair of output n repeats with period
// P = 32 den / gcd(num, 32 den) outputs = Q = num / gcd(...) inputs (936 outputs = 625 inputs at
// sps 4).  A lane therefore owns ONE output phase p for a run of periods: its 30 effective taps
// e[k] = b_j[k] + frac d_j[k] are computed once and stay in registers; per output it only reads its
// 30 input samples from a wave-private LDS window (the 64 phases of a wave span ~43 + 30 inputs),
// which the wave stages with coalesced loads one period ahead.  Single-wavefront work-groups, no
// s_barrier; outputs of a wave are 64 consecutive samples = one 512-byte store.
// ---------------------------------------------------------------------------
// taps per filter: 30 for the 941-tap / 32-phase bank of every plan that raises the rate; the direct mode at 1.0 Msps
// resamples DOWN (rate 0.468) and its root-raised cosine is 95 taps per phase long -- a second instantiation
static constexpr int kRsTapsShort = 30, kRsTapsLong = 96;
static constexpr int kRsPeriods = 32;        // periods one wave walks
static constexpr int kRsWinTight = 128, kRsWinShort = 256, kRsWinLong = 512;     // LDS window (samples) per wave, >= span

// acc += e (s.re, s.im) for the real tap e that sits in the LOW (HI = false) or HIGH half of the register pair `ee`: one
// v_pk_fma_f32 with the half picked by the operand selects.  The compiler's own code for the two scalar chains
// orr = fma(e, s.re, orr), oi = fma(e, s.im, oi) is the same packed FMA, but on a pair (e, e) it builds for every tap -- 60
// registers of taps instead of 30 and five waves per SIMD instead of eight, with two LDS reads in flight per wave.
// (s_nop: a packed result needs one wait state before its next use.)
typedef float rs_v2f __attribute__((ext_vector_type(2)));
template <bool HI>
__device__ __forceinline__ void rs_mac(rs_v2f &acc, rs_v2f ee, rs_v2f s)
{
	if constexpr (HI)
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\ts_nop 0" : "+v"(acc) : "v"(ee), "v"(s));
	else
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\ts_nop 0" : "+v"(acc) : "v"(ee), "v"(s));
}

// EXT: the instantiation that can rotate its input (the pre-resampler of an off-grid capture) and write polyphase-planar
// output; the plain one does neither and keeps its registers
// (the 30-tap instantiations are compiled for at least five waves per SIMD: the compiler otherwise keeps all thirty samples
// of a period in flight at once)
template <int kRsTaps, bool EXT>
constexpr int kRsWaves = kRsTaps <= 32 ? 5 : 2;
template <int kRsTaps, int kRsWin, bool EXT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kRsWaves<kRsTaps, EXT>)))
void k_resamp(ResampArgs a, long long P, long long Q, int span)
{
	static_assert(kRsTaps % 2 == 0, "taps are kept two to a register pair");
	// D periods are handled as one block: their windows are written to LDS together, the next block's windows are requested,
	// then the block's arithmetic runs and its outputs are stored.  On this ISA loads and stores share one counter and complete
	// out of order with respect to each other, so a wait for a window is a wait for EVERYTHING outstanding, the stores just
	// issued included; in the block form whatever the wait covers was issued a whole block of arithmetic earlier.  (Two periods
	// per block for the 128-sample window = every plan that raises the rate: 80 registers, six waves; four periods per block
	// cost the sixth wave and gain nothing, and an aligned ds_read_b128 per two taps out of a doubled window is no faster than
	// the two 8-byte reads: what bounds the kernel after this is LDS bytes, 240 per output.)
	constexpr int D = kRsWin <= 128 ? 2 : 1;
	__shared__ float2 xsb[D][kRsWin];
	constexpr int NH = kRsWin / 64;
	const int lane = threadIdx.x;
	const int sl = blockIdx.y;
	const long long p0 = (long long)blockIdx.x * 64;
	const long long p = p0 + lane;
	const bool live = p < P;
	// phase of this lane's outputs: exact integer arithmetic, once
	const long long Np = (long long)a.j0 * a.den + (live ? p : p0) * a.num;
	const long long fl = Np / a.den;
	const float frac = (float)(Np - fl * a.den) / (float)a.den;
	const int j = (int)(fl % a.nfilt);
	const long long ip = fl / a.nfilt;                                  // input index of period 0
	const long long N0 = (long long)a.j0 * a.den + p0 * a.num;
	const long long i_first = (N0 / a.den) / a.nfilt - (kRsTaps - 1);   // first input the wave needs (period 0)
	const int base = (int)(ip - i_first);                               // window index of this lane's newest sample
	rs_v2f e2[kRsTaps / 2];                                             // taps 2 i, 2 i + 1
#pragma unroll
	for (int k = 0; k < kRsTaps; k += 2) {
		const float2 b0 = a.bank[j * kRsTaps + k], b1 = a.bank[j * kRsTaps + k + 1];
		e2[k / 2] = (rs_v2f){fmaf(frac, b0.y, b0.x), fmaf(frac, b1.y, b1.x)};
	}
	const float2 *__restrict__ y = a.y + (long long)sl * a.T;
	float2 *__restrict__ out = a.out + (long long)sl * a.out_stride;
	const long long m0 = (long long)blockIdx.z * kRsPeriods;

	// window of period m: inputs i_first + m Q + [0, span)
	static_assert(kRsPeriods % D == 0, "the period loop runs in blocks of D");
	float2 ring[D][NH];
	auto fetch = [&](long long m, float2 (&nx)[NH]) {
#pragma unroll
		for (int h = 0; h < NH; h++) {
			const int w = lane + 64 * h;
			const long long s = i_first + m * Q + w;
			nx[h] = (w < span && s >= 0 && s < a.T) ? y[s] : make_float2(0.f, 0.f);
			if (EXT && a.rotation != 0.0f) {
				// e^{j rotation s}, the angle reduced in double as in k_pfb64 (long captures keep their phase)
				const double ph = (double)a.rotation * (double)s;
				const float fr = (float)(ph - 6.283185307179586 * rint(ph * 0.15915494309189535));
				float sn, cs;
				__sincosf(fr, &sn, &cs);
				nx[h] = make_float2(nx[h].x * cs - nx[h].y * sn, nx[h].x * sn + nx[h].y * cs);
			}
		}
	};
	// polyphase-planar output: flat index g = slot out_stride + n -> plane g % sps, place g / sps; n advances by P per
	// period, so plane and place are carried along instead of divided out per sample
	long long pl_q = 0, pl_dq = 0;
	int pl_r = 0, pl_dr = 0;
	if (EXT && a.planar_sps > 0) {
		const long long g0 = (long long)sl * a.out_stride + m0 * P + p;
		pl_q = g0 / a.planar_sps;
		pl_r = (int)(g0 % a.planar_sps);
		pl_dq = P / a.planar_sps;
		pl_dr = (int)(P % a.planar_sps);
	}
#pragma unroll
	for (int d = 0; d < D; d++)
		fetch(m0 + d, ring[d]);
	for (int mm = 0; mm < kRsPeriods; mm += D) {
		if ((m0 + mm) * P >= a.n_out)
			return;
		WSYNC();
#pragma unroll
		for (int d = 0; d < D; d++)
#pragma unroll
			for (int h = 0; h < NH; h++)
				if (lane + 64 * h < span)
					xsb[d][lane + 64 * h] = ring[d][h];
		WSYNC();
		if (mm + D < kRsPeriods) {
#pragma unroll
			for (int d = 0; d < D; d++)
				fetch(m0 + mm + D + d, ring[d]);    // the next block's windows travel during this block's arithmetic
		}
		rs_v2f res[D];
#pragma unroll
		for (int d = 0; d < D; d++) {
			const float2 *xs = xsb[d];
			// (tap order 0, 1, 2, ... as the scalar chains had it)
			rs_v2f acc = {0.f, 0.f};
#pragma unroll
			for (int k = 0; k < kRsTaps; k += 2) {
				const float2 s0 = xs[base - k], s1 = xs[base - k - 1];
				rs_mac<false>(acc, e2[k / 2], (rs_v2f){s0.x, s0.y});
				rs_mac<true>(acc, e2[k / 2], (rs_v2f){s1.x, s1.y});
			}
			res[d] = acc;
		}
#pragma unroll
		for (int d = 0; d < D; d++) {
			const long long n = (m0 + mm + d) * P + p;
			if (live && n < a.n_out) {
				if (EXT && a.planar_sps > 0)
					a.out[(long long)pl_r * a.plane_stride + pl_q] = make_float2(res[d].x, res[d].y);
				else
					out[n] = make_float2(res[d].x, res[d].y);
			}
			if constexpr (EXT) {
				pl_q += pl_dq;
				pl_r += pl_dr;
				if (pl_r >= a.planar_sps && a.planar_sps > 0) {
					pl_r -= a.planar_sps;
					pl_q++;
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------
// k_ddc_fir -- decimating FIR of the direct mode: y[s][m] = rot_s(m) * sum_k taps[s][k] x_s[m D - k].
// One work-group = 256 consecutive outputs of one carrier: the 256 D + ntaps input samples they span are staged in
// LDS with coalesced loads (the windows of neighbouring outputs overlap by ntaps - D samples), the carrier's taps too;
// every thread then runs its own dot product out of LDS.  Stage 1 reads the one wideband stream for every carrier
// with that carrier's complex taps  taps[k] exp(+j 2 pi f k / fs)  and turns the output by exp(-j 2 pi f m D / fs)
// (freq_xlating_fir_filter_ccc: the same as mixing x down by f first), the angle reduced in double so that a minute
// of capture keeps its phase; stage 2 has real taps and no rotation.
// ---------------------------------------------------------------------------
static constexpr int kDdcOut = 256;

__global__ __launch_bounds__(256) void k_ddc_fir(DdcFirArgs a)
{
	extern __shared__ __align__(16) unsigned char ddc_lds[];
	float2 *tp = reinterpret_cast<float2 *>(ddc_lds);                     // ntaps
	float2 *xs = tp + a.ntaps;                                           // kDdcOut * decim + ntaps
	const int s = blockIdx.y, tid = threadIdx.x;
	const long long m0 = (long long)blockIdx.x * kDdcOut;
	const float2 *__restrict__ x = a.x + (long long)s * a.in_stride;
	const int span = kDdcOut * a.decim + a.ntaps;
	const long long first = m0 * a.decim - (a.ntaps - 1);                 // oldest sample the block touches
	for (int i = tid; i < a.ntaps; i += 256)
		tp[i] = a.taps[(long long)s * a.ntaps + i];
	for (int i = tid; i < span; i += 256) {
		const long long n = first + i;
		xs[i] = (n >= 0 && n < a.n_in) ? x[n] : make_float2(0.f, 0.f);
	}
	__syncthreads();
	const long long m = m0 + tid;
	if (m >= a.n_out)
		return;
	// x[m D - k] = xs[tid D + ntaps - 1 - k]
	const float2 *w = xs + tid * a.decim + a.ntaps - 1;
	float ar = 0.f, ai = 0.f;
	for (int k = 0; k < a.ntaps; k++) {
		const float2 h = tp[k], v = w[-k];
		ar = fmaf(h.x, v.x, fmaf(-h.y, v.y, ar));
		ai = fmaf(h.x, v.y, fmaf(h.y, v.x, ai));
	}
	if (a.rot) {
		double ph = (double)m * a.rot[s];
		ph -= floor(ph);
		float sn, cs;
		__sincosf(-2.0f * kPif * (float)ph, &sn, &cs);
		const float r = ar * cs - ai * sn, q = ar * sn + ai * cs;
		ar = r;
		ai = q;
	}
	a.y[(long long)s * a.n_out + m] = make_float2(ar, ai);
}

hipError_t launch_ddc_fir(const DdcFirArgs &a, hipStream_t stream)
{
	if (a.n_out <= 0 || a.n_sel <= 0)
		return hipSuccess;
	if (a.ntaps < 1 || a.ntaps > kDdcMaxTaps || a.decim < 1 || a.decim > 64)
		return hipErrorInvalidValue;
	const size_t lds = ((size_t)a.ntaps + (size_t)kDdcOut * a.decim + a.ntaps) * sizeof(float2);
	if (lds > 150 * 1024)
		return hipErrorInvalidValue;
	if (lds > 64 * 1024)
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ddc_fir), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k_ddc_fir, dim3((unsigned)((a.n_out + kDdcOut - 1) / kDdcOut), (unsigned)a.n_sel), dim3(256), lds, stream, a);
	return hipGetLastError();
}

hipError_t launch_resamp(const ResampArgs &a, hipStream_t stream)
{
	if (a.n_out <= 0 || a.n_slots <= 0)
		return hipSuccess;
	const bool lng = a.tpf == kRsTapsLong;
	if (a.tpf != kRsTapsShort && !lng)
		return hipErrorInvalidValue;
	const int taps = lng ? kRsTapsLong : kRsTapsShort, win = lng ? kRsWinLong : kRsWinShort;
	long long g = a.num, b = a.den * a.nfilt;
	while (b) { const long long t = g % b; g = b; b = t; }
	const long long P = a.den * a.nfilt / g, Q = a.num / g;
	// inputs the 64 phases of a wave can touch
	const int span = (int)((63 * a.num) / (a.den * a.nfilt)) + taps + 2;
	if (span > win)
		return hipErrorInvalidValue;
	const long long periods = (a.n_out + P - 1) / P;
	const unsigned gx = (unsigned)((P + 63) / 64), gz = (unsigned)((periods + kRsPeriods - 1) / kRsPeriods);
	const bool ext = a.rotation != 0.0f || a.planar_sps > 0;
	if (lng && ext)
		return hipErrorInvalidValue;             // (the long bank is the direct mode's: neither option reaches it)
	if (lng)
		hipLaunchKernelGGL((k_resamp<kRsTapsLong, kRsWinLong>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (ext && span <= kRsWinTight)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinTight, true>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (ext)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinShort, true>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (span <= kRsWinTight)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinTight>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinShort>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	return hipGetLastError();
}

}  // namespace gmr1
