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

// one decimation-in-frequency stage over the lanes: pairs (l, l ^ SPAN); the lane without the bit
// keeps a + b, the other (a - b) * w, w = e^{-j 2 pi (l mod SPAN) / (2 SPAN)} (held in wr / wi)
template <int SPAN>
__device__ __forceinline__ void dif_stage(float &re, float &im, bool hi, float wr, float wi)
{
	const float pr = lane_xor<SPAN>(re), pi = lane_xor<SPAN>(im);
	const float sr = re + pr, si = im + pi;
	const float dr = pr - re, di = pi - im;         // (a - b) seen from the hi lane: partner is a
	const float tr = dr * wr - di * wi, ti = dr * wi + di * wr;
	re = hi ? tr : sr;
	im = hi ? ti : si;
}

__global__ __launch_bounds__(64) void k_pfb64(PfbArgs a)
{
	__shared__ float2 tile[64 * (kPfbTile + 1)];
	const int r = threadIdx.x;
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
	float wr[6], wi[6];
	bool hb[6];
#pragma unroll
	for (int s = 0; s < 6; s++) {
		const int span = 32 >> s;
		const float ang = -kPif * (float)(r & (span - 1)) / (float)span;
		wr[s] = __cosf(ang);
		wi[s] = __sinf(ang);
		hb[s] = (r & span) != 0;
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
			if (a.rotation != 0.0f) {
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

	for (int tt = 0; tt < kPfbSteps; tt += kPfbTile) {
#pragma unroll 1
		for (int u = 0; u < kPfbTile; u += 2) {
			const long long t = t0 + tt + u;
			// new block: it serves instants t (even) and t + 1
			w[0] = load_block(t / 2);
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
			const int sl = a.slot[c];
			const long long t = t0 + tt + u;
			if (sl >= 0 && t < a.T)
				a.y[(long long)sl * a.T + t] = tile[c * (kPfbTile + 1) + u];
		}
		WSYNC();
	}
	(void)slot;
}

hipError_t launch_pfb(const PfbArgs &a, hipStream_t stream)
{
	if (a.T <= 0)
		return hipSuccess;
	const long long grid = (a.T + kPfbSteps - 1) / kPfbSteps;
	hipLaunchKernelGGL(k_pfb64, dim3((unsigned)grid), dim3(64), 0, stream, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// arbitrary resampler
// ---------------------------------------------------------------------------
static constexpr int kRsBlock = 256;

__global__ __launch_bounds__(kRsBlock) void k_resamp(ResampArgs a, int span)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float2 *bank = reinterpret_cast<float2 *>(lds_raw);                   // nfilt x tpf (b, d)
	float2 *xs = bank + a.nfilt * a.tpf;                                  // input span of this block
	const int tid = threadIdx.x;
	const int sl = blockIdx.y;
	const long long n0 = (long long)blockIdx.x * kRsBlock;
	for (int i = tid; i < a.nfilt * a.tpf; i += kRsBlock)
		bank[i] = a.bank[i];
	// input index of the block's first output; every output of the block reads [i - tpf + 1, i]
	const long long N0 = (long long)a.j0 * a.den + n0 * a.num;
	const long long i_first = (N0 / a.den) / a.nfilt - (a.tpf - 1);
	const float2 *__restrict__ y = a.y + (long long)sl * a.T;
	for (int i = tid; i < span; i += kRsBlock) {
		const long long s = i_first + i;
		xs[i] = (s >= 0 && s < a.T) ? y[s] : make_float2(0.f, 0.f);
	}
	__syncthreads();
	const long long n = n0 + tid;
	if (n >= a.n_out)
		return;
	const long long N = (long long)a.j0 * a.den + n * a.num;
	const long long fl = N / a.den;
	const float frac = (float)(N - fl * a.den) / (float)a.den;
	const int j = (int)(fl % a.nfilt);
	const int base = (int)(fl / a.nfilt - i_first);                        // xs index of input sample i
	const float2 *bj = bank + j * a.tpf;
	float orr = 0.f, oi = 0.f;
	for (int k = 0; k < a.tpf; k++) {
		const float2 bd = bj[k];
		const float e = fmaf(frac, bd.y, bd.x);
		const float2 s = xs[base - k];
		orr = fmaf(e, s.x, orr);
		oi = fmaf(e, s.y, oi);
	}
	a.out[(long long)sl * a.out_stride + n] = make_float2(orr, oi);
}

hipError_t launch_resamp(const ResampArgs &a, hipStream_t stream)
{
	if (a.n_out <= 0 || a.n_slots <= 0)
		return hipSuccess;
	// inputs one block of outputs can touch: (kRsBlock - 1) num / (den nfilt) + tpf, rounded up generously
	const int span = (int)(((long long)(kRsBlock - 1) * a.num) / (a.den * a.nfilt)) + a.tpf + 3;
	const size_t lds = ((size_t)a.nfilt * a.tpf + (size_t)span) * sizeof(float2);
	const long long gx = (a.n_out + kRsBlock - 1) / kRsBlock;
	hipLaunchKernelGGL(k_resamp, dim3((unsigned)gx, (unsigned)a.n_slots), dim3(kRsBlock), lds, stream, a, span);
	return hipGetLastError();
}

}  // namespace gmr1
