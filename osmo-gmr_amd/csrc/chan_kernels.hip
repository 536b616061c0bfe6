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
//                All of its arithmetic is packed single precision (pf_fir, pf_stage).
//   k_pfb_any  : the same filterbank for any even channel count (direct DFT of the selected channels).
//   k_resamp   : out[n] = sum_k (b_j[k] + frac d_j[k]) y[i - k],  phase = j0 + n num/den in 1/32 input
//                samples kept in integers (j = phase mod 32, i = phase / 32, frac = remainder / den).
//                A lane owns one output phase of the plan's period: its effective taps in registers, its
//                samples from a wave-private LDS window.
//   k_resamp2  : the same sums with two consecutive outputs per lane sharing their window (aligned
//                16-byte LDS reads) and the windows brought in by LDS-DMA into a ring, seven periods
//                ahead; every plan that raises the rate.
//   k_ddc_fir  : decimating FIR of the direct mode.
//
// HBM-streaming by construction: 8 B in + 16 B out per wideband sample (all 64 channels kept), then
// 8 B in + 12 B out per channel sample.
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include "gmr1_dev.h"
#include "profile_env.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr float kPif = 3.14159265358979323846f;
static constexpr int kPfbSteps = 64;        // output instants per wavefront (32 new blocks; 128 / 256: 5 / 22 % slower, r06aq)
static constexpr int kPfbTile = 16;         // instants per LDS transpose tile
static constexpr int kPfbWb = 4;            // tile reads answered together in k_pfb64's write-out

// (every lane of these permutations has a source lane: no "old" value is needed, and none is set up)
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
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

// ---- packed single precision (v_pk_*_f32: two multiply-adds per lane and instruction, the rate the vector peak is quoted
// at).  The filterbank's arithmetic is complex-by-real and complex-by-complex throughout, so every step is a pair: with
// scalar instructions the kernel issued 6 455 vector instructions per wave at 88 % VALU busy.  The operand selects pick the
// half of a register pair a lane-wide "scalar" (a tap, a sign, a twiddle component) sits in, so none of them is duplicated.
// A packed result needs one wait state before its next use: the helpers take the even and the odd instant together, each
// one's instruction is the other's wait state.
typedef float pf_v2f __attribute__((ext_vector_type(2)));

// ae += taps.lo * w ; ao += taps.hi * w          (taps = (even instant's tap, odd instant's tap))
__device__ __forceinline__ void pf_fir(pf_v2f &ae, pf_v2f &ao, pf_v2f taps, pf_v2f w)
{
	asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
	    "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
	    : "+v"(ae), "+v"(ao) : "v"(taps), "v"(w));
}

// one decimation-in-frequency stage over the lanes: pairs (l, l ^ SPAN); the lane without the bit keeps a + b, the other
// (a - b) * w, w = e^{-j 2 pi (l mod SPAN) / (2 SPAN)}.  Without selects: every lane forms partner + sg * own (sg = +1 in
// the lane without the bit, -1 in the other: exact, a product with +-1) and multiplies by (wr, wi), which is (1, 0) in the
// lanes without the bit -- for finite samples the same values as choosing between the two forms afterwards (a sum's -0
// can come out +0).  NOT for non-finite ones: dr * 1 - di * 0 turns an Inf or NaN in ONE component of a clipped or
// corrupt sample into NaNs in BOTH (Inf * 0), where a select would have passed the other component through; such a
// sample poisons the output instants its filter taps reach either way (tests/test_gpu_chan.py: the other instants stay
// exact).
// the stage's butterfly and twiddle for both instants: d = partner + sg * own ; out = d * (wr + j wi)
//   sgp: the pair holding this stage's sign in half SH ; w: (wr, wi)
//   out.re = d.re wr - d.im wi, out.im = d.im wr + d.re wi, as  t = d * wr ; out = (-d.im, d.re) * wi + t
template <int SH>
__device__ __forceinline__ void pf_stage(pf_v2f &xe, pf_v2f &xo, pf_v2f pe, pf_v2f po, pf_v2f sgp, pf_v2f w)
{
	pf_v2f de, dq, te, to;
	if constexpr (SH == 0)
		asm("v_pk_fma_f32 %0, %2, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
		    "v_pk_fma_f32 %1, %3, %4, %6 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
		    : "=&v"(de), "=&v"(dq) : "v"(xe), "v"(xo), "v"(sgp), "v"(pe), "v"(po));
	else
		asm("v_pk_fma_f32 %0, %2, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
		    "v_pk_fma_f32 %1, %3, %4, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
		    : "=&v"(de), "=&v"(dq) : "v"(xe), "v"(xo), "v"(sgp), "v"(pe), "v"(po));
	asm("v_pk_mul_f32 %0, %2, %4 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
	    "v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[1,0]"
	    : "=&v"(te), "=&v"(to) : "v"(de), "v"(dq), "v"(w));
	asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
	    "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
	    : "+v"(te), "+v"(to) : "v"(de), "v"(dq), "v"(w));
	xe = te;
	xo = to;
}

// ... and the last stage (pairs of neighbouring lanes): its twiddle is 1 in every lane
template <int SH>
__device__ __forceinline__ void pf_stage_last(pf_v2f &xe, pf_v2f &xo, pf_v2f pe, pf_v2f po, pf_v2f sgp)
{
	pf_v2f de, dq;
	if constexpr (SH == 0)
		asm("v_pk_fma_f32 %0, %2, %4, %5 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
		    "v_pk_fma_f32 %1, %3, %4, %6 op_sel:[0,0,0] op_sel_hi:[1,0,1]"
		    : "=&v"(de), "=&v"(dq) : "v"(xe), "v"(xo), "v"(sgp), "v"(pe), "v"(po));
	else
		asm("v_pk_fma_f32 %0, %2, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
		    "v_pk_fma_f32 %1, %3, %4, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
		    : "=&v"(de), "=&v"(dq) : "v"(xe), "v"(xo), "v"(sgp), "v"(pe), "v"(po));
	xe = de;
	xo = dq;
}

template <int SPAN>
__device__ __forceinline__ pf_v2f pf_partner(pf_v2f v)
{
	return (pf_v2f){lane_xor<SPAN>(v.x), lane_xor<SPAN>(v.y)};
}

template <bool ROT>
__global__ __launch_bounds__(64) void k_pfb64(PfbArgs a)
{
	__shared__ float2 tile[64 * (kPfbTile + 1)];
	const int r = threadIdx.x;
	const long long t0 = (long long)blockIdx.x * kPfbSteps;       // first instant of this wave (even)
	const int NB = a.n_blocks;

	// taps of this branch: even instants h[64 q - r], odd instants h[64 q + 32 - r]
	pf_v2f heo[kPfbMaxBlocks];                   // (even instant's tap, odd instant's tap) of block q
#pragma unroll
	for (int q = 0; q < kPfbMaxBlocks; q++) {
		const int ie = 64 * q - r, io = 64 * q + 32 - r;
		heo[q] = (pf_v2f){(q < NB && ie >= 0 && ie < a.ntaps) ? a.taps[ie] : 0.0f,
		                  (q < NB && io >= 0 && io < a.ntaps) ? a.taps[io] : 0.0f};
	}
	// twiddles of the six stages and the channel this lane ends up holding
	pf_v2f tw[6];                                // (wr, wi)
	float hb[6];                                 // (hb: -1 in the lanes that hold the second element of the stage's pairs, else +1)
#pragma unroll
	for (int s = 0; s < 6; s++) {
		const int span = 32 >> s;
		const float ang = -kPif * (float)(r & (span - 1)) / (float)span;
		const bool hi = (r & span) != 0;
		tw[s] = (pf_v2f){hi ? __cosf(ang) : 1.0f, hi ? __sinf(ang) : 0.0f};
		hb[s] = hi ? -1.0f : 1.0f;
	}
	const pf_v2f sg01 = {hb[0], hb[1]}, sg23 = {hb[2], hb[3]}, sg45 = {hb[4], hb[5]};
	const int chan = (int)(__brev((unsigned)r) >> 26);
	const int slot = a.slot[chan];
	// output slots of the sixteen channels this lane writes out (16 (r >> 4) + i), a byte each, 0xff = not kept
	unsigned slpk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
	for (int i = 0; i < 16; i++) {
		const int sv = a.slot[16 * (r >> 4) + i];
		slpk[i >> 2] |= (sv < 0 ? 0xffu : (unsigned)sv & 0xffu) << (8 * (i & 3));
	}

	// window of the lane's samples: w[q] = x[64 (b - q) + r] for the current block b
	pf_v2f w[kPfbMaxBlocks];
	const long long b0 = t0 / 2;                                   // block of the first instant
	auto load_block = [&](long long b) -> pf_v2f {
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
		return (pf_v2f){v.x, v.y};
	};
#pragma unroll
	for (int q = 1; q < kPfbMaxBlocks; q++)
		w[q] = load_block(b0 - q);

	// (two blocks on their way: at half the instruction count a block's arithmetic no longer covers a trip to memory)
	pf_v2f nxt = load_block(b0), nxt2 = load_block(b0 + 1);
	constexpr int kBlk = kPfbTile / 2;           // new blocks per tile
	for (int tt = 0; tt < kPfbSteps; tt += kPfbTile) {
		// The tile's eight blocks unrolled over ONE register array W: the window of block k is W[kBlk - 1 - k + q], the new
		// block goes to W[kBlk - 1 - k] -- static indices, nothing is shifted per block; the window moves back up once per tile.
		pf_v2f W[kBlk + kPfbMaxBlocks - 1];
#pragma unroll
		for (int q = 1; q < kPfbMaxBlocks; q++)
			W[kBlk - 1 + q] = w[q];
#pragma unroll
		for (int k = 0; k < kBlk; k++) {
			const int u = 2 * k;
			const long long t = t0 + tt + u;
			// new block: it serves instants t (even) and t + 1
			W[kBlk - 1 - k] = nxt;
			nxt = nxt2;
			nxt2 = load_block(t / 2 + 2);
			pf_v2f xe = {0.f, 0.f}, xo = {0.f, 0.f};
#pragma unroll
			for (int q = 0; q < kPfbMaxBlocks; q++)
				pf_fir(xe, xo, heo[q], W[kBlk - 1 - k + q]);
			pf_stage<0>(xe, xo, pf_partner<32>(xe), pf_partner<32>(xo), sg01, tw[0]);
			pf_stage<1>(xe, xo, pf_partner<16>(xe), pf_partner<16>(xo), sg01, tw[1]);
			pf_stage<0>(xe, xo, pf_partner<8>(xe), pf_partner<8>(xo), sg23, tw[2]);
			pf_stage<1>(xe, xo, pf_partner<4>(xe), pf_partner<4>(xo), sg23, tw[3]);
			pf_stage<0>(xe, xo, pf_partner<2>(xe), pf_partner<2>(xo), sg45, tw[4]);
			pf_stage_last<1>(xe, xo, pf_partner<1>(xe), pf_partner<1>(xo), sg45);
			// (a half-wave's channels are all even or all odd -- bit-reversed lanes --: their 136-byte rows start 4 k banks
			// apart, k = 0 ... 31, so rows k and k + 16 would meet; the upper half of the channels sits one column on, in the
			// row's spare seventeenth place)
			tile[chan * (kPfbTile + 1) + (chan >> 5) + u] = make_float2(xe.x, xe.y);
			tile[chan * (kPfbTile + 1) + (chan >> 5) + u + 1] = make_float2(xo.x, xo.y);
		}
		// (the next tile's "previous blocks": what the last block's window held, one further back)
#pragma unroll
		for (int q = 1; q < kPfbMaxBlocks; q++)
			w[q] = W[q - 1];
		WSYNC();
		// write-out: 16 instants x 8 bytes = one 128-byte run per kept channel
		// (sixteen lanes per channel; the two channels a half-wave reads are sixteen apart: their rows start 32 banks apart)
		// The slots come out of four registers (slpk), the tile in batches of kPfbWb reads answered together: with a slot
		// lookup and a tile read each waited for, one after the other, a tile's write-out was 32 trips to LDS in a row.
		static_assert(kPfbTile == 16, "the write-out's lane map is for sixteen instants per tile");
		{
			const int u = r & 15, hq = r >> 4;
			const long long t = t0 + tt + u;
			const bool tin = t < a.T;
			const float2 *row = &tile[(16 * hq) * (kPfbTile + 1) + (hq >> 1) + u];      // channel 16 hq + i is (kPfbTile + 1) i on
			// (the slots looked at afresh per tile: sixteen store addresses carried across the loop are 32 registers and a wave)
#pragma unroll
			for (int j = 0; j < 4; j++)
				asm volatile("" : "+v"(slpk[j]));
#pragma unroll
			for (int i0 = 0; i0 < 16; i0 += kPfbWb) {
				float2 v[kPfbWb];
#pragma unroll
				for (int k = 0; k < kPfbWb; k++)
					v[k] = row[(i0 + k) * (kPfbTile + 1)];
#pragma unroll
				for (int k = 0; k < kPfbWb; k++) {
					const int i = i0 + k;
					const unsigned sl = (slpk[i >> 2] >> (8 * (i & 3))) & 0xffu;
					// (streamed past the caches: nothing of it is read again before the launch is over -- with ordinary stores the
					// 640 MB evict the input still to be read and are written back under the next kernel; A/B profiles/r06am)
					if (sl != 0xffu && tin)
						__builtin_nontemporal_store((pf_v2f){v[k].x, v[k].y}, reinterpret_cast<pf_v2f *>(&a.y[(long long)sl * a.T + t]));
				}
			}
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
static constexpr int kRsRing = 8;            // k_resamp2: ring slots of one period's window each (seven periods in flight)
static constexpr int kRsWinTight = 128, kRsWinShort = 256, kRsWinLong = 512;     // LDS window (samples) per wave, >= span

// acc += e (s.re, s.im) for the real tap e that sits in the LOW (HI = false) or HIGH half of the register pair `ee`: one
// v_pk_fma_f32 with the half picked by the operand selects.  The compiler's own code for the two scalar chains
// orr = fma(e, s.re, orr), oi = fma(e, s.im, oi) is the same packed FMA, but on a pair (e, e) it builds for every tap -- 60
// registers of taps instead of 30 and five waves per SIMD instead of eight, with two LDS reads in flight per wave.
// (s_nop: a packed result needs one wait state before its next use.)
typedef float rs_v2f __attribute__((ext_vector_type(2)));
typedef float rs_v4f __attribute__((ext_vector_type(4)));
template <bool HI>
__device__ __forceinline__ void rs_mac(rs_v2f &acc, rs_v2f ee, rs_v2f s)
{
	if constexpr (HI)
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\ts_nop 0" : "+v"(acc) : "v"(ee), "v"(s));
	else
		asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\ts_nop 0" : "+v"(acc) : "v"(ee), "v"(s));
}

// the same for TWO accumulators (k_resamp2's two outputs, each with its own taps and its own view of the sample): the second
// multiply-add is the wait state of the first and the other way round, in whatever order the statements end up
template <bool HI>
__device__ __forceinline__ void rs_mac2(rs_v2f &acc0, rs_v2f &acc1, rs_v2f e0, rs_v2f e1, rs_v2f s0, rs_v2f s1)
{
	if constexpr (HI)
		asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
		    : "+v"(acc0), "+v"(acc1) : "v"(e0), "v"(e1), "v"(s0), "v"(s1));
	else
		asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
		    : "+v"(acc0), "+v"(acc1) : "v"(e0), "v"(e1), "v"(s0), "v"(s1));
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
// k_resamp2 -- the same resampler with TWO consecutive outputs per lane, for the plans that raise the rate.
//
// What bounds k_resamp is LDS delivery: 30 eight-byte reads per output, 240 bytes, 4 cycles per wave-wide read (0.42 ms of
// its 0.51 ms on the bench's capture; SQ_LDS_IDX_ACTIVE says the same).  Where the rate goes up, consecutive outputs use
// input windows that start at the same sample or one apart, so a lane that owns outputs p and p + 1 needs 31 samples for
// the two.  It fetches them as SIXTEEN ALIGNED PAIRS (ds_read_b128: the 32 samples from the odd index T >= its newest one
// down): half the read instructions per output, and conflict-free -- the eight lanes of a pass ask for five or six
// consecutive 16-byte pairs, where 8-byte reads at this lane stride (4/3 of a sample) collide two ways (tried first:
// SQ_LDS_BANK_CONFLICT 3 500 per wave, no gain).  The price is registers, two sets of taps, and that the windows'
// offsets against T differ from lane to lane: an output's taps are therefore stored SHIFTED by its offset sh = T - (its
// newest sample) in 0 .. 2 (E[i] = e[i - sh], zero outside; 30 + 2 = 32 entries = the 16 pairs), so that the multiply-adds
// run over the shared sample index with static register numbers.  Per output the sum still runs over its own taps
// k = 0 .. 29 in that order; at the four positions that can lie outside an output's taps (0, 1, 30, 31) the SAMPLE is
// replaced by zero where it does (a per-lane select), so that neither a non-finite sample outside an output's thirty taps
// nor the two unwritten slots at either end of the window reach it.  Results are bit for bit k_resamp's.
// ---------------------------------------------------------------------------
//
// ... and that alone gained nothing (0.55 ms against 0.51): with the LDS reads halved the counters show neither pipe busy
// (VALU 43 %, LDS 43 %) -- the kernel moves 1.6 GB at 3 TB/s because that is what the bytes it keeps IN FLIGHT allow (two
// periods' windows per wave in registers: ~30 KB per CU against ~2.5 us of memory latency; the one-output form sits at the
// same product with six waves).  So the windows come in by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = the
// 128-sample window of one period in ONE instruction, no registers) into a ring of kRing slots, kRing - 1 periods ahead.
//   * One counter covers loads and stores on this ISA and they complete out of order with respect to each other, but loads
//     complete in order among themselves: with exactly kRing - 1 younger window loads behind the one about to be used,
//     `s_waitcnt vmcnt(kRing - 1)` cannot be satisfied while that one is outstanding (it and all younger ones would make
//     kRing) -- stores still in flight only make the wait longer, never too short.  Hence EXACTLY one DMA per iteration:
//     windows that cross the stream's ends are fetched from clamped addresses and their outside samples zeroed in LDS
//     afterwards; periods past the end fetch the last window again.
//   * The compiler does not know what an LDS-DMA writes and answers any LDS read of its own with vmcnt(0), which would
//     drain the ring every period: the window is read with inline ds_read_b128 and counted lgkmcnt waits (LDS operations of
//     a wave complete in order), four pairs per batch, the next batch in flight while one is used.
template <int kRsWin, int kRing, int WG = 1>
__global__ __launch_bounds__(64 * WG) __attribute__((amdgpu_waves_per_eu(4)))
void k_resamp2(ResampArgs a, long long P, long long Q, int span)
{
	// (the body exists in the device pass only: the host pass drops the whole stub, silently, over the LDS-DMA builtin)
#if defined(__HIP_DEVICE_COMPILE__)
	constexpr int R = 2;
	constexpr int kTaps = kRsTapsShort;
	constexpr int L = kTaps + R;              // sample positions a lane walks (its pairs cover T - 31 .. T)
	constexpr int LP = L / 2;
	constexpr int kPad = 2;                   // window sample w sits in slot position w + 2: T - 31 >= -2
	constexpr int kSlot = (kRsWin + kPad) / 2;          // 16-byte pairs per ring slot
	static_assert(kRsWin == 128, "one global_load_lds_dwordx4 per window: 64 lanes x two samples");
	static_assert(LP % 4 == 0, "batches of four pairs");
	// (WG > 1: the waves of WG consecutive output groups in one work-group -- nothing shared, nothing synchronised, each its
	// own ring: they run on one CU at about one pace, so where their windows overlap -- a third of each, at 2/3 -- the second
	// to ask finds the line in the CU's or the XCD's cache.  As single-wave work-groups, even dealt to one XCD, they drift
	// periods apart and every window comes through the fabric: 1 426 MB fetched for a 640 MB input, 750 MB this way)
	__shared__ float4 ring_all[WG][kRing][kSlot];
	const int lane = threadIdx.x & 63;
	const int wv = WG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
	float4 (*ring4)[kSlot] = ring_all[wv];
	const int sl = blockIdx.y;
	const long long p0 = ((long long)blockIdx.x * WG + wv) * 64 * R;
	if (WG > 1 && p0 >= P)
		return;
	// phases of this lane's outputs: exact integer arithmetic, once (an output past the period's last takes the last one's
	// phase: it is not stored, and its window stays inside the wave's)
	long long ip[R];
	int jf[R];
	float frac[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		const long long p = p0 + (long long)R * lane + r;
		const long long Np = (long long)a.j0 * a.den + (p < P ? p : P - 1) * a.num;
		const long long fl = Np / a.den;
		frac[r] = (float)(Np - fl * a.den) / (float)a.den;
		jf[r] = (int)(fl % a.nfilt);
		ip[r] = fl / a.nfilt;
	}
	const long long N0 = (long long)a.j0 * a.den + p0 * a.num;
	const long long i_first = (N0 / a.den) / a.nfilt - (kTaps - 1);     // first input the wave needs (period 0)
	const int top = (int)(ip[R - 1] - i_first) + kPad;                  // slot position of the lane's newest sample
	const int T = top | 1;                                              // the odd position at or above it: pairs are (T - 1, T), ...
	int sh[R];
	rs_v2f E[R][LP];
#pragma unroll
	for (int r = 0; r < R; r++) {
		sh[r] = T - ((int)(ip[r] - i_first) + kPad);                    // 0 .. 2: the rate goes up (the launcher checks)
#pragma unroll
		for (int i = 0; i < L; i += 2) {
			float e[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				// (every lane loads, from a clamped index: the 64 loads are in flight together)
				const int k = i + h - sh[r];
				const bool in = k >= 0 && k < kTaps;
				const float2 b = a.bank[jf[r] * kTaps + (k < 0 ? 0 : (k >= kTaps ? kTaps - 1 : k))];
				e[h] = fmaf(frac[r], b.y, b.x) * (in ? 1.0f : 0.0f);      // (a product, not a select: the load stays unconditional)
			}
			E[r][i / 2] = (rs_v2f){e[0], e[1]};
		}
	}
	// (the taps have arrived before the first window is asked for: no ordinary load is pending inside the loop, where the
	// compiler would wait for it with vmcnt(0))
#pragma unroll
	for (int r = 0; r < R; r++)
#pragma unroll
		for (int q = 0; q < LP; q++)
			asm volatile("" : "+v"(E[r][q]));
	const float2 *__restrict__ y = a.y + (long long)sl * a.T;
	float2 *__restrict__ out = a.out + (long long)sl * a.out_stride;
	const long long m0 = (long long)blockIdx.z * kRsPeriods;
	const long long periods = (a.n_out + P - 1) / P;                    // periods that have an output at all

	// window of period m: inputs i_first + m Q + [0, 128), lane l brings samples 2 l and 2 l + 1
	auto issue = [&](long long m, int slot) {
		if (m >= periods)
			m = periods - 1;
		long long s0 = i_first + m * Q + 2 * lane;
		s0 = s0 < 0 ? 0 : (s0 > a.T - 2 ? a.T - 2 : s0);                // (a stream has at least two samples: the launcher checks)
		__builtin_amdgcn_global_load_lds(y + s0, &ring4[slot][kPad / 2], 16, 0, 0);
	};
	const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&ring4[0][0]);
	const unsigned rd0 = lds0 + (unsigned)(T >> 1) * 16u;               // pair q of slot s: rd0 + s * kSlot * 16 - q * 16

#pragma unroll
	for (int d = 0; d < kRing - 1; d++)
		issue(m0 + d, d);
	int slot = 0;
	for (int mm = 0; mm < kRsPeriods; mm++) {
		const long long m = m0 + mm;
		if (m * P >= a.n_out)
			break;
		issue(m + kRing - 1, slot == 0 ? kRing - 1 : slot - 1);
		asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRing - 1) : "memory");
		{
			// a window that crosses the stream's ends: its outside samples are zeros (k_resamp's fetch), whatever the clamped
			// addresses brought
			const long long w0 = i_first + m * Q;
			if (w0 < 0 || w0 + kRsWin > a.T) {
				float2 *xs = reinterpret_cast<float2 *>(&ring4[slot][kPad / 2]);
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const long long sidx = w0 + 2 * lane + h;
					float2 v = make_float2(0.f, 0.f);
					if (sidx >= 0 && sidx < a.T)
						v = y[sidx];
					xs[2 * lane + h] = v;
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			}
		}
		const unsigned rd = rd0 + (unsigned)slot * (unsigned)(kSlot * 16);
		rs_v2f acc[R];
#pragma unroll
		for (int r = 0; r < R; r++)
			acc[r] = (rs_v2f){0.f, 0.f};
		rs_v4f pr[2][4];
		const unsigned rdb = rd - 16u * (unsigned)(LP - 1);            // lowest pair of the lane; pair q at offset 16 (LP - 1 - q)
#if defined(GMR1_EXP_RS) && (GMR1_EXP_RS & 2)
		// (timing experiment: no window reads -- results are garbage)
#define GMR1_RS_READ(dst, q) asm volatile("" : "=v"(dst))
#else
#define GMR1_RS_READ(dst, q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(rdb), "n"(16 * (LP - 1 - (q))))
#endif
#pragma unroll
		for (int q = 0; q < 4; q++)
			GMR1_RS_READ(pr[0][q], q);
#pragma unroll
		for (int b = 0; b < LP / 4; b++) {
			if (b + 1 < LP / 4) {
#pragma unroll
				for (int q = 0; q < 4; q++)
					GMR1_RS_READ(pr[(b + 1) & 1][q], 4 * (b + 1) + q);
				asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pr[b & 1][0]), "+v"(pr[b & 1][1]), "+v"(pr[b & 1][2]), "+v"(pr[b & 1][3]));
			} else {
				asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr[b & 1][0]), "+v"(pr[b & 1][1]), "+v"(pr[b & 1][2]), "+v"(pr[b & 1][3]));
			}
#pragma unroll
			for (int qq = 0; qq < 4; qq++) {
				const int q = 4 * b + qq;
				const rs_v4f pq = pr[b & 1][qq];
				// position i = 2 q is slot position T - 2 q (the pair's upper sample), i = 2 q + 1 the lower
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const int i = 2 * q + h;
					rs_v2f sv[R];
#pragma unroll
					for (int r = 0; r < R; r++) {
						sv[r] = h ? (rs_v2f){pq.x, pq.y} : (rs_v2f){pq.z, pq.w};
						if (i < R || i >= kTaps) {
							// a position outside this output's taps in some lanes (or one of the slot's unwritten ends)
							const bool outside = i < sh[r] || i > sh[r] + kTaps - 1;
							sv[r] = outside ? (rs_v2f){0.f, 0.f} : sv[r];
						}
					}
#if defined(GMR1_EXP_RS) && (GMR1_EXP_RS & 1)
					// (timing experiment: no multiply-adds -- results are garbage)
					asm volatile("" :: "v"(sv[0]), "v"(sv[1]));
#else
					if (h)
						rs_mac2<true>(acc[0], acc[1], E[0][q], E[1][q], sv[0], sv[1]);
					else
						rs_mac2<false>(acc[0], acc[1], E[0][q], E[1][q], sv[0], sv[1]);
#endif
				}
			}
		}
#undef GMR1_RS_READ
		const long long pl = p0 + (long long)R * lane;
		const long long n = m * P + pl;
#if defined(GMR1_EXP_RS) && (GMR1_EXP_RS & 4)
		// (timing experiment: no output -- one store per wave keeps the sums alive)
		if (m == m0 + kRsPeriods - 1 && lane == 0)
			out[n] = make_float2(acc[0].x + acc[1].x, acc[0].y + acc[1].y);
		else if (false)
#endif
		if (pl + 1 < P && n + 1 < a.n_out && ((reinterpret_cast<uintptr_t>(out + n) & 15) == 0)) {
			// (a streaming store, as k_pfb64's: -8 % on this kernel, -5 % on the one before it in the step)
			__builtin_nontemporal_store((rs_v4f){acc[0].x, acc[0].y, acc[1].x, acc[1].y}, reinterpret_cast<rs_v4f *>(out + n));
		} else {
#pragma unroll
			for (int r = 0; r < R; r++)
				if (pl + r < P && n + r < a.n_out)
					out[n + r] = make_float2(acc[r].x, acc[r].y);
		}
		slot = slot + 1 == kRing ? 0 : slot + 1;
	}
	// nothing of this wave is left in flight towards its LDS when it ends
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---------------------------------------------------------------------------
// k_resamp2w -- k_resamp2 with ONE window ring per work-group of kRsWg waves.  AN EXPERIMENT THAT LOST: only the profiling
// build runs it (GMR1_HIP_RESAMP_WG=1), for the comparison; results are bit for bit k_resamp2's.
//
// Of the bytes k_resamp2 moves the reads are mostly waste -- every wave fetches a 128-sample window for 128 outputs that
// consume 85.5 new inputs at 2/3, and waves that share input drift periods apart, so no L2 holds a line for the next one:
// 1 426 MB fetched for a 640 MB input (tools/exp/traffic_chan.sh).  Here the kRsWg waves that own consecutive 128-output
// groups of a period share the union of their windows: it is fetched ONCE per period as `nch` consecutive 128-sample
// chunks (wave w brings chunk w: six chunks for eight waves at 2/3), every wave reads its own thirty-one-sample stretches
// out of the shared slot exactly as k_resamp2 does out of its own, with the same taps, order and selects.  Fetched:
// 783 MB.  Time: 0.47 ms with a barrier per period, 0.50 ms with the counters below, against 0.42 ms -- the loads, stores
// and synchronisation alone (arithmetic and window reads compiled out, -DGMR1_EXP_RS=3) take 0.38 ms in this form and
// 0.39 ms in k_resamp2's: neither form is bound by the bytes it reads, and what the waves of a group now wait for each
// other costs more than the reads saved (profiles/r06ah, r06aj, r06ak).  What did move both kernels were streaming stores.
//   * No barrier per period (tried first: the eight waves in step took 0.47 ms for 0.78 GB of reads, against 0.42 ms for
//     1.43 GB the other way -- every wave waiting for the slowest at every period, all of them in the same phase of their
//     work).  Two counters per slot instead, in LDS: `ready` (a wave adds one once its chunk of the slot's period has
//     landed: behind its own s_waitcnt vmcnt) and `done` (a wave adds one behind its last read of the slot).  A wave reads
//     period m's slot once ready says all nch chunks of that period are in, and sends its chunk of period m + kRsAhead
//     into the slot of period m + kRsAhead - kRing once done says every computing wave is through with that one: the
//     waves may drift kRing - kRsAhead - 1 periods apart.  Waits at period i depend on events of periods <= i only, and the
//     wave furthest behind never waits for one ahead of it: no cycle.  Polls are bounded; a bound that runs out traps.
//   * All LDS traffic inside the loop is inline assembly: the compiler answers LDS operations of its own behind an
//     LDS-DMA with vmcnt(0), which would drain the ring.
//   * Every wave issues exactly one DMA per period (the vmcnt arithmetic of k_resamp2): waves without a chunk fetch into
//     a dump slot.  Waves beyond the period's last output group still fetch and count as producers; they skip the rest.
// ---------------------------------------------------------------------------
static constexpr int kRsWg = 8;              // waves of a k_resamp2w work-group
static constexpr int kRsAhead = 4;           // periods its chunks are asked for ahead of their use

__device__ __forceinline__ unsigned rsw_peek(unsigned addr)
{
	unsigned v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void rsw_wait_ge(unsigned addr, unsigned want)
{
#pragma clang loop unroll(disable)
	for (int spin = 0; rsw_peek(addr) < want; spin++) {
		if (spin > (1 << 20))
			__builtin_trap();                 // (a counter that never arrives: cannot happen, must not hang)
		__builtin_amdgcn_s_sleep(1);
	}
}
__device__ __forceinline__ void rsw_add1(unsigned addr, int lane)
{
	if (lane == 0) {
		const unsigned one = 1u;
		asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
	}
}

template <int kRing>
__global__ __launch_bounds__(64 * kRsWg) __attribute__((amdgpu_waves_per_eu(4)))
void k_resamp2w(ResampArgs a, long long P, long long Q, int nch)
{
#if defined(__HIP_DEVICE_COMPILE__)
	constexpr int R = 2;
	constexpr int kTaps = kRsTapsShort;
	constexpr int L = kTaps + R;
	constexpr int LP = L / 2;
	constexpr int kPad = 2;                   // union sample k sits in slot position k + 2 (T - 31 >= -2); two more behind the last
	static_assert(LP % 4 == 0, "batches of four pairs");
	extern __shared__ float4 ringw[];         // kRing slots of (nch 128 + 4) samples, then the dump slot (128 samples)
	__shared__ unsigned cnt_ready[kRing], cnt_done[kRing];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (the wave's number: a scalar)
	const int sl = blockIdx.y;
	const long long p0u = (long long)blockIdx.x * kRsWg * 64 * R;        // first phase of the work-group
	const long long p0 = p0u + (long long)w * 64 * R;                     // ... of this wave
	const bool live_wave = p0 < P;
	const int slot_pairs = (nch * 128 + 2 * kPad) / 2;
	long long ip[R];
	int jf[R];
	float frac[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		const long long p = p0 + (long long)R * lane + r;
		const long long Np = (long long)a.j0 * a.den + (p < P ? p : P - 1) * a.num;
		const long long fl = Np / a.den;
		frac[r] = (float)(Np - fl * a.den) / (float)a.den;
		jf[r] = (int)(fl % a.nfilt);
		ip[r] = fl / a.nfilt;
	}
	const long long N0 = (long long)a.j0 * a.den + p0u * a.num;
	const long long i_first = (N0 / a.den) / a.nfilt - (kTaps - 1);     // first input the WORK-GROUP needs (period 0)
	const int top = (int)(ip[R - 1] - i_first) + kPad;
	const int T = top | 1;
	int sh[R];
	rs_v2f E[R][LP];
#pragma unroll
	for (int r = 0; r < R; r++) {
		sh[r] = T - ((int)(ip[r] - i_first) + kPad);
#pragma unroll
		for (int i = 0; i < L; i += 2) {
			float e[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int k = i + h - sh[r];
				const bool in = k >= 0 && k < kTaps;
				const float2 b = a.bank[jf[r] * kTaps + (k < 0 ? 0 : (k >= kTaps ? kTaps - 1 : k))];
				e[h] = fmaf(frac[r], b.y, b.x) * (in ? 1.0f : 0.0f);
			}
			E[r][i / 2] = (rs_v2f){e[0], e[1]};
		}
	}
#pragma unroll
	for (int r = 0; r < R; r++)
#pragma unroll
		for (int q = 0; q < LP; q++)
			asm volatile("" : "+v"(E[r][q]));
	const float2 *__restrict__ y = a.y + (long long)sl * a.T;
	float2 *__restrict__ out = a.out + (long long)sl * a.out_stride;
	const long long m0 = (long long)blockIdx.z * kRsPeriods;
	const long long periods = (a.n_out + P - 1) / P;
	const bool has_chunk = w < nch;
	// computing waves of this work-group (those with an output group inside the period)
	const long long groups_left = (P - p0u + 64 * R - 1) / (64 * R);
	const unsigned n_live = (unsigned)(groups_left < kRsWg ? (groups_left > 0 ? groups_left : 0) : kRsWg);
	if (threadIdx.x < kRing) {
		cnt_ready[threadIdx.x] = 0u;
		cnt_done[threadIdx.x] = 0u;
	}
	__syncthreads();
	const unsigned a_ready = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&cnt_ready[0]));
	const unsigned a_done = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&cnt_done[0]));
	// this wave's chunk of period m: inputs i_first + m Q + 128 w + [0, 128), lane l brings samples 2 l and 2 l + 1
	auto issue = [&](long long m, int slot) {
		if (m >= periods)
			m = periods - 1;
		long long s0 = i_first + m * Q + (has_chunk ? 128 * w : 0) + 2 * lane;
		s0 = s0 < 0 ? 0 : (s0 > a.T - 2 ? a.T - 2 : s0);
		float4 *dst = has_chunk ? &ringw[slot * slot_pairs + kPad / 2 + 64 * w] : &ringw[kRing * slot_pairs];
		__builtin_amdgcn_global_load_lds(y + s0, dst, 16, 0, 0);
	};
	const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void *)&ringw[0]);
	const unsigned rd0 = lds0 + (unsigned)(T >> 1) * 16u;

	static_assert(kRsAhead >= 1 && kRsAhead < kRing, "a slot is refilled kRing - kRsAhead periods after its use");
#pragma unroll
	for (int d = 0; d < kRsAhead; d++)
		issue(m0 + d, d);
	int slot = 0;
	for (int mm = 0; mm < kRsPeriods; mm++) {
		const long long m = m0 + mm;
		if (m * P >= a.n_out)
			break;
		{
			// the chunk of period m + kRsAhead, into the slot whose earlier periods every computing wave is through with
			const int sa = (mm + kRsAhead) % kRing;
			if (has_chunk && mm + kRsAhead >= kRing)
				rsw_wait_ge(a_done + 4u * (unsigned)sa, n_live * (unsigned)((mm + kRsAhead) / kRing));
			issue(m + kRsAhead, sa);
		}
		asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRsAhead) : "memory");
		if (has_chunk) {
			// a chunk that crosses the stream's ends: its outside samples are zeros, whatever the clamped addresses brought
			const long long c0 = i_first + m * Q + 128 * w;
			if (c0 < 0 || c0 + 128 > a.T) {
				float2 *xs = reinterpret_cast<float2 *>(&ringw[slot * slot_pairs + kPad / 2 + 64 * w]);
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const long long sidx = c0 + 2 * lane + h;
					float2 v = make_float2(0.f, 0.f);
					if (sidx >= 0 && sidx < a.T)
						v = y[sidx];
					xs[2 * lane + h] = v;
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			}
		}
		if (has_chunk)
			rsw_add1(a_ready + 4u * (unsigned)slot, lane);
		if (live_wave) {
			// every chunk of this period is in its slot
			rsw_wait_ge(a_ready + 4u * (unsigned)slot, (unsigned)nch * (unsigned)(mm / kRing + 1));
			const unsigned rd = rd0 + (unsigned)slot * (unsigned)(slot_pairs * 16);
			rs_v2f acc[R];
#pragma unroll
			for (int r = 0; r < R; r++)
				acc[r] = (rs_v2f){0.f, 0.f};
			// (pairs in batches of kB, the next batch in flight while one is used: two, not k_resamp2's four -- the counters'
			// operands and the wider slot arithmetic need the registers, and a spilled tap comes back behind vmcnt(0))
			constexpr int kB = 2;
			static_assert(LP % kB == 0, "whole batches");
			rs_v4f pr[2][kB];
			const unsigned rdb = rd - 16u * (unsigned)(LP - 1);
#if defined(GMR1_EXP_RS) && (GMR1_EXP_RS & 2)
#define GMR1_RS_READ(dst, q) asm volatile("" : "=v"(dst))
#else
#define GMR1_RS_READ(dst, q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(rdb), "n"(16 * (LP - 1 - (q))))
#endif
#pragma unroll
			for (int q = 0; q < kB; q++)
				GMR1_RS_READ(pr[0][q], q);
#pragma unroll
			for (int b = 0; b < LP / kB; b++) {
				if (b + 1 < LP / kB) {
#pragma unroll
					for (int q = 0; q < kB; q++)
						GMR1_RS_READ(pr[(b + 1) & 1][q], kB * (b + 1) + q);
					asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(pr[b & 1][0]), "+v"(pr[b & 1][1]) : "n"(kB));
				} else {
					asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr[b & 1][0]), "+v"(pr[b & 1][1]));
					rsw_add1(a_done + 4u * (unsigned)slot, lane);       // (the slot's last read of this wave has returned)
				}
#pragma unroll
				for (int qq = 0; qq < kB; qq++) {
					const int q = kB * b + qq;
					const rs_v4f pq = pr[b & 1][qq];
#pragma unroll
					for (int h = 0; h < 2; h++) {
						const int i = 2 * q + h;
						rs_v2f sv[R];
#pragma unroll
						for (int r = 0; r < R; r++) {
							sv[r] = h ? (rs_v2f){pq.x, pq.y} : (rs_v2f){pq.z, pq.w};
							if (i < R || i >= kTaps) {
								const bool outside = i < sh[r] || i > sh[r] + kTaps - 1;
								sv[r] = outside ? (rs_v2f){0.f, 0.f} : sv[r];
							}
						}
#if defined(GMR1_EXP_RS) && (GMR1_EXP_RS & 1)
						asm volatile("" :: "v"(sv[0]), "v"(sv[1]));
#else
						if (h)
							rs_mac2<true>(acc[0], acc[1], E[0][q], E[1][q], sv[0], sv[1]);
						else
							rs_mac2<false>(acc[0], acc[1], E[0][q], E[1][q], sv[0], sv[1]);
#endif
					}
				}
			}
#undef GMR1_RS_READ
			const long long pl = p0 + (long long)R * lane;
			const long long n = m * P + pl;
			if (pl + 1 < P && n + 1 < a.n_out && ((reinterpret_cast<uintptr_t>(out + n) & 15) == 0)) {
				*reinterpret_cast<float4 *>(out + n) = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
			} else {
#pragma unroll
				for (int r = 0; r < R; r++)
					if (pl + r < P && n + r < a.n_out)
						out[n + r] = make_float2(acc[r].x, acc[r].y);
			}
		}
		slot = slot + 1 == kRing ? 0 : slot + 1;
	}
	// nothing of this wave is left in flight towards the work-group's LDS when it ends
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
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
	// two consecutive outputs per lane where the rate goes up (their windows then start at most one sample apart) and the
	// 128 phases of a wave still fit the tight window (profiling build: GMR1_HIP_RESAMP_R=1 keeps one output per lane)
	static const bool one_only = []{ const char *e = profile_env("GMR1_HIP_RESAMP_R"); return e && atoi(e) == 1; }();
	const int span_r = (int)((127 * a.num) / (a.den * a.nfilt)) + taps + 2;
	const int multi_r = (!lng && !ext && !one_only && a.num < a.den * a.nfilt && span_r <= kRsWinTight && P >= 2 && a.T >= 2) ? 2 : 1;
	if (lng && ext)
		return hipErrorInvalidValue;             // (the long bank is the direct mode's: neither option reaches it)
	if (lng)
		hipLaunchKernelGGL((k_resamp<kRsTapsLong, kRsWinLong>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (ext && span <= kRsWinTight)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinTight, true>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (ext)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinShort, true>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else if (multi_r > 1) {
		// one window ring per work-group of kRsWg waves where the union of their windows is fewer chunks than they are and
		// the period has at least that many output groups (k_resamp2w); else a ring per wave
		const long long gx2 = (P + 127) / 128;
		const int span_w = (int)(((long long)(kRsWg * 128 - 1) * a.num) / (a.den * a.nfilt)) + taps + 2;
		const int nch = (span_w + 127) / 128;
		// (measured slower than a ring per wave -- 0.47-0.50 against 0.42 ms with 0.78 against 1.43 GB fetched: DESIGN 4.7 --, so
		// only the profiling build runs it, on request)
		static const bool shared_ring = profile_env("GMR1_HIP_RESAMP_WG") != nullptr;
		if (shared_ring && nch < kRsWg && gx2 >= kRsWg) {
			const size_t lds_w = ((size_t)kRsRing * (nch * 128 + 4) + 128) * sizeof(float2);
			static std::mutex mu;
			static bool told[64] = {};
			int dev = 0;
			if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
				return hipErrorInvalidDevice;
			{
				std::lock_guard<std::mutex> lk(mu);
				if (!told[dev]) {
					const hipError_t e = hipFuncSetAttribute((const void *)k_resamp2w<kRsRing>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)kRsRing * (kRsWg * 128 + 4) + 128) * sizeof(float2)));
					if (e != hipSuccess)
						return e;
					told[dev] = true;
				}
			}
			hipLaunchKernelGGL((k_resamp2w<kRsRing>), dim3((unsigned)((gx2 + kRsWg - 1) / kRsWg), (unsigned)a.n_slots, gz), dim3(64 * kRsWg), lds_w, stream, a, P, Q, nch);
		} else {
			// eight consecutive output groups per work-group (each wave its own ring, nothing synchronised): the same time as a
			// wave per work-group, 750 MB fetched instead of 1 426 MB (profiles/r06aq; the profiling build keeps the other form:
			// GMR1_HIP_RESAMP_WG1=1)
			// (the largest of 8, 4, 2 groups that divides the period's groups: a work-group's LDS is its waves' rings whether or
			// not they have an output group, so a part-filled one would cost residency)
			static const bool wg1 = profile_env("GMR1_HIP_RESAMP_WG1") != nullptr;
			const int wgw = wg1 ? 1 : (gx2 % 8 == 0 ? 8 : (gx2 % 4 == 0 ? 4 : (gx2 % 2 == 0 ? 2 : 1)));
			const dim3 grid2((unsigned)(gx2 / wgw), (unsigned)a.n_slots, gz);
			if (wgw == 8)
				hipLaunchKernelGGL((k_resamp2<kRsWinTight, kRsRing, 8>), grid2, dim3(64 * 8), 0, stream, a, P, Q, span_r);
			else if (wgw == 4)
				hipLaunchKernelGGL((k_resamp2<kRsWinTight, kRsRing, 4>), grid2, dim3(64 * 4), 0, stream, a, P, Q, span_r);
			else if (wgw == 2)
				hipLaunchKernelGGL((k_resamp2<kRsWinTight, kRsRing, 2>), grid2, dim3(64 * 2), 0, stream, a, P, Q, span_r);
			else
				hipLaunchKernelGGL((k_resamp2<kRsWinTight, kRsRing>), grid2, dim3(64), 0, stream, a, P, Q, span_r);
		}
	}

	else if (span <= kRsWinTight)
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinTight>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	else
		hipLaunchKernelGGL((k_resamp<kRsTapsShort, kRsWinShort>), dim3(gx, (unsigned)a.n_slots, gz), dim3(64), 0, stream, a, P, Q, span);
	return hipGetLastError();
}

}  // namespace gmr1
