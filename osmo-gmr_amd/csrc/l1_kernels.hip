// l1_kernels.hip -- layer-1 channel decoders of the traffic channel bursts (gfx950).
//
//   k_facch3 : FACCH3, 4 bursts x 104 soft bits -> 10 bytes + 32 status bits
//              (reference src/l1/facch3.c:121-170): status demux, [decipher], descramble,
//              intra-burst de-interleave (N=12), 4-way burst demux, K=5 rate-1/4 Viterbi
//              (92 bits + flush), CRC16, LSB-first packing.  Four frames per wavefront, one per
//              16-lane DPP row, the same packed [metric | window decisions] words and in-place
//              butterfly as the BCCH decoder; the cost of a coded 4-bit word is the sum of two
//              byte-table entries (coded bits 0-1, 2-3).
//   k_tch3   : TCH3 speech, 212 soft bits -> 2 x 10 bytes + 4 status bits
//              (reference src/l1/tch3.c:124-183): status demux, [decipher], descramble,
//              frame demux (m), 104-permutation, K=7 tail-biting rate-1/2 Viterbi with
//              P(1;2) puncturing (two passes of 48 steps; one burst per wave, a frame per half-wave,
//              two states per lane), 32 hard
//              class-2 bits, MSB-first packing.  One (burst, frame) per wavefront.
//
// Both follow libosmocore's generic osmo_conv_decode (oracle/orc_3p.c decisions D1, D4) or, instantiated with ACC, its
// accelerated decoder osmo_conv_decode_acc (decision D1b, oracle/orc_3p_acc.c): correlation metric -- here as the
// equivalent non-negative cost  sum of |in| over the coded bits that contradict the soft bit's sign  --, every start
// state allowed (flushed codes: state 0 leads by 127 * N * K; tail-biting: all equal), full butterflies in the flush
// steps, tail-biting end state = best sum, first in the decoder's own (bit-reversed) state numbering, no metric returned.
#include <type_traits>

#include "gmr1_dev.h"
#include "tch3_body.h"

namespace gmr1 {
using namespace t3;

__constant__ ScrBits c_scr = make_scr();

// CRC16 syndromes of a 76-bit message followed by its 16 CRC bits (see rx_kernels.hip)
struct Syn92 { uint16_t s[92]; };
static constexpr Syn92 make_syn92()
{
	Syn92 t{};
	for (int k = 0; k < 76; k++) {
		uint32_t crc = 0x8000u;
		for (int i = k; i < 76; i++)
			crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xffffu) : ((crc << 1) & 0xffffu);
		t.s[k] = (uint16_t)crc;
	}
	for (int i = 0; i < 16; i++)
		t.s[76 + i] = (uint16_t)(1u << (15 - i));
	return t;
}
__constant__ Syn92 c_syn92 = make_syn92();

template <int X>
__device__ __forceinline__ uint32_t row_xor(uint32_t v)
{
	if constexpr (X == 8) return dpp<0x128>(v);
	else if constexpr (X == 4) return dpp<0x1B>(dpp<0x141>(v));
	else if constexpr (X == 2) return dpp<0x4E>(v);
	else return dpp<0xB1>(v);
}
template <bool ACC = false>
__device__ __forceinline__ int sbit_cost(int v, int bit)
{
	if constexpr (ACC) {
		// conv_acc.c maximises sum in * (+-1): the cost of contradicting the soft bit is |in| (half the correlation lost)
		return bit ? (v > 0 ? v : 0) : (v < 0 ? -v : 0);
	} else {
		// ((in - (+-127))^2 >> 9), erasures cost nothing (libosmocore conv.c, generic decoder)
		const int e = bit ? v + 127 : v - 127;
		return v ? (__mul24(e, e) >> 9) : 0;
	}
}

// ---------------------------------------------------------------------------
// FACCH3
// ---------------------------------------------------------------------------
static constexpr int kF3Steps = 96;       // 92 bits + 4 flush

// K=5 rate-1/4: g0 = 1+D^3+D^4, g1 = 1+D+D^2+D^4, g2 = 1+D^2+D^4, g3 = 1+D+D^2+D^3+D^4 (conv.c:174-198).
// Per row location constants of the in-place 16-state butterfly with the masks 8, 7, 2, 1 (one DPP
// control each; see decode4_k5_12 in rx_kernels.hip for the construction):
//   ov[loc] : 4-bit coded word of the own transition per phase (bits 0-15) and of the partner's (16-31)
//   hi[loc] : bit j set when the lane holds the HIGH predecessor in phase j & 3 (16-step pattern)
struct K5r4Tab { uint32_t ov[16]; uint32_t hi[16]; };
static constexpr uint32_t k5r4_out(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	const uint32_t g[4] = {0x19u, 0x17u, 0x15u, 0x1fu};
	uint32_t o = 0;
	for (int i = 0; i < 4; i++) {
		uint32_t p = reg & g[i];
		p ^= p >> 4; p ^= p >> 2; p ^= p >> 1;
		o = (o << 1) | (p & 1u);
	}
	return o;
}
static constexpr K5r4Tab make_k5r4()
{
	K5r4Tab t{};
	for (uint32_t loc = 0; loc < 16; loc++) {
		uint32_t c[4] = {0, 0, 0, 0};
		c[0] = (loc >> 3) & 1u;
		uint32_t x = loc & 7u;
		c[1] = (x >> 2) & 1u;
		x ^= c[1] ? 7u : 0u;
		c[2] = (x >> 1) & 1u;
		c[3] = x & 1u;
		for (int ph = 0; ph < 4; ph++) {
			uint32_t sp = 0;
			for (int i = 0; i < 4; i++)
				sp |= c[(3 - i + ph) & 3] << i;
			const uint32_t b = sp >> 3;
			t.ov[loc] |= k5r4_out(sp, b) << (4 * ph);
			t.ov[loc] |= k5r4_out(sp ^ 8u, b) << (16 + 4 * ph);
			for (int j = ph; j < 16; j += 4)
				t.hi[loc] |= b << j;
		}
	}
	return t;
}
__constant__ K5r4Tab c_k5r4 = make_k5r4();

template <int PH>
__device__ __forceinline__ uint32_t k5_partner(uint32_t w)
{
	if constexpr (PH == 0) return dpp<0x128>(w);            // row_ror:8
	else if constexpr (PH == 1) return dpp<0x141>(w);       // row_half_mirror: xor 7
	else if constexpr (PH == 2) return dpp<0x4E>(w);        // quad_perm [2,3,0,1]
	else return dpp<0xB1>(w);                               // quad_perm [1,0,3,2]
}

// one rate-1/4 trellis step on the packed word [metric:16 | window decisions:16]; the cost of a coded
// 4-bit word is A[word >> 2] + B[word & 3] (two byte tables per step: coded bits 0-1 and 2-3)
template <int PH, typename CT>
__device__ __forceinline__ uint32_t k5r4_step(uint32_t w, const CT *__restrict__ ab, uint32_t ov_own, uint32_t ov_par)
{
	const uint32_t p = k5_partner<PH>(w);
	const uint32_t c_own = (uint32_t)ab[ov_own >> 2] + (uint32_t)ab[4 + (ov_own & 3u)];
	const uint32_t c_par = (uint32_t)ab[ov_par >> 2] + (uint32_t)ab[4 + (ov_par & 3u)];
	const uint32_t t1 = (c_own << 16) + w;
	const uint32_t t2 = (c_par << 16) + p;
	return t1 < t2 ? t1 : t2;
}

// ACC: two soft bits of -128 in one table entry cost 256 -- 16-bit table entries in that mode
template <bool ACC>
__global__ __launch_bounds__(64) void k_facch3(Facch3Args a)
{
	typedef typename std::conditional<ACC, uint16_t, uint8_t>::type CT;
	typedef typename std::conditional<ACC, uint4, uint2>::type CW;
	__shared__ __align__(16) int8_t s_eb[4][416];
	__shared__ __align__(16) CW s_cst[4][kF3Steps];         // per step: A[4] (coded bits 0-1), B[4] (coded bits 2-3)
	__shared__ uint16_t s_win[6][64];                       // window decisions per row location
	__shared__ __align__(16) uint32_t s_ub[4][4];
	const int lane = threadIdx.x;
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	const int f0 = blockIdx.x * 4;

	// ---- soft bits of the 4 frames (4 x 104 each) HBM -> LDS
	for (int q = 0; q < 4; q++) {
		const int f = f0 + q;
		uint32_t *dst = reinterpret_cast<uint32_t *>(&s_eb[q][0]);
		if (f < a.n) {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)f * 416);
			for (int i = lane; i < 104; i += 64)
				dst[i] = src[i];
		} else {
			for (int i = lane; i < 104; i += 64)
				dst[i] = 0;
		}
	}
	WSYNC();

	// ---- status bits: sign of e[22..29] of each burst (facch3.c:141-142)
	if (a.bits_s) {
		for (int q = 0; q < 4; q++) {
			const int f = f0 + q;
			if (f < a.n && lane < 32)
				a.bits_s[(size_t)f * 32 + lane] = s_eb[q][104 * (lane >> 3) + 22 + (lane & 7)] < 0;
		}
	}

	// ---- per trellis step: costs of the 4 coded bits
	// c[i] = cp[(i&3)*96 + (i>>2)], cp[kc] = ep[12*((5 kc)&7) + (kc>>3)] per burst, ep = descrambled xmy,
	// xmy = e[0..21] | e[30..103]   (facch3.c:144-158, interleave.c:73-87)
	for (int it = lane; it < 4 * kF3Steps; it += 64) {
		const int q = it / kF3Steps, k = it % kF3Steps;
		uint32_t c0[4], c1[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int i = 4 * k + j;              // index into bits_c
			const int burst = i & 3, kc = i >> 2;
			const int p = 12 * ((5 * kc) & 7) + (kc >> 3);       // position in xmy / ep (0..95)
			const int e = p < 22 ? p : p + 8;
			int v = s_eb[q][104 * burst + e];
			bool flip = (c_scr.w[p >> 5] >> (p & 31)) & 1u;
			if (a.ciph && (f0 + q) < a.n)
				flip ^= a.ciph[(size_t)(f0 + q) * 384 + 96 * burst + p] != 0;
			if (flip)
				v = (int8_t)(-v);
			c0[j] = (uint32_t)sbit_cost<ACC>(v, 0);
			c1[j] = (uint32_t)sbit_cost<ACC>(v, 1);
		}
		// entry x of A = cost of coded bits (0, 1) = (x >> 1, x & 1); B likewise for coded bits (2, 3)
		if constexpr (ACC) {
			s_cst[q][k] = make_uint4((c0[0] + c0[1]) | ((c0[0] + c1[1]) << 16), (c1[0] + c0[1]) | ((c1[0] + c1[1]) << 16),
			                         (c0[2] + c0[3]) | ((c0[2] + c1[3]) << 16), (c1[2] + c0[3]) | ((c1[2] + c1[3]) << 16));
		} else {
			const uint32_t aw = (c0[0] + c0[1]) | ((c0[0] + c1[1]) << 8) | ((c1[0] + c0[1]) << 16) | ((c1[0] + c1[1]) << 24);
			const uint32_t bw = (c0[2] + c0[3]) | ((c0[2] + c1[3]) << 8) | ((c1[2] + c0[3]) << 16) | ((c1[2] + c1[3]) << 24);
			s_cst[q][k] = make_uint2(aw, bw);
		}
	}
	WSYNC();

	// ---- forward pass on packed words (see decode4_k5_12 in rx_kernels.hip): 4 steps whose decisions
	// are u[-4..-1], five windows of 16 steps (u[16m .. 16m+15]) and a last one of 12 (u[80..91], the
	// final four being the flush: only b = 0 transitions survive)
	const uint32_t ovt = c_k5r4.ov[loc], hit = c_k5r4.hi[loc];
	uint32_t ov_own[4], ov_par[4];
	bool hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		ov_own[ph] = (ovt >> (4 * ph)) & 15u;
		ov_par[ph] = (ovt >> (16 + 4 * ph)) & 15u;
		hi[ph] = ((hit >> ph) & 1u) != 0;
	}
	uint32_t T[16];
#pragma unroll
	for (int j = 0; j < 16; j++)
		T[j] = hit & (1u << j);
	// every other start state: unreachable (generic decoder) / behind state 0 by 127 * N * K, halved like the costs (ACC)
	constexpr uint32_t kSent = ACC ? (127u * 4u * 5u / 2u) << 16 : 0xF0000000u;
	const CT *cb = reinterpret_cast<const CT *>(&s_cst[row][0]);
	uint32_t w = (loc ? kSent : 0u) | T[0];
	w = k5r4_step<0>(w, cb + 8 * 0, ov_own[0], ov_par[0]) + T[1];
	w = k5r4_step<1>(w, cb + 8 * 1, ov_own[1], ov_par[1]) + T[2];
	w = k5r4_step<2>(w, cb + 8 * 2, ov_own[2], ov_par[2]) + T[3];
	w = k5r4_step<3>(w, cb + 8 * 3, ov_own[3], ov_par[3]);
	w = (w & 0xffff0000u) | T[0];
#pragma unroll 1
	for (int wm = 0; wm < 5; wm++) {
		const CT *c = cb + 8 * (4 + 16 * wm);
#pragma unroll
		for (int j = 0; j < 16; j += 4) {
			w = k5r4_step<0>(w, c + 8 * (j + 0), ov_own[0], ov_par[0]) + T[(j + 1) & 15];
			w = k5r4_step<1>(w, c + 8 * (j + 1), ov_own[1], ov_par[1]) + T[(j + 2) & 15];
			w = k5r4_step<2>(w, c + 8 * (j + 2), ov_own[2], ov_par[2]) + T[(j + 3) & 15];
			w = k5r4_step<3>(w, c + 8 * (j + 3), ov_own[3], ov_par[3]) + (j + 4 < 16 ? T[(j + 4) & 15] : 0u);
		}
		s_win[wm][lane] = (uint16_t)w;
		w = (w & 0xffff0000u) | T[0];
	}
	{
		const CT *c = cb + 8 * 84;
#pragma unroll
		for (int j = 0; j < 8; j += 4) {
			w = k5r4_step<0>(w, c + 8 * (j + 0), ov_own[0], ov_par[0]) + T[j + 1];
			w = k5r4_step<1>(w, c + 8 * (j + 1), ov_own[1], ov_par[1]) + T[j + 2];
			w = k5r4_step<2>(w, c + 8 * (j + 2), ov_own[2], ov_par[2]) + T[j + 3];
			w = k5r4_step<3>(w, c + 8 * (j + 3), ov_own[3], ov_par[3]) + T[j + 4];
		}
		w = k5r4_step<0>(w, c + 8 * 8, ov_own[0], ov_par[0]);
		w = !ACC && hi[0] ? kSent : (w + T[9]);
		w = k5r4_step<1>(w, c + 8 * 9, ov_own[1], ov_par[1]);
		w = !ACC && hi[1] ? kSent : (w + T[10]);
		w = k5r4_step<2>(w, c + 8 * 10, ov_own[2], ov_par[2]);
		w = !ACC && hi[2] ? kSent : (w + T[11]);
		w = k5r4_step<3>(w, c + 8 * 11, ov_own[3], ov_par[3]);
		w = !ACC && hi[3] ? kSent : w;
		s_win[5][lane] = (uint16_t)w;
	}
	const uint32_t final_ae = ACC ? 0u : w >> 16;        // state 0 ends in location 0 of the row; osmo_conv_decode_acc returns 0
	WSYNC();

	// ---- survivor chain, one lane per row: six dependent 16-bit reads
	if (loc == 0) {
		// location of the state whose reversed nibble is x (basis 8, 7, 2, 1)
		constexpr unsigned long long kLocOf =
			0x0ull | (0x8ull << 4) | (0x7ull << 8) | (0xFull << 12) | (0x2ull << 16) | (0xAull << 20) |
			(0x5ull << 24) | (0xDull << 28) | (0x1ull << 32) | (0x9ull << 36) | (0x6ull << 40) |
			(0xEull << 44) | (0x3ull << 48) | (0xBull << 52) | (0x4ull << 56) | (0xCull << 60);
		const uint16_t *d16 = &s_win[0][row * 16];
		uint32_t L = 0, prev = 0;
#pragma unroll
		for (int wm = 5; wm >= 0; wm--) {
			const uint32_t h = d16[wm * 64 + L];
			L = (uint32_t)(kLocOf >> (4 * (h & 15u))) & 15u;
			if (wm & 1)
				prev = wm == 5 ? (h & 0xfffu) : h;
			else
				s_ub[row][wm >> 1] = h | (prev << 16);
		}
	}
	WSYNC();

	// ---- CRC16 over 76 + 16 bits: 6 bits per lane (16 x 6 = 96 >= 92)
	uint32_t syn = 0;
#pragma unroll
	for (int q = 0; q < 6; q++) {
		const int k = (int)loc * 6 + q;
		if (k < 92) {
			const uint32_t bit = (s_ub[row][k >> 5] >> (k & 31)) & 1u;
			syn ^= bit ? (uint32_t)c_syn92.s[k] : 0u;
		}
	}
	syn ^= row_xor<1>(syn);
	syn ^= row_xor<2>(syn);
	syn ^= row_xor<4>(syn);
	syn ^= row_xor<8>(syn);

	// ---- outputs: 76 bits LSB first -> 10 bytes, upper nibble of l2[9] = 0 (facch3.c:166-167)
	const int f = f0 + row;
	if (f < a.n) {
		if (loc < 5) {
			const uint32_t w = s_ub[row][loc >> 1] >> (16 * (loc & 1));
			uint32_t h = w & 0xffffu;
			if (loc == 4)
				h &= 0x0fffu;
			*reinterpret_cast<uint16_t *>(a.l2 + (size_t)f * 10 + 2 * loc) = (uint16_t)h;
		}
		if (loc == 0) {
			a.crc[f] = syn ? 1 : 0;
			a.conv[f] = (int32_t)final_ae;
		}
	}
}

// ---------------------------------------------------------------------------
// TCH3 speech (the decoder itself: tch3_body.h)
// ---------------------------------------------------------------------------
template <bool ACC>
__global__ __launch_bounds__(64) void k_tch3(Tch3Args a)
{
	__shared__ __align__(16) int8_t s_e[216];
	__shared__ __align__(16) Tch3Lds s_t3;
	const int lane = threadIdx.x;
	const int g = blockIdx.x;
	tch3_fill_locof(&s_t3, lane);
	{
		const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)g * 212);
		uint32_t *dst = reinterpret_cast<uint32_t *>(s_e);
		if (lane < 53)
			dst[lane] = src[lane];
		if (lane == 53)
			dst[53] = 0;                             // byte 212: what a punctured position reads
	}
	Tch3Lane lc;
	tch3_lane(lc, &s_t3, lane);
	WSYNC();
	tch3_burst<ACC>(a, g, lane, s_e, &s_t3, lc);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_facch3(const Facch3Args &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.conv_acc)
		hipLaunchKernelGGL(k_facch3<true>, dim3((a.n + 3) / 4), dim3(64), 0, st, a);
	else
		hipLaunchKernelGGL(k_facch3<false>, dim3((a.n + 3) / 4), dim3(64), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_tch3(const Tch3Args &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.conv_acc)
		hipLaunchKernelGGL(k_tch3<true>, dim3(a.n), dim3(64), 0, st, a);
	else
		hipLaunchKernelGGL(k_tch3<false>, dim3(a.n), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace gmr1
