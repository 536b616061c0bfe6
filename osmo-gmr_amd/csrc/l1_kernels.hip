// l1_kernels.hip -- layer-1 channel decoders of the traffic channel bursts (gfx950).
//
//   k_facch3 : FACCH3, 4 bursts x 104 soft bits -> 10 bytes + 32 status bits
//              (reference src/l1/facch3.c:121-170): status demux, [decipher], descramble,
//              intra-burst de-interleave (N=12), 4-way burst demux, K=5 rate-1/4 Viterbi
//              (92 bits + flush), CRC16, LSB-first packing.  Four frames per wavefront, one per
//              16-lane DPP row, the same in-place butterfly as the BCCH decoder; branch
//              metrics are two v_dot4_u32_u8 per candidate.
//   k_tch3   : TCH3 speech, 212 soft bits -> 2 x 10 bytes + 4 status bits
//              (reference src/l1/tch3.c:124-183): status demux, [decipher], descramble,
//              frame demux (m), 104-permutation, K=7 tail-biting rate-1/2 Viterbi with
//              P(1;2) puncturing (two passes of 48 steps, 64 states = 64 lanes), 32 hard
//              class-2 bits, MSB-first packing.  One (burst, frame) per wavefront.
//
// Both follow libosmocore's generic osmo_conv_decode (oracle/orc_3p.c decisions D1, D4).
#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr uint32_t kMaxAe = 0x00ffffffu;

// scrambler bits (reference src/l1/scramb.c:39-52), bit i of the sequence in word i>>5
struct ScrBits { uint32_t w[8]; };
static constexpr ScrBits make_scr()
{
	ScrBits t{};
	uint16_t r = 0x4d4b;
	for (int i = 0; i < 256; i++) {
		uint32_t b = ((r >> 14) ^ r) & 1u;
		r = (uint16_t)((r << 1) | b);
		t.w[i >> 5] |= b << (i & 31);
	}
	return t;
}
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

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int X>
__device__ __forceinline__ uint32_t row_xor(uint32_t v)
{
	if constexpr (X == 8) return dpp<0x128>(v);
	else if constexpr (X == 4) return dpp<0x1B>(dpp<0x141>(v));
	else if constexpr (X == 2) return dpp<0x4E>(v);
	else return dpp<0xB1>(v);
}
template <int X>
__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
	if constexpr (X >= 16) return (uint32_t)__shfl_xor((int)v, X);
	else return row_xor<X>(v);
}

__device__ __forceinline__ uint32_t rotl_n(uint32_t x, int r, int bits)
{
	const uint32_t m = (1u << bits) - 1u;
	return ((x << r) | (x >> (bits - r))) & m;
}

__device__ __forceinline__ int sbit_cost(int v, int bit)
{
	// ((in - (+-127))^2 >> 9), erasures cost nothing (libosmocore conv.c, generic decoder)
	const int e = bit ? v + 127 : v - 127;
	return v ? (__mul24(e, e) >> 9) : 0;
}

// ---------------------------------------------------------------------------
// FACCH3
// ---------------------------------------------------------------------------
static constexpr int kF3Steps = 96;       // 92 bits + 4 flush

// K=5 rate-1/4 (g0 = 1+D^3+D^4, g1 = 1+D+D^2+D^4, g2 = 1+D^2+D^4, g3 = 1+D+D^2+D^3+D^4; conv.c:174-198)
__device__ __forceinline__ uint32_t out_k5_14(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	return ((uint32_t)(__popc(reg & 0x19u) & 1) << 3) | ((uint32_t)(__popc(reg & 0x17u) & 1) << 2) |
	       ((uint32_t)(__popc(reg & 0x15u) & 1) << 1) | (uint32_t)(__popc(reg & 0x1fu) & 1);
}
// byte j of the mask = 1 when coded bit j (MSB of ov first) is a 1
__device__ __forceinline__ uint32_t ones_mask4(uint32_t ov)
{
	return ((ov >> 3) & 1u) | (((ov >> 2) & 1u) << 8) | (((ov >> 1) & 1u) << 16) | ((ov & 1u) << 24);
}

template <int PH, bool EDGE>
__device__ __forceinline__ unsigned long long acs14(uint32_t &ae, uint2 cst, uint32_t m1_own, uint32_t m1_par,
                                                    unsigned long long own_is_hi, bool b_is_one, bool flush)
{
	const uint32_t par = row_xor<(8 >> PH)>(ae);
	// branch metric = sum over the 4 coded bits of c0 (bit = 0) or c1 (bit = 1): two dot4 each
	uint32_t n_own = __builtin_amdgcn_udot4(cst.x, 0x01010101u ^ m1_own, ae, false);
	n_own = __builtin_amdgcn_udot4(cst.y, m1_own, n_own, false);
	uint32_t n_par = __builtin_amdgcn_udot4(cst.x, 0x01010101u ^ m1_par, par, false);
	n_par = __builtin_amdgcn_udot4(cst.y, m1_par, n_par, false);
	const unsigned long long own_lt = __ballot(n_own < n_par);
	const unsigned long long par_lt = __ballot(n_par < n_own);
	uint32_t nw = n_own < n_par ? n_own : n_par;
	if (EDGE) {
		nw = nw < kMaxAe ? nw : kMaxAe;
		if (flush && b_is_one)
			nw = kMaxAe;
	}
	ae = nw;
	return (own_lt & own_is_hi) | (par_lt & ~own_is_hi);
}

#define ACS14_4(EDGE, FL)                                                                         \
	do {                                                                                          \
		const uint2 c0 = cstr[k + 0], c1 = cstr[k + 1], c2 = cstr[k + 2], c3 = cstr[k + 3];       \
		const unsigned long long q0 = acs14<0, EDGE>(ae, c0, m_own[0], m_par[0], hi[0], b1[0], FL); \
		const unsigned long long q1 = acs14<1, EDGE>(ae, c1, m_own[1], m_par[1], hi[1], b1[1], FL); \
		const unsigned long long q2 = acs14<2, EDGE>(ae, c2, m_own[2], m_par[2], hi[2], b1[2], FL); \
		const unsigned long long q3 = acs14<3, EDGE>(ae, c3, m_own[3], m_par[3], hi[3], b1[3], FL); \
		if (lane == 0) {                                                                          \
			uint4 *sp = reinterpret_cast<uint4 *>(surv + k);                                      \
			sp[0] = make_uint4((uint32_t)q0, (uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32)); \
			sp[1] = make_uint4((uint32_t)q2, (uint32_t)(q2 >> 32), (uint32_t)q3, (uint32_t)(q3 >> 32)); \
		}                                                                                         \
	} while (0)

__global__ __launch_bounds__(64) void k_facch3(Facch3Args a)
{
	__shared__ __align__(16) int8_t s_eb[4][416];
	__shared__ __align__(16) uint2 s_cst[4][kF3Steps];      // per step: 4 x c0 bytes, 4 x c1 bytes
	__shared__ __align__(16) uint64_t s_surv[kF3Steps];
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
		uint32_t c0w = 0, c1w = 0;
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
			c0w |= (uint32_t)sbit_cost(v, 0) << (8 * j);
			c1w |= (uint32_t)sbit_cost(v, 1) << (8 * j);
		}
		s_cst[q][k] = make_uint2(c0w, c1w);
	}
	WSYNC();

	// ---- forward pass (same in-place layout as decode4_k5_12 in rx_kernels.hip)
	uint32_t m_own[4], m_par[4];
	bool b1[4];
	unsigned long long hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		const uint32_t s = rotl_n(loc, ph, 4);
		const uint32_t b = s >> 3;
		b1[ph] = b != 0;
		hi[ph] = __ballot(b1[ph]);
		m_own[ph] = ones_mask4(out_k5_14(s, b));
		m_par[ph] = ones_mask4(out_k5_14(s ^ 8u, b));
	}
	uint32_t ae = loc ? kMaxAe : 0u;
	const uint2 *cstr = &s_cst[row][0];
	uint64_t *surv = s_surv;
	{
		int k = 0;
		ACS14_4(true, false);
		for (k = 4; k < 92; k += 4)
			ACS14_4(false, false);
		ACS14_4(true, true);      // k = 92: flush
	}
	const uint32_t final_ae = ae;
	WSYNC();

	// ---- traceback in location space; u[k-4] = decision of step k (see rx_kernels.hip)
	if (loc == 0) {
		const uint16_t *s16 = reinterpret_cast<const uint16_t *>(s_surv) + row;
		uint32_t L = 0, ub = 0;
#define TB_STEP(W, PB)                                           \
		do {                                                     \
			const uint32_t x = ((uint32_t)(W) << (PB)) >> L;     \
			L = (L & ~(1u << (PB))) | (x & (1u << (PB)));        \
			ub = (ub << 1) | ((x >> (PB)) & 1u);                 \
		} while (0)
		for (int g = 23; g >= 1; g--) {
			const int k = 4 * g;
			const uint32_t w0 = s16[4 * (k + 0)], w1 = s16[4 * (k + 1)];
			const uint32_t w2 = s16[4 * (k + 2)], w3 = s16[4 * (k + 3)];
			TB_STEP(w3, 0);
			TB_STEP(w2, 1);
			TB_STEP(w1, 2);
			TB_STEP(w0, 3);
			if (((g - 1) & 7) == 0) {
				s_ub[row][(g - 1) >> 3] = ub;
				ub = 0;
			}
		}
#undef TB_STEP
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
// TCH3 speech
// ---------------------------------------------------------------------------
static constexpr int kT3Steps = 48;

// K=7 rate-1/2 (g0 = 1+D^2+D^3+D^5+D^6, g1 = 1+D+D^2+D^3+D^6; conv.c:518-571)
__device__ __forceinline__ uint32_t out_k7_12(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	return ((uint32_t)(__popc(reg & 0x6du) & 1) << 1) | (uint32_t)(__popc(reg & 0x4fu) & 1);
}

// soft bit c[kc] of frame `fr` (tch3.c:141-172): returns the descrambled / deciphered value
__device__ __forceinline__ int tch3_c(const int8_t *__restrict__ e, const uint8_t *__restrict__ ciph,
                                      int fr, int m, int kc)
{
	const int ii = kc % 24, ij = kc / 24;
	const int kep = (ii < 8) ? (ij + 5 * ii) : (ij + 4 * ii + 8);   // bits_c[kc] = bits_ep[kep]
	const int q = m ? (104 * fr + kep) : ((kep << 1) + fr);          // index into epp / xmy
	int v = e[q < 52 ? q : q + 4];                                   // xmy = e[0..51] | e[56..211]
	bool flip = (c_scr.w[q >> 5] >> (q & 31)) & 1u;
	if (ciph)
		flip ^= ciph[q] != 0;
	return flip ? (int)(int8_t)(-v) : v;
}

template <int PH, bool EDGE>
__device__ __forceinline__ unsigned long long acs_k7(uint32_t &ae, uint32_t bmw, uint32_t sh_own, uint32_t sh_par,
                                                     unsigned long long own_is_hi)
{
	const uint32_t par = wave_xor<(32 >> PH)>(ae);
	const uint32_t n_own = ae + ((bmw >> sh_own) & 0xffu);
	const uint32_t n_par = par + ((bmw >> sh_par) & 0xffu);
	const unsigned long long own_lt = __ballot(n_own < n_par);
	const unsigned long long par_lt = __ballot(n_par < n_own);
	uint32_t nw = n_own < n_par ? n_own : n_par;
	if (EDGE)
		nw = nw < kMaxAe ? nw : kMaxAe;
	ae = nw;
	return (own_lt & own_is_hi) | (par_lt & ~own_is_hi);
}

#define ACS_K7_6(EDGE, REC)                                                                       \
	do {                                                                                          \
		unsigned long long q[6];                                                                  \
		q[0] = acs_k7<0, EDGE>(ae, s_bm[k + 0], sh_own[0], sh_par[0], hi[0]);                     \
		q[1] = acs_k7<1, EDGE>(ae, s_bm[k + 1], sh_own[1], sh_par[1], hi[1]);                     \
		q[2] = acs_k7<2, EDGE>(ae, s_bm[k + 2], sh_own[2], sh_par[2], hi[2]);                     \
		q[3] = acs_k7<3, EDGE>(ae, s_bm[k + 3], sh_own[3], sh_par[3], hi[3]);                     \
		q[4] = acs_k7<4, EDGE>(ae, s_bm[k + 4], sh_own[4], sh_par[4], hi[4]);                     \
		q[5] = acs_k7<5, EDGE>(ae, s_bm[k + 5], sh_own[5], sh_par[5], hi[5]);                     \
		if (REC && lane == 0) {                                                                   \
			_Pragma("unroll") for (int u = 0; u < 6; u++) s_surv[k + u] = q[u];                   \
		}                                                                                         \
	} while (0)

__global__ __launch_bounds__(64) void k_tch3(Tch3Args a)
{
	__shared__ __align__(16) int8_t s_e[216];
	__shared__ __align__(16) uint32_t s_bm[kT3Steps];
	__shared__ __align__(16) uint64_t s_surv[kT3Steps];
	const int lane = threadIdx.x;
	const int g = blockIdx.x >> 1, fr = blockIdx.x & 1;
	const int m = a.m;
	const uint8_t *ciph = a.ciph ? a.ciph + (size_t)g * 208 : nullptr;

	{
		const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)g * 212);
		uint32_t *dst = reinterpret_cast<uint32_t *>(s_e);
		if (lane < 53)
			dst[lane] = src[lane];
	}
	WSYNC();

	// status bits (tch3.c:133-134), written by the frame-0 wave
	if (fr == 0 && a.bits_s && lane < 4)
		a.bits_s[(size_t)g * 4 + lane] = s_e[52 + lane] < 0;

	// ---- branch metrics: step s has coded bits 2s (always sent) and 2s+1 (punctured when
	// 2s+1 = 3 mod 4, i.e. s odd); the sent bits are c[idx - (idx>>2)]   (punct.c:48-133, P(1;2))
	if (lane < kT3Steps) {
		const int s = lane;
		const int i0 = 2 * s, i1 = 2 * s + 1;
		const int v0 = tch3_c(s_e, ciph, fr, m, i0 - (i0 >> 2));
		const int v1 = (s & 1) ? 0 : tch3_c(s_e, ciph, fr, m, i1 - (i1 >> 2));
		const int a0 = sbit_cost(v0, 0), a1 = sbit_cost(v0, 1);
		const int b0 = sbit_cost(v1, 0), b1c = sbit_cost(v1, 1);
		s_bm[s] = (uint32_t)(a0 + b0) | ((uint32_t)(a0 + b1c) << 8) |
		          ((uint32_t)(a1 + b0) << 16) | ((uint32_t)(a1 + b1c) << 24);
	}
	WSYNC();

	// ---- in-place 64-state trellis: state s sits in lane rotr6^k(s) at step k
	const uint32_t loc = (uint32_t)lane;
	uint32_t sh_own[6], sh_par[6];
	unsigned long long hi[6];
#pragma unroll
	for (int ph = 0; ph < 6; ph++) {
		const uint32_t s = rotl_n(loc, ph, 6);
		const uint32_t b = s >> 5;
		hi[ph] = __ballot(b != 0);
		sh_own[ph] = 8u * out_k7_12(s, b);
		sh_par[ph] = 8u * out_k7_12(s ^ 32u, b);
	}
	uint32_t ae = loc ? kMaxAe : 0u;          // D4: first pass starts in state 0
	// pass 1 (warm-up), decisions not kept
	{
		int k = 0;
		ACS_K7_6(true, false);
		for (k = 6; k < kT3Steps; k += 6)
			ACS_K7_6(false, false);
	}
	// rewind: subtract the minimum (osmo_conv_decode_rewind)
	{
		uint32_t mn = ae;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const uint32_t ov = (uint32_t)__shfl_xor((int)mn, o);
			mn = ov < mn ? ov : mn;
		}
		ae -= mn;
	}
	// pass 2
	for (int k = 0; k < kT3Steps; k += 6)
		ACS_K7_6(false, true);
	WSYNC();

	// best end state: smallest metric, lowest state on ties (48 = 8 * 6 steps: layout = identity)
	unsigned long long key = ((unsigned long long)ae << 32) | loc;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned long long ok = __shfl_xor(key, o);
		key = ok < key ? ok : key;
	}
	const uint32_t end_state = (uint32_t)key & 63u;
	const uint32_t min_ae = (uint32_t)(key >> 32);

	// ---- traceback (lane 0): u[k-6] = decision of step k for k = 47..6; u[47-j] = bit j of the end state
	__shared__ uint32_t s_d[3];       // 80 decoded bits, bit k of the frame at word k>>5, bit k&31
	if (lane == 0) {
		uint32_t L = end_state;
		uint32_t lo = 0, hi32 = 0;    // bits 0..31, 32..47
		for (int j = 0; j < 6; j++) {
			const uint32_t bit = (end_state >> j) & 1u;
			hi32 |= bit << (47 - j - 32);
		}
		for (int k = kT3Steps - 1; k >= 6; k--) {
			const int pb = 5 - (k % 6);
			const uint32_t d = (uint32_t)((s_surv[k] >> L) & 1ull);
			L = (L & ~(1u << pb)) | (d << pb);
			const int j = k - 6;
			if (j < 32) lo |= d << j; else hi32 |= d << (j - 32);
		}
		s_d[0] = lo;
		s_d[1] = hi32;
	}
	WSYNC();

	// ---- class-2 bits: d[48..79] = c[72..103] < 0 (tch3.c:178-179); pack MSB first (osmo_ubit2pbit)
	uint32_t bitv = 0;
	if (lane < 48) {
		bitv = (s_d[lane >> 5] >> (lane & 31)) & 1u;
	}
	const unsigned long long m_lo = __ballot(bitv != 0);          // bits 0..47 in lanes 0..47
	uint32_t hv = 0;
	if (lane < 32)
		hv = tch3_c(s_e, ciph, fr, m, 72 + lane) < 0;
	const unsigned long long m_hi = __ballot(hv != 0);            // bits 48..79 in lanes 0..31
	if (lane < 10) {
		// byte `lane` holds frame bits 8*lane .. 8*lane+7, first bit in the MSB
		uint32_t byte = 0;
#pragma unroll
		for (int t = 0; t < 8; t++) {
			const int k = 8 * lane + t;
			const uint32_t bit = k < 48 ? (uint32_t)((m_lo >> k) & 1ull) : (uint32_t)((m_hi >> (k - 48)) & 1ull);
			byte |= bit << (7 - t);
		}
		a.frames[((size_t)g * 2 + fr) * 10 + lane] = (uint8_t)byte;
	}
	if (lane == 0 && a.conv)
		a.conv[(size_t)g * 2 + fr] = (int32_t)min_ae;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_facch3(const Facch3Args &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_facch3, dim3((a.n + 3) / 4), dim3(64), 0, st, a);
	return hipGetLastError();
}

hipError_t launch_tch3(const Tch3Args &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_tch3, dim3(2 * a.n), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace gmr1
