// tch3_body.h -- the TCH3 speech decoder of one burst on one wavefront (reference src/l1/tch3.c:124-183), as a device
// function: k_tch3 (l1_kernels.hip) runs it on soft bits read from HBM, k_rx4g_tch3 (rx_kernels.hip) right behind the
// demodulation of the same burst, on the soft bits still in LDS.
#pragma once

#include "gmr1_dev.h"

#ifndef WSYNC
#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)
#endif

namespace gmr1 {
namespace t3 {

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

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);   // every lane has a source: nothing to preset
}

// ---------------------------------------------------------------------------
// TCH3 speech
// ---------------------------------------------------------------------------
static constexpr int kT3Steps = 48;

// where soft bit c[kc] of frame fr sits (tch3.c:141-172), for both multiplexing modes: index q into the
// 208 bits of xmy (= cipher stream position), bit 8: the scrambler flips it.  The 104-bit permutation
// (kep), the frame de-multiplexing and the scrambler are folded into one table lookup.
struct T3Map { uint16_t q[2][2][104]; };       // [m][fr][kc]
static constexpr T3Map make_t3map()
{
	T3Map t{};
	const ScrBits scr = make_scr();
	for (int m = 0; m < 2; m++)
		for (int fr = 0; fr < 2; fr++)
			for (int kc = 0; kc < 104; kc++) {
				const int ii = kc % 24, ij = kc / 24;
				const int kep = (ii < 8) ? (ij + 5 * ii) : (ij + 4 * ii + 8);   // bits_c[kc] = bits_ep[kep]
				const int q = m ? (104 * fr + kep) : ((kep << 1) + fr);          // index into epp / xmy
				const uint32_t flip = (scr.w[q >> 5] >> (q & 31)) & 1u;
				t.q[m][fr][kc] = (uint16_t)(q | (flip << 8));
			}
	return t;
}
static __constant__ T3Map c_t3map = make_t3map();

// The same per TRELLIS STEP s (coded bits 2s, always sent, and 2s+1, punctured when s is odd; the sent bits are
// c[idx - (idx >> 2)], punct.c:48-133 with P(1;2)):  bits 0-7 index of the first soft bit in the burst's 212 e-bits,
// bit 8 the scrambler flips it, bits 16-23 / 24 the same for the second one -- a punctured second bit points at byte 212
// of the LDS copy, which is kept zero (an erasure costs nothing).
struct T3Steps { uint32_t w[2][2][kT3Steps]; };  // [m][fr][s]
static constexpr T3Steps make_t3steps()
{
	const T3Map mp = make_t3map();
	T3Steps t{};
	for (int m = 0; m < 2; m++)
		for (int fr = 0; fr < 2; fr++)
			for (int s = 0; s < kT3Steps; s++) {
				const int i0 = 2 * s, i1 = 2 * s + 1;
				const uint32_t m0 = mp.q[m][fr][i0 - (i0 >> 2)];
				const uint32_t q0 = m0 & 0xffu;
				uint32_t w = (q0 < 52 ? q0 : q0 + 4) | (m0 & 0x100u);
				if (s & 1) {
					w |= 212u << 16;
				} else {
					const uint32_t m1 = mp.q[m][fr][i1 - (i1 >> 2)];
					const uint32_t q1 = m1 & 0xffu;
					w |= ((q1 < 52 ? q1 : q1 + 4) | (m1 & 0x100u)) << 16;
				}
				t.w[m][fr][s] = w;
			}
	return t;
}
static __constant__ T3Steps c_t3steps = make_t3steps();

// What a soft bit contributes to a step's table: index = the soft bit as uint8, + 256 when scrambler / cipher flip it
// ((int8)(-v): -128 stays -128 as in gmr1_scramble_sbit); entry = (cost as a 1 - cost as a 0) in the low half, their sum
// in the high half.  [0]: the generic decoder's ((in -+ 127)^2 >> 9, erasure 0), [1]: the accelerated one's (|in| for the
// contradicted value).
struct T3Cost { uint32_t w[2][512]; };
static constexpr T3Cost make_t3cost()
{
	T3Cost t{};
	for (int idx = 0; idx < 512; idx++) {
		int v = (int)(int8_t)(uint8_t)(idx & 255);
		if (idx & 256)
			v = (int)(int8_t)(uint8_t)(-v);
		const int e0 = v - 127, e1 = v + 127;
		const int c0 = v ? ((e0 * e0) >> 9) : 0, c1 = v ? ((e1 * e1) >> 9) : 0;
		t.w[0][idx] = ((uint32_t)(c1 - c0) & 0xffffu) | ((uint32_t)(c0 + c1) << 16);
		const int a0 = v < 0 ? -v : 0, a1 = v > 0 ? v : 0;
		t.w[1][idx] = ((uint32_t)(a1 - a0) & 0xffffu) | ((uint32_t)(a0 + a1) << 16);
	}
	return t;
}
static __constant__ T3Cost c_t3cost = make_t3cost();

// soft bit c[kc] of frame `fr`: the descrambled / deciphered value
__device__ __forceinline__ int tch3_c(const int8_t *__restrict__ e, const uint8_t *__restrict__ ciph,
                                      int fr, int m, int kc)
{
	const uint32_t me = c_t3map.q[m][fr][kc];
	const int q = (int)(me & 0xffu);
	int v = e[q < 52 ? q : q + 4];                                   // xmy = e[0..51] | e[56..211]
	bool flip = (me >> 8) != 0;
	if (ciph)
		flip ^= ciph[q] != 0;
	return flip ? (int)(int8_t)(-v) : v;
}

// K=7 rate-1/2: g0 = 1+D^2+D^3+D^5+D^6, g1 = 1+D+D^2+D^3+D^6 (conv.c:518-571).
// One burst per wavefront: its two speech frames occupy the two 32-lane halves, and the 64 trellis states of
// a frame live two to a lane.  Position bits c0..c4 = lane within the half (xor masks 16, 8, 7, 2, 1),
// c5 = register index; with the in-place butterfly the predecessor state a position holds in phase
// ph = step % 6 has bit i = c[(5 - i + ph) % 6], so the two predecessors of a position differ in position bit
// ph: lane xor 16 (ds_bpermute), 8, 7, 2, 1 (one DPP control each) for phases 0-4 and the register index for
// phase 5 (no cross-lane traffic).  After 6 steps the layout is back where it started.
//   o[r][p]   : code word (2 bits) of the own transition per phase
//   st[r][p]  : state held in phase 0;  loc_of[state] = r * 32 + p
struct K7Tab { uint16_t o[2][32]; uint8_t st[2][32]; uint8_t loc_of[64]; };
static constexpr uint32_t k7_out(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	uint32_t p0 = reg & 0x6du, p1 = reg & 0x4fu;
	p0 ^= p0 >> 4; p0 ^= p0 >> 2; p0 ^= p0 >> 1;
	p1 ^= p1 >> 4; p1 ^= p1 >> 2; p1 ^= p1 >> 1;
	return ((p0 & 1u) << 1) | (p1 & 1u);
}
static constexpr K7Tab make_k7()
{
	K7Tab t{};
	for (uint32_t r = 0; r < 2; r++)
		for (uint32_t p = 0; p < 32; p++) {
			uint32_t c[6] = {0, 0, 0, 0, 0, 0};
			c[0] = (p >> 4) & 1u;
			c[1] = (p >> 3) & 1u;
			uint32_t x = p & 7u;
			c[2] = (x >> 2) & 1u;
			x ^= c[2] ? 7u : 0u;
			c[3] = (x >> 1) & 1u;
			c[4] = x & 1u;
			c[5] = r;
			uint32_t e = 0;
			for (int ph = 0; ph < 6; ph++) {
				uint32_t sp = 0;
				for (int i = 0; i < 6; i++)
					sp |= c[(5 - i + ph) % 6] << i;
				e |= k7_out(sp, sp >> 5) << (2 * ph);
				if (ph == 0) {
					t.st[r][p] = (uint8_t)sp;
					t.loc_of[sp] = (uint8_t)(r * 32 + p);
				}
			}
			t.o[r][p] = (uint16_t)e;
		}
	return t;
}
static __constant__ K7Tab c_k7 = make_k7();

// The two states of a lane differ in position bit c5, which is state bit `ph` in phase ph: by linearity their own code
// words differ by the code word of that unit state, whatever the lane.  The packed warm-up pass (k7_step_pk) builds on it.
static constexpr uint32_t k7_pair_mask(int ph)
{
	return k7_out(1u << ph, (1u << ph) >> 5);
}
static constexpr bool k7_pair_mask_holds()
{
	const K7Tab t = make_k7();
	for (int p = 0; p < 32; p++)
		for (int ph = 0; ph < 6; ph++)
			if ((((uint32_t)t.o[1][p] >> (2 * ph)) & 3u) != ((((uint32_t)t.o[0][p] >> (2 * ph)) & 3u) ^ k7_pair_mask(ph)))
				return false;
	return true;
}
static_assert(k7_pair_mask_holds(), "code words of a lane's two states must differ by a per-phase constant");

// the word of the position whose lane differs in position bit PH (PH < 5)
template <int PH>
__device__ __forceinline__ uint32_t k7_partner(uint32_t w)
{
	if constexpr (PH == 0) return (uint32_t)__shfl_xor((int)w, 16);
	else if constexpr (PH == 1) return dpp<0x128>(w);       // row_ror:8
	else if constexpr (PH == 2) return dpp<0x141>(w);       // row_half_mirror: xor 7
	else if constexpr (PH == 3) return dpp<0x4E>(w);        // quad_perm [2,3,0,1]
	else return dpp<0xB1>(w);                               // quad_perm [1,0,3,2]
}

typedef __attribute__((address_space(3))) const uint32_t t3_lds_cu32;

// One trellis step at window position J (phase J % 6) on the packed words [metric:16 | decisions of the
// current 12-step window:16] (see decode4_k5_12 in rx_kernels.hip).  Both generators have the D^0 and D^6
// taps, so the two transitions into a state carry complementary code words and their costs add up to a
// per-step constant K: the words hold 2 * metric - sum K, a candidate is `own + m` / `partner - m` with ONE
// table value m = (2 cost - K) << 16 (the subtraction takes the DPP operand directly), comparisons and ties
// are those of the plain metric.  The HIGH predecessor carries the tie-break / decision bit of the position,
// so v_min_u32 selects, breaks ties towards the low predecessor and records the decision at once (REC).
// ad[ph][r]: LDS byte address of the position's table value in step 0.
template <int J, int K, bool REC>
__device__ __forceinline__ void k7_step(uint32_t (&w)[2], const uint32_t (&ad)[6][2], const uint32_t (&hi)[5])
{
	constexpr int PH = J % 6;
	const uint32_t m0 = *(t3_lds_cu32 *)(uintptr_t)(ad[PH][0] + 16u * K);
	const uint32_t m1 = *(t3_lds_cu32 *)(uintptr_t)(ad[PH][1] + 16u * K);
	if constexpr (PH < 5) {
		const uint32_t v0 = REC ? w[0] + (hi[PH] << J) : w[0], v1 = REC ? w[1] + (hi[PH] << J) : w[1];
		const uint32_t a1 = v0 + m0, a2 = k7_partner<PH>(v0) - m0;
		const uint32_t b1 = v1 + m1, b2 = k7_partner<PH>(v1) - m1;
		w[0] = a1 < a2 ? a1 : a2;
		w[1] = b1 < b2 ? b1 : b2;
	} else {
		const uint32_t vl = w[0], vh = REC ? w[1] + (1u << J) : w[1];
		const uint32_t a1 = vl + m0, a2 = vh - m0;
		const uint32_t b1 = vh + m1, b2 = vl - m1;
		w[0] = a1 < a2 ? a1 : a2;
		w[1] = b1 < b2 ? b1 : b2;
	}
}

// twelve steps starting at step K0
template <int K0, bool REC>
__device__ __forceinline__ void k7_window(uint32_t (&w)[2], const uint32_t (&ad)[6][2], const uint32_t (&hi)[5])
{
	k7_step<0, K0 + 0, REC>(w, ad, hi); k7_step<1, K0 + 1, REC>(w, ad, hi); k7_step<2, K0 + 2, REC>(w, ad, hi);
	k7_step<3, K0 + 3, REC>(w, ad, hi); k7_step<4, K0 + 4, REC>(w, ad, hi); k7_step<5, K0 + 5, REC>(w, ad, hi);
	k7_step<6, K0 + 6, REC>(w, ad, hi); k7_step<7, K0 + 7, REC>(w, ad, hi); k7_step<8, K0 + 8, REC>(w, ad, hi);
	k7_step<9, K0 + 9, REC>(w, ad, hi); k7_step<10, K0 + 10, REC>(w, ad, hi); k7_step<11, K0 + 11, REC>(w, ad, hi);
}

// The warm-up pass needs metrics only, and a metric is 16 bits: both states of a lane in ONE register, low half the
// r = 0 state.  v_pk_add_u16 / v_pk_sub_u16 / v_pk_min_u16 do the two add-compare-selects at once and the partner costs
// one DPP move for both (VOP3P takes no DPP operand) - 4 VALU and one table read per step where the word form takes
// 6 and two.  The arithmetic is the word form's high half: sums wrap modulo 2^16 there as well, the minimum is unsigned.
// The table holds, per step and own code word c, (m[c ^ pair_mask] << 16) | m[c]  (TAB_DELTA bytes after the word table).
typedef unsigned short k7_us2 __attribute__((ext_vector_type(2)));
template <int J, int K, int TAB_DELTA>
__device__ __forceinline__ void k7_step_pk(uint32_t &P, const uint32_t (&ad)[6][2])
{
	constexpr int PH = J % 6;
	const uint32_t pm = *(t3_lds_cu32 *)(uintptr_t)(ad[PH][0] + 16u * K + (uint32_t)TAB_DELTA);
	uint32_t other;
	if constexpr (PH < 5)
		other = k7_partner<PH>(P);
	else
		other = __builtin_amdgcn_alignbit(P, P, 16);           // the lane's two states swap halves
	const k7_us2 m2 = __builtin_bit_cast(k7_us2, pm);
	const k7_us2 a = __builtin_bit_cast(k7_us2, P) + m2;
	const k7_us2 b = __builtin_bit_cast(k7_us2, other) - m2;
	P = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(a, b));
}

template <int K0, int TAB_DELTA>
__device__ __forceinline__ void k7_window_pk(uint32_t &P, const uint32_t (&ad)[6][2])
{
	k7_step_pk<0, K0 + 0, TAB_DELTA>(P, ad); k7_step_pk<1, K0 + 1, TAB_DELTA>(P, ad); k7_step_pk<2, K0 + 2, TAB_DELTA>(P, ad);
	k7_step_pk<3, K0 + 3, TAB_DELTA>(P, ad); k7_step_pk<4, K0 + 4, TAB_DELTA>(P, ad); k7_step_pk<5, K0 + 5, TAB_DELTA>(P, ad);
	k7_step_pk<6, K0 + 6, TAB_DELTA>(P, ad); k7_step_pk<7, K0 + 7, TAB_DELTA>(P, ad); k7_step_pk<8, K0 + 8, TAB_DELTA>(P, ad);
	k7_step_pk<9, K0 + 9, TAB_DELTA>(P, ad); k7_step_pk<10, K0 + 10, TAB_DELTA>(P, ad); k7_step_pk<11, K0 + 11, TAB_DELTA>(P, ad);
}

// minimum over the 32 lanes of each half
__device__ __forceinline__ uint32_t half_min(uint32_t v)
{
	uint32_t o;
	o = dpp<0xB1>(v); v = o < v ? o : v;
	o = dpp<0x4E>(v); v = o < v ? o : v;
	o = dpp<0x141>(v); v = o < v ? o : v;
	o = dpp<0x128>(v); v = o < v ? o : v;
	o = (uint32_t)__shfl_xor((int)v, 16); v = o < v ? o : v;
	return v;
}

// LDS of one burst's decode: per frame (2 cost(word) - K) << 16 per step and code word; behind both frames' tables the same
// again in the packed form of the warm-up pass (k7_step_pk), at a fixed distance so that one address register serves both;
// the window decisions; the position of a state (the survivor walk reads it four times in a row)
struct Tch3Lds {
	uint32_t tab_all[4][kT3Steps * 4];
	uint16_t win[2][4][64];
	uint8_t locof[64];
};

// one (burst, its two frames) on one wavefront; s_e: the burst's 212 soft bits in LDS, byte 212 zero (what a punctured
// position reads); S->locof filled by the caller (tch3_fill_locof)
__device__ __forceinline__ void tch3_fill_locof(Tch3Lds *S, int lane)
{
	const int p = lane & 31;
	S->locof[c_k7.st[0][p]] = (uint8_t)p;       // (both halves write the same values)
	S->locof[c_k7.st[1][p]] = (uint8_t)(32 + p);
}

// What a lane needs for every burst and that depends on nothing but the lane and where the tables sit: computed once per
// wave (tch3_lane), kept in registers over the bursts a wave decodes.  The values pass through an empty asm so that the
// compiler keeps them instead of recomputing them burst after burst (it did: 36 VALU per burst).
struct Tch3Lane {
	uint32_t ad[6][2];       // LDS byte address of the position's table value in step 0, per phase and state of the lane
	uint32_t hi[5];          // the position's own predecessor is the high one, per phase
};
__device__ __forceinline__ void tch3_lane(Tch3Lane &L, const Tch3Lds *S, int lane)
{
	const int fr = lane >> 5, p = lane & 31;
	{
		uint32_t x = (uint32_t)p & 7u;
		L.hi[0] = ((uint32_t)p >> 4) & 1u;
		L.hi[1] = ((uint32_t)p >> 3) & 1u;
		L.hi[2] = (x >> 2) & 1u;
		x ^= L.hi[2] ? 7u : 0u;
		L.hi[3] = (x >> 1) & 1u;
		L.hi[4] = x & 1u;
	}
	// (the tables are 16-byte aligned: the code word's four bytes are the address's low bits, and the lane's second state has
	// the first one's code word xor the phase's constant, k7_pair_mask)
	const uint32_t tab_base = (uint32_t)(uintptr_t)(t3_lds_cu32 *)S->tab_all[fr];
	const uint32_t e4 = (uint32_t)c_k7.o[0][p] << 2;
#pragma unroll
	for (int ph = 0; ph < 6; ph++) {
		L.ad[ph][0] = tab_base + ((e4 >> (2 * ph)) & 12u);
		L.ad[ph][1] = L.ad[ph][0] ^ (4u * k7_pair_mask(ph));
		asm volatile("" : "+v"(L.ad[ph][0]), "+v"(L.ad[ph][1]));
	}
#pragma unroll
	for (int i = 0; i < 5; i++)
		asm volatile("" : "+v"(L.hi[i]));
}

template <bool ACC>
__device__ __forceinline__ void tch3_burst(const Tch3Args &a, int g, int lane, const int8_t *__restrict__ s_e, Tch3Lds *S,
                                           const Tch3Lane &LC)
{
	uint32_t (*s_tab_all)[kT3Steps * 4] = S->tab_all;
	uint32_t (*s_tab)[kT3Steps * 4] = s_tab_all;
	constexpr int kPkDelta = 2 * kT3Steps * 4 * 4;
	uint16_t (*s_win)[4][64] = S->win;
	const uint8_t *s_locof = S->locof;
	const int fr = lane >> 5, p = lane & 31;   // the half-wave's frame, position within the half
	const int m = a.m;
	const uint8_t *ciph = a.ciph ? a.ciph + (size_t)g * 208 : nullptr;

	// status bits (tch3.c:133-134)
	if (a.bits_s && lane < 4)
		a.bits_s[(size_t)g * 4 + lane] = s_e[52 + lane] < 0;

	// ---- branch metrics.  Lane s < 48 owns trellis step s of both frames: one descriptor (c_t3steps) says where its two
	// soft bits sit, one table word per soft bit (c_t3cost) gives the cost difference d = cost as a 1 - cost as a 0 and the
	// cost sum; the four code words o = (g0 bit << 1) | g1 bit then cost (+-da +-db) (+ a constant that cancels).
	// Word table for the recording pass, and behind it the packed form of the warm-up pass: high half the lane's r = 1
	// state, whose code word is the own one xor the phase's constant -- i.e. da / db with their signs turned.
	uint32_t ksp = 0;                                                // sum of K: frame 0 in the low half, frame 1 in the high half
	if (lane < kT3Steps) {
		const int s = lane;
		constexpr uint32_t kPairMasks = k7_pair_mask(0) | (k7_pair_mask(1) << 2) | (k7_pair_mask(2) << 4) |
		                                (k7_pair_mask(3) << 6) | (k7_pair_mask(4) << 8) | (k7_pair_mask(5) << 10);
		const uint32_t ph = (uint32_t)s - 6u * (((uint32_t)s * 43u) >> 8);      // s % 6
		const uint32_t mk = (kPairMasks >> (2u * ph)) & 3u;
		const int sga = 1 - (int)(mk & 2u), sgb = 1 - 2 * (int)(mk & 1u);
		const uint32_t *ct = c_t3cost.w[ACC ? 1 : 0];
		const uint8_t *se = reinterpret_cast<const uint8_t *>(s_e);
#pragma unroll
		for (int f = 0; f < 2; f++) {
			const uint32_t st = c_t3steps.w[m][f][s];
			uint32_t ia = (uint32_t)se[st & 0xffu] | (st & 0x100u);
			uint32_t ib = (uint32_t)se[(st >> 16) & 0xffu] | ((st >> 16) & 0x100u);
			if (ciph) {
				const uint32_t e0 = st & 0xffu, e1 = (st >> 16) & 0xffu;
				ia ^= ciph[e0 < 52 ? e0 : e0 - 4] ? 0x100u : 0u;
				if (!(s & 1))
					ib ^= ciph[e1 < 52 ? e1 : e1 - 4] ? 0x100u : 0u;
			}
			const uint32_t A = ct[ia], B = ct[ib];
			const int da = (int)(int16_t)A, db = (int)(int16_t)B;
			if constexpr (!ACC)                                          // (the accelerated decoder returns no metric: nothing to sum)
				ksp += ((A >> 16) + (B >> 16)) << (16 * f);
			const int sum = da + db, dif = da - db;
			*reinterpret_cast<uint4 *>(&s_tab[f][4 * s]) =
			    make_uint4((uint32_t)(-sum) << 16, (uint32_t)(-dif) << 16, (uint32_t)dif << 16, (uint32_t)sum << 16);
			const int dap = da * sga, dbp = db * sgb;
			const int sump = dap + dbp, difp = dap - dbp;
			// (high << 16) | (low & 0xffff)
			*reinterpret_cast<uint4 *>(&s_tab_all[2 + f][4 * s]) =
			    make_uint4(__builtin_amdgcn_perm((uint32_t)(-sump), (uint32_t)(-sum), 0x05040100u),
			               __builtin_amdgcn_perm((uint32_t)(-difp), (uint32_t)(-dif), 0x05040100u),
			               __builtin_amdgcn_perm((uint32_t)difp, (uint32_t)dif, 0x05040100u),
			               __builtin_amdgcn_perm((uint32_t)sump, (uint32_t)sum, 0x05040100u));
		}
	}
	// both sums over the wave at once: rows by DPP, the four row totals through the scalar unit
	int ksum = 0;
	if constexpr (!ACC) {
		ksp += dpp<0xB1>(ksp);
		ksp += dpp<0x4E>(ksp);
		ksp += dpp<0x141>(ksp);
		ksp += dpp<0x140>(ksp);
		const uint32_t ks_all = (uint32_t)__builtin_amdgcn_readlane((int)ksp, 0) + (uint32_t)__builtin_amdgcn_readlane((int)ksp, 16) +
		                        (uint32_t)__builtin_amdgcn_readlane((int)ksp, 32) + (uint32_t)__builtin_amdgcn_readlane((int)ksp, 48);
		ksum = (int)(fr ? ks_all >> 16 : ks_all & 0xffffu);
	}
	WSYNC();

	// ---- per-lane constants (tch3_lane)
	const uint32_t (&hi)[5] = LC.hi;
	const uint32_t (&ad)[6][2] = LC.ad;
	uint32_t w[2];
	constexpr uint32_t kSent = 0xF000u;           // unreachable (libosmocore: MAX_AE)
	constexpr uint32_t kBias = 0x4000u;           // |2 cost - K| <= 252 (ACC: 256) per step, 48 steps: stays inside 16 bits
#pragma unroll
	for (int r = 0; r < 2; r++)
		// pass 1 starts from state 0 (D4); conv_acc.c: from every state alike
		w[r] = (!ACC && c_k7.st[r][p] ? kSent : kBias) << 16;

	// pass 1 (warm-up): only the metrics matter - both states of the lane in one register (k7_step_pk)
	{
		uint32_t P = (w[1] & 0xffff0000u) | (w[0] >> 16);
		k7_window_pk<0, kPkDelta>(P, ad);
		k7_window_pk<12, kPkDelta>(P, ad);
		k7_window_pk<24, kPkDelta>(P, ad);
		k7_window_pk<36, kPkDelta>(P, ad);
		w[0] = P << 16;
		w[1] = P & 0xffff0000u;
	}
	// rewind: subtract the minimum (osmo_conv_decode_rewind)
	{
		const uint32_t mn = half_min((w[0] < w[1] ? w[0] : w[1]) >> 16);
		w[0] = ((w[0] >> 16) - mn + kBias) << 16;
		w[1] = ((w[1] >> 16) - mn + kBias) << 16;
	}
	// pass 2: four windows of 12 steps; window m's decisions at a position are u[12m-6 .. 12m+5] of the
	// path ending there, and the first six name the state at the start of the window
	k7_window<0, true>(w, ad, hi);
	s_win[fr][0][p] = (uint16_t)w[0]; s_win[fr][0][32 + p] = (uint16_t)w[1];
	w[0] &= 0xffff0000u; w[1] &= 0xffff0000u;
	k7_window<12, true>(w, ad, hi);
	s_win[fr][1][p] = (uint16_t)w[0]; s_win[fr][1][32 + p] = (uint16_t)w[1];
	w[0] &= 0xffff0000u; w[1] &= 0xffff0000u;
	k7_window<24, true>(w, ad, hi);
	s_win[fr][2][p] = (uint16_t)w[0]; s_win[fr][2][32 + p] = (uint16_t)w[1];
	w[0] &= 0xffff0000u; w[1] &= 0xffff0000u;
	k7_window<36, true>(w, ad, hi);
	s_win[fr][3][p] = (uint16_t)w[0]; s_win[fr][3][32 + p] = (uint16_t)w[1];
	WSYNC();

	// best end state of each frame: smallest metric, lowest state on ties (48 = 8 * 6 steps: every state is
	// back in its phase-0 position)
	// (ACC: best sum, the first in conv_acc.c's own state numbering -- newest bit on top, i.e. bit-reversed -- on ties)
	uint32_t key;
	{
		const uint32_t s0 = c_k7.st[0][p], s1 = c_k7.st[1][p];
		const uint32_t k0 = (w[0] & 0xffff0000u) | (ACC ? __brev(s0) >> 26 : s0);
		const uint32_t k1 = (w[1] & 0xffff0000u) | (ACC ? __brev(s1) >> 26 : s1);
		key = half_min(k1 < k0 ? k1 : k0);
	}
	const uint32_t end_state = ACC ? __brev(key & 63u) >> 26 : key & 63u;
	// words hold 2 * ae - sum K (+ bias); osmo_conv_decode_acc returns 0
	const int32_t min_ae = ACC ? 0 : ((int)(key >> 16) - (int)kBias + ksum) >> 1;

	// ---- survivor chain (uniform across the half-wave): four dependent 16-bit reads.
	// u[42..47] are the end state's bits (bit j = u[47 - j]); window m gives u[12m-6 .. 12m+5] LSB first
	unsigned long long u;
	{
		uint32_t L = s_locof[end_state];
		const uint32_t h3 = s_win[fr][3][L];
		L = s_locof[__brev(h3 & 63u) >> 26];
		const uint32_t h2 = s_win[fr][2][L];
		L = s_locof[__brev(h2 & 63u) >> 26];
		const uint32_t h1 = s_win[fr][1][L];
		L = s_locof[__brev(h1 & 63u) >> 26];
		const uint32_t h0 = s_win[fr][0][L];
		// u[0..5] = h0 >> 6, u[6..17] = h1, u[18..29] = h2, u[30..41] = h3, u[42..47] = rev6(end_state)
		u = (unsigned long long)((h0 >> 6) & 63u) | ((unsigned long long)(h1 & 0xfffu) << 6) |
		    ((unsigned long long)(h2 & 0xfffu) << 18) | ((unsigned long long)(h3 & 0xfffu) << 30) |
		    ((unsigned long long)(__brev(end_state) >> 26) << 42);
	}

	// ---- class-2 bits: d[48..79] = c[72..103] < 0 (tch3.c:178-179): 32 per frame = one per lane of the half;
	// pack MSB first (osmo_ubit2pbit)
	const uint32_t hv = tch3_c(s_e, ciph, fr, m, 72 + p) < 0;
	const unsigned long long m_all = __ballot(hv != 0);
	const uint32_t m_hi = (uint32_t)(m_all >> (32 * fr));            // bits 48..79 of this half's frame
	if (p < 10) {
		// byte p holds frame bits 8p .. 8p+7, first bit in the MSB: eight bits, reversed
		const uint32_t raw = p < 6 ? (uint32_t)(u >> (8 * p)) : (m_hi >> (8 * (p - 6)));
		a.frames[((size_t)g * 2 + fr) * 10 + p] = (uint8_t)(__brev(raw & 0xffu) >> 24);
	}
	if (p == 0 && a.conv)
		a.conv[(size_t)g * 2 + fr] = min_ae;
}


}  // namespace t3
}  // namespace gmr1
