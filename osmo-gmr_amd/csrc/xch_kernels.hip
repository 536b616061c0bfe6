// xch_kernels.hip -- layer-1 decoder of the xCH channels carried on a DC12 burst (reference
// src/l1/xch_dc12.c:81-108), gfx950:  descramble 432 soft bits, intra-burst de-interleave (N = 54),
// K = 9 rate 1/3 tail-biting Viterbi over 208 bits with P(12;13) puncturing (conv.c:345-429,
// punct.c:1105-1125), CRC16, LSB-first packing into 24 bytes.
//
// One burst per wavefront.  The 256 trellis states live four to a lane as packed words
// [metric:16 | decisions of the current 16-step window:16] and never move: with the in-place butterfly
// the state a (lane, register) position holds rotates by one bit per step, so in phase ph = step % 8
// the two predecessors of a position differ in position bit ph -- lane xor 32, 16 (ds_bpermute), 8, 7,
// 2, 1 (one DPP control each) for phases 0-5, the register index for phases 6 and 7 (no cross-lane
// traffic at all).  After 8 steps every state is back where it started.
//
// Branch metric: all three generators have the D^0 and D^8 taps, so the two transitions into a state
// carry complementary code words and cost(w) + cost(~w) = K is the same for the whole step.  The words
// hold 2 * metric - sum K: a candidate is `own + m` or `partner - m` with ONE table value m = 2 cost - K
// per position and step, comparisons (and ties) are those of the plain metric, and conv_rv is recovered
// at the end.  The HIGH-predecessor position carries the decision bit of the step, so v_min_u32 selects,
// breaks ties towards the low predecessor (libosmocore's generic decoder keeps the first) and records
// the decision at once.
//
// Tail biting as osmo_conv_decode does it: one pass from state 0 for the metrics only, minimum
// subtracted, a second pass that records; best end state (lowest on ties); the survivor path is read
// back through 13 chained 16-bit LDS reads (a window's first 8 decisions name the state it started in).
#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr int kXchLen = 208;               // data bits = trellis steps per pass
static constexpr int kXchWin = kXchLen / 16;      // 13 windows
static constexpr uint32_t kBias = 0x6000u;        // metrics are renormalised to min = kBias every 64 steps
static constexpr uint32_t kUnreach = 0xF000u;     // libosmocore: MAX_AE

// ---- where the soft bit of coded bit ii = 3 * step + j sits in the burst ------------------------------
// bit 15 punctured; bits 0-8 index into the 432 e-bits; bit 9: the scrambler flips it
struct XchMap { uint16_t m[3 * kXchLen]; };
static constexpr XchMap make_xch_map()
{
	XchMap t{};
	bool scr[432] = {};
	uint32_t r = 0x4d4bu;                          // scramb.c:39-52
	for (int i = 0; i < 432; i++) {
		const uint32_t b = ((r >> 14) ^ r) & 1u;
		r = ((r << 1) | b) & 0xffffu;
		scr[i] = b != 0;
	}
	// gmr1_punct_k9_13_P1213 (0 = punctured), repeated by gmr1_puncturer_generate over 624 bits
	const uint8_t p[39] = {1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0, 1, 1,
	                       1, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 1};
	int q = 0;                                     // index into bits_c (the sent coded bits)
	for (int ii = 0; ii < 3 * kXchLen; ii++) {
		if (!p[ii % 39]) {
			t.m[ii] = 0x8000u;
			continue;
		}
		const int ei = 54 * ((5 * q) & 7) + (q >> 3);    // gmr1_deinterleave_intra, interleave.c:73-87
		t.m[ii] = (uint16_t)(ei | (scr[ei] ? 0x200 : 0));
		q++;
	}
	return t;
}
__constant__ XchMap c_xch_map = make_xch_map();

// ---- per position (register r, lane loc) constants of the in-place 256-state butterfly -----------------
// position bits c0..c5 = lane (xor masks 32, 16, 8, 7, 2, 1), c6 c7 = register index; the predecessor
// state held in phase ph has bit i = c[(7 - i + ph) % 8]
//   o[r][loc]   : code word (3 bits) of the own transition per phase, 3 bits each
//   st[r][loc]  : state held in phase 0;  loc_of[state] = r * 64 + loc
struct K9Tab { uint32_t o[4][64]; uint8_t st[4][64]; uint8_t loc_of[256]; };
static constexpr uint32_t parity9(uint32_t v)
{
	v ^= v >> 8; v ^= v >> 4; v ^= v >> 2; v ^= v >> 1;
	return v & 1u;
}
static constexpr uint32_t k9_out(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;             // bit i = D^i
	return (parity9(reg & 0x1edu) << 2) | (parity9(reg & 0x19bu) << 1) | parity9(reg & 0x127u);
}
static constexpr K9Tab make_k9()
{
	K9Tab t{};
	for (uint32_t r = 0; r < 4; r++) {
		for (uint32_t loc = 0; loc < 64; loc++) {
			uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			c[0] = (loc >> 5) & 1u;
			c[1] = (loc >> 4) & 1u;
			c[2] = (loc >> 3) & 1u;
			uint32_t x = loc & 7u;
			c[3] = (x >> 2) & 1u;
			x ^= c[3] ? 7u : 0u;
			c[4] = (x >> 1) & 1u;
			c[5] = x & 1u;
			c[6] = (r >> 1) & 1u;
			c[7] = r & 1u;
			uint32_t e = 0;
			for (int ph = 0; ph < 8; ph++) {
				uint32_t sp = 0;
				for (int i = 0; i < 8; i++)
					sp |= c[(7 - i + ph) % 8] << i;
				e |= k9_out(sp, sp >> 7) << (3 * ph);
				if (ph == 0) {
					t.st[r][loc] = (uint8_t)sp;
					t.loc_of[sp] = (uint8_t)(r * 64 + loc);
				}
			}
			t.o[r][loc] = e;
		}
	}
	return t;
}
__constant__ K9Tab c_k9 = make_k9();

// CRC16 syndromes of a 192-bit message followed by its 16 CRC bits (crc.c:58-63)
struct Syn208 { uint16_t s[kXchLen]; };
static constexpr Syn208 make_syn208()
{
	Syn208 t{};
	for (int k = 0; k < kXchLen; k++) {
		uint32_t v = 0;
		if (k < 192) {
			uint32_t crc = 0x8000u;
			for (int i = k; i < 192; i++)
				crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xffffu) : ((crc << 1) & 0xffffu);
			v = crc;
		} else {
			v = 1u << (15 - (k - 192));
		}
		t.s[k] = (uint16_t)v;
	}
	return t;
}
__constant__ Syn208 c_syn208 = make_syn208();

template <int CTRL>
__device__ __forceinline__ uint32_t dpp9(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
// the word of the position whose lane differs in position bit PH (PH < 6)
template <int PH>
__device__ __forceinline__ uint32_t k9_partner(uint32_t w)
{
	if constexpr (PH == 0) return (uint32_t)__shfl_xor((int)w, 32);
	else if constexpr (PH == 1) return (uint32_t)__shfl_xor((int)w, 16);
	else if constexpr (PH == 2) return dpp9<0x128>(w);      // row_ror:8
	else if constexpr (PH == 3) return dpp9<0x141>(w);      // row_half_mirror: xor 7
	else if constexpr (PH == 4) return dpp9<0x4E>(w);       // quad_perm [2,3,0,1]
	else return dpp9<0xB1>(w);                              // quad_perm [1,0,3,2]
}

typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
__device__ __forceinline__ uint32_t lds_read(uint32_t addr)
{
	return *(lds_cu32 *)(uintptr_t)addr;
}

// One trellis step at window position J (phase J % 8).  ad[ph][r]: LDS byte address of the position's
// table value in step 0 of the current window;  T[ph]: decision bit pattern of the lane phases.
// REC = false (the warm-up pass): no decision bits, one add less per state.
template <int J, bool REC>
__device__ __forceinline__ void k9_step(uint32_t (&w)[4], const uint32_t (&ad)[8][4], const uint32_t (&hi)[6])
{
	constexpr int PH = J & 7;
	uint32_t m[4];
#pragma unroll
	for (int r = 0; r < 4; r++)
		m[r] = lds_read(ad[PH][r] + 32u * J);
	if constexpr (PH < 6) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t v = REC ? w[r] + (hi[PH] << J) : w[r];
			const uint32_t t1 = v + m[r];
			const uint32_t t2 = k9_partner<PH>(v) - m[r];
			w[r] = t1 < t2 ? t1 : t2;
		}
	} else {
		constexpr int D = PH == 6 ? 2 : 1;           // register distance of the pair
#pragma unroll
		for (int q = 0; q < 2; q++) {
			const int lo = PH == 6 ? q : 2 * q, h = lo + D;
			const uint32_t vl = w[lo], vh = REC ? w[h] + (1u << J) : w[h];
			const uint32_t a1 = vl + m[lo], a2 = vh - m[lo];
			const uint32_t b1 = vh + m[h], b2 = vl - m[h];
			w[lo] = a1 < a2 ? a1 : a2;
			w[h] = b1 < b2 ? b1 : b2;
		}
	}
}

__device__ __forceinline__ uint32_t wave_min_metric(const uint32_t (&w)[4])
{
	uint32_t a = w[0] < w[1] ? w[0] : w[1], b = w[2] < w[3] ? w[2] : w[3];
	uint32_t mn = (a < b ? a : b) >> 16;
	uint32_t o;
	o = dpp9<0xB1>(mn); mn = o < mn ? o : mn;
	o = dpp9<0x4E>(mn); mn = o < mn ? o : mn;
	o = dpp9<0x141>(mn); mn = o < mn ? o : mn;
	o = dpp9<0x128>(mn); mn = o < mn ? o : mn;
	o = (uint32_t)__shfl_xor((int)mn, 16); mn = o < mn ? o : mn;
	o = (uint32_t)__shfl_xor((int)mn, 32); mn = o < mn ? o : mn;
	return mn;
}

// 208 steps = 13 windows.  REC: store each window's decisions and return what was subtracted.
template <bool REC>
__device__ __forceinline__ int k9_pass(uint32_t (&w)[4], uint32_t (&ad)[8][4], const uint32_t (&hi)[6],
                                       uint16_t *__restrict__ win, int lane)
{
	int off = 0;
#pragma unroll 1
	for (int wm = 0; wm < kXchWin; wm++) {
		k9_step<0, REC>(w, ad, hi); k9_step<1, REC>(w, ad, hi); k9_step<2, REC>(w, ad, hi); k9_step<3, REC>(w, ad, hi);
		k9_step<4, REC>(w, ad, hi); k9_step<5, REC>(w, ad, hi); k9_step<6, REC>(w, ad, hi); k9_step<7, REC>(w, ad, hi);
		k9_step<8, REC>(w, ad, hi); k9_step<9, REC>(w, ad, hi); k9_step<10, REC>(w, ad, hi); k9_step<11, REC>(w, ad, hi);
		k9_step<12, REC>(w, ad, hi); k9_step<13, REC>(w, ad, hi); k9_step<14, REC>(w, ad, hi); k9_step<15, REC>(w, ad, hi);
#pragma unroll
		for (int r = 0; r < 4; r++) {
			if (REC)
				win[(wm * 4 + r) * 64 + lane] = (uint16_t)w[r];
			w[r] &= 0xffff0000u;
		}
#pragma unroll
		for (int ph = 0; ph < 8; ph++)
#pragma unroll
			for (int r = 0; r < 4; r++)
				ad[ph][r] += 16u * 32u;
		if ((wm & 3) == 3) {
			const int d = (int)wave_min_metric(w) - (int)kBias;
#pragma unroll
			for (int r = 0; r < 4; r++)
				w[r] -= (uint32_t)d << 16;
			off += d;
		}
	}
	return off;
}

__global__ __launch_bounds__(64) void k_xch(XchArgs a)
{
	__shared__ __align__(16) int8_t s_e[432];
	__shared__ __align__(16) uint32_t s_tab[kXchLen * 8];      // (2 cost(word) - K) << 16 per step and code word
	__shared__ uint16_t s_win[kXchWin * 4 * 64];
	__shared__ uint16_t s_u[kXchWin];                          // decoded bits, LSB first
	__shared__ uint8_t s_locof[256];                           // position of a state (the survivor walk's 13 hops)
	const int lane = threadIdx.x;
	const int g = blockIdx.x;
#pragma unroll
	for (int r = 0; r < 4; r++)
		s_locof[c_k9.st[r][lane]] = (uint8_t)(r * 64 + lane);

	{
		const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)g * 432);
		uint32_t *dst = reinterpret_cast<uint32_t *>(s_e);
		dst[lane] = src[lane];
		if (lane < 44)
			dst[64 + lane] = src[64 + lane];
	}
	WSYNC();

	// ---- branch metric table; sum of K over the steps
	int ksum = 0;
	for (int k = lane; k < kXchLen; k += 64) {
		int d[3];
#pragma unroll
		for (int j = 0; j < 3; j++) {
			const uint32_t me = c_xch_map.m[3 * k + j];
			int v = (me & 0x8000u) ? 0 : (int)s_e[me & 0x1ffu];
			if (me & 0x200u)
				v = (int8_t)(-v);
			// cost of a soft bit against a coded 0 / 1: ((v -/+ 127)^2) >> 9, nothing for an erasure
			const int e0 = v - 127, e1 = v + 127;
			const int c0 = v ? (__mul24(e0, e0) >> 9) : 0, c1 = v ? (__mul24(e1, e1) >> 9) : 0;
			d[j] = c1 - c0;
			ksum += c0 + c1;
		}
#pragma unroll
		for (int o = 0; o < 8; o++) {
			const int m2 = ((o & 4) ? d[0] : -d[0]) + ((o & 2) ? d[1] : -d[1]) + ((o & 1) ? d[2] : -d[2]);
			s_tab[k * 8 + o] = (uint32_t)m2 << 16;
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		ksum += __shfl_xor(ksum, o);
	WSYNC();

	// ---- per-lane constants
	uint32_t hi[6];
	{
		const uint32_t loc = (uint32_t)lane;
		uint32_t x = loc & 7u;
		hi[0] = (loc >> 5) & 1u;
		hi[1] = (loc >> 4) & 1u;
		hi[2] = (loc >> 3) & 1u;
		hi[3] = (x >> 2) & 1u;
		x ^= hi[3] ? 7u : 0u;
		hi[4] = (x >> 1) & 1u;
		hi[5] = x & 1u;
	}
	const uint32_t tab_base = (uint32_t)(uintptr_t)(lds_cu32 *)s_tab;
	uint32_t ad[8][4];
	uint32_t w[4];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const uint32_t e = c_k9.o[r][lane];
#pragma unroll
		for (int ph = 0; ph < 8; ph++)
			ad[ph][r] = tab_base + 4u * ((e >> (3 * ph)) & 7u);
		// pass 1 starts from state 0 (libosmocore initialises every other state to MAX_AE)
		w[r] = (c_k9.st[r][lane] ? kUnreach : kBias) << 16;
	}

	// pass 1: metrics only; osmo_conv_decode_rewind subtracts the minimum
	k9_pass<false>(w, ad, hi, s_win, lane);
	{
		const int d = (int)wave_min_metric(w) - (int)kBias;
#pragma unroll
		for (int r = 0; r < 4; r++)
			w[r] -= (uint32_t)d << 16;
#pragma unroll
		for (int ph = 0; ph < 8; ph++)
#pragma unroll
			for (int r = 0; r < 4; r++)
				ad[ph][r] -= (uint32_t)kXchLen * 32u;
	}
	// pass 2
	const int off = k9_pass<true>(w, ad, hi, s_win, lane);
	WSYNC();

	// best end state: smallest metric, lowest state on ties
	unsigned long long key = ~0ull;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const unsigned long long kr = ((unsigned long long)(w[r] >> 16) << 32) | c_k9.st[r][lane];
		key = kr < key ? kr : key;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned long long ok = __shfl_xor(key, o);
		key = ok < key ? ok : key;
	}
	const uint32_t end_state = (uint32_t)key & 255u;
	// words hold 2 * ae - sum K (+ bias, - what the renormalisations took)
	const int min_ae = ((int)(uint32_t)(key >> 32) - (int)kBias + off + ksum) >> 1;

	// ---- survivor chain (uniform): window m's decisions are u[16m-8 .. 16m+7] of the path ending at the
	// position, LSB first; its low byte, bit-reversed, is the state the window started in
	{
		uint32_t P = s_locof[end_state];
		uint32_t nxt = __brev(end_state) >> 24;              // u[200..207]
		for (int wm = kXchWin - 1; wm >= 0; wm--) {
			const uint32_t h = s_win[wm * 256 + P];
			if (lane == 0)
				s_u[wm] = (uint16_t)((h >> 8) | (nxt << 8));
			nxt = h & 0xffu;
			P = s_locof[__brev(nxt) >> 24];
		}
	}
	WSYNC();

	// ---- CRC16 over u[0..191] against u[192..207]; L2 = u[0..191] LSB first (osmo_ubit2pbit_ext lsb mode)
	uint32_t syn = 0;
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const int k = 4 * lane + q;
		if (k < kXchLen && ((s_u[k >> 4] >> (k & 15)) & 1u))
			syn ^= c_syn208.s[k];
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		syn ^= (uint32_t)__shfl_xor((int)syn, o);
	if (lane < 12)
		reinterpret_cast<uint16_t *>(a.l2 + (size_t)g * 24)[lane] = s_u[lane];
	if (lane == 0) {
		a.crc[g] = syn ? 1 : 0;
		if (a.conv)
			a.conv[g] = min_ae;
	}
}

hipError_t launch_xch(const XchArgs &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_xch, dim3(a.n), dim3(64), 0, st, a);
	return hipGetLastError();
}

}  // namespace gmr1
