// nt9_kernels.hip -- layer-1 decoders of the NT9 burst (662 soft bits): FACCH9 (reference
// src/l1/facch9.c:106-144) and TCH9 in its three modes (reference src/l1/tch9.c:139-175), gfx950.
//
// All four are K=5 codes (16 states): FACCH9 rate 1/2 over 316 bits + CRC16; TCH9 9k6 rate 1/2 over 480
// bits, 4k8 rate 1/3 over 240, 2k4 rate 1/5 over 144, each punctured down to 648 coded bits
// (gmr1_puncturer_generate, punct.c:48-133).  One kernel, four bursts per wavefront, one per 16-lane DPP
// row, with the packed [metric:16 | window decisions:16] word and the in-place butterfly (masks 8, 7, 2, 1)
// of decode4_k5_12 (rx_kernels.hip).  What differs between the four is data:
//
//   * a per-kind MAP (built on the host, capi_nt9.cpp) that says, for every coded bit of every trellis
//     step, whether it was punctured and otherwise where its soft bit sits in the burst: status / SACCH
//     demux, [decipher], descramble, TCH9's depth-3 inter-burst de-interleaver (a coded bit may come from
//     this burst or one of the two before it in the channel's sequence) and the intra-burst de-interleaver
//     are all folded into that gather;
//   * the number of byte tables a candidate's cost is the sum of (coded bits 0-1, 2-3, 4).
//
// The metric of 320 ... 484 steps does not fit 16 bits, so every 64 steps the row minimum is subtracted
// (and added back into conv_rv at the end); comparisons never see the difference.
//
// Instantiated with ACC the kernels follow libosmocore's accelerated decoder (osmo_conv_decode_acc: K = 5, N <= 4, so
// not the rate-1/5 TCH9 2k4 code) instead of its generic one -- decision D1b, oracle/orc_3p_acc.c; see l1_kernels.hip.
// The cost of contradicting a soft bit is |in| (N even) or 2 |in| (N = 3: state 0's lead of 127 * N * K correlation units
// is odd there, so costs stay in correlation units), table entries are 16 bits wide (two soft bits of -128 cost 256).
#include <type_traits>

#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

// map entry: bit 31 punctured; bits 0-9 index into the burst's 662 e-bits; bit 10 scrambler flips it;
// bits 11-20 index into bits_my (cipher stream position); bits 21-22 how many bursts back it was sent
static constexpr uint32_t kMapPunct = 0x80000000u;

// scrambler bits over 648 positions (reference src/l1/scramb.c:39-52) are folded into the map on the host

template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int PH>
__device__ __forceinline__ uint32_t partner(uint32_t w)
{
	if constexpr (PH == 0) return dppu<0x128>(w);            // row_ror:8
	else if constexpr (PH == 1) return dppu<0x141>(w);       // row_half_mirror: xor 7
	else if constexpr (PH == 2) return dppu<0x4E>(w);        // quad_perm [2,3,0,1]
	else return dppu<0xB1>(w);                               // quad_perm [1,0,3,2]
}
template <int X>
__device__ __forceinline__ uint32_t row_xor_min(uint32_t v)
{
	uint32_t o;
	if constexpr (X == 8) o = dppu<0x128>(v);
	else if constexpr (X == 4) o = dppu<0x1B>(dppu<0x141>(v));
	else if constexpr (X == 2) o = dppu<0x4E>(v);
	else o = dppu<0xB1>(v);
	return o < v ? o : v;
}

// CRC16 syndromes of a 300-bit message followed by its 16 CRC bits, 20 bits per lane of a row
struct Syn316 { uint16_t s[16][20]; };
static constexpr Syn316 make_syn316()
{
	Syn316 t{};
	for (int k = 0; k < 316; k++) {
		uint32_t v = 0;
		if (k < 300) {
			uint32_t crc = 0x8000u;
			for (int i = k; i < 300; i++)
				crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xffffu) : ((crc << 1) & 0xffffu);
			v = crc;
		} else {
			v = 1u << (15 - (k - 300));
		}
		t.s[k / 20][k % 20] = (uint16_t)v;
	}
	return t;
}
__constant__ Syn316 c_syn316 = make_syn316();

// per row location: coded word (<= 5 bits) of the own / partner transition per phase, HIGH-predecessor pattern
struct K5Loc { uint32_t state[4]; uint32_t hi; };
static constexpr K5Loc k5_loc(uint32_t loc)
{
	K5Loc r{};
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
		r.state[ph] = sp;
		for (int j = ph; j < 16; j += 4)
			r.hi |= (sp >> 3) << j;
	}
	return r;
}
struct K5Locs { uint32_t st[16]; uint32_t hi[16]; };       // st: 4 states x 4 bits packed
static constexpr K5Locs make_k5locs()
{
	K5Locs t{};
	for (uint32_t loc = 0; loc < 16; loc++) {
		const K5Loc r = k5_loc(loc);
		t.st[loc] = r.state[0] | (r.state[1] << 4) | (r.state[2] << 8) | (r.state[3] << 12);
		t.hi[loc] = r.hi;
	}
	return t;
}
__constant__ K5Locs c_k5locs = make_k5locs();

// generator polynomials (bit i = D^i; reference src/l1/conv.c:123-128, 148-154, 201-209), MSB of the word = g0
__device__ __forceinline__ uint32_t coded_word(int N, uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	uint32_t o = 0;
	if (N == 2) {
		o = ((uint32_t)(__popc(reg & 0x19u) & 1) << 1) | (uint32_t)(__popc(reg & 0x17u) & 1);
	} else if (N == 4) {
		const uint32_t g4[4] = {0x19u, 0x17u, 0x15u, 0x1fu};          // conv.c:174-181
		for (int i = 0; i < 4; i++)
			o = (o << 1) | (uint32_t)(__popc(reg & g4[i]) & 1);
	} else {
		const uint32_t g[5] = {0x15u, 0x1bu, 0x1fu, 0x1du, 0x17u};
		for (int i = 0; i < N; i++)
			o = (o << 1) | (uint32_t)(__popc(reg & g[i]) & 1);
	}
	return o;
}

template <bool ACC = false, int SCALE = 1>
__device__ __forceinline__ int sbit_cost(int v, int bit)
{
	if constexpr (ACC) {
		return SCALE * (bit ? (v > 0 ? v : 0) : (v < 0 ? -v : 0));
	} else {
		const int e = bit ? v + 127 : v - 127;
		return v ? (__mul24(e, e) >> 9) : 0;      // erasures (and punctured bits) cost nothing
	}
}

static constexpr int kNt9MaxSteps = 484;
static constexpr int kNt9MaxWin = 31;
constexpr uint32_t kSent = 0xF0000000u;

template <int PH, int NW, typename CT>
__device__ __forceinline__ uint32_t nt9_step(uint32_t w, const CT *__restrict__ ct, const uint32_t (&o_own)[3],
                                             const uint32_t (&o_par)[3])
{
	const uint32_t p = partner<PH>(w);
	uint32_t c_own = ct[o_own[0]], c_par = ct[o_par[0]];
	if (NW > 1) { c_own += ct[4 + o_own[1]]; c_par += ct[4 + o_par[1]]; }
	if (NW > 2) { c_own += ct[8 + o_own[2]]; c_par += ct[8 + o_par[2]]; }
	const uint32_t t1 = (c_own << 16) + w;
	const uint32_t t2 = (c_par << 16) + p;
	return t1 < t2 ? t1 : t2;
}

template <int NW, bool ACC>
__global__ __launch_bounds__(64) void k_nt9(Nt9Args a)
{
	typedef typename std::conditional<ACC, uint16_t, uint8_t>::type CT;
	constexpr int SC = NW == 2 ? 2 : 1;              // ACC cost unit (see the head of the file)
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	const int S = a.len + 4;                         // trellis steps
	const int N = a.N;
	const int stride = 4 * NW;                       // entries of cost tables per step
	CT *ct_all = reinterpret_cast<CT *>(lds_raw);                                  // 4 x S x stride
	uint16_t *win = reinterpret_cast<uint16_t *>(lds_raw + (size_t)4 * kNt9MaxSteps * stride * sizeof(CT));   // kNt9MaxWin x 64
	uint32_t *ub = reinterpret_cast<uint32_t *>(win + kNt9MaxWin * 64);            // 4 x 16 words
	const int g0 = blockIdx.x * 4;

	// ---- cost tables of the four bursts: word A = coded bits (0, 1), B = (2, 3), C = (4)
	for (int it = lane; it < 4 * S; it += 64) {
		const int q = it / S, k = it % S;
		const int g = g0 + q;
		uint32_t c0[5] = {0, 0, 0, 0, 0}, c1[5] = {0, 0, 0, 0, 0};
		if (g < a.n) {
			// position of the burst in its channel's sequence (runs of unequal length: given per burst)
			const int pos = a.seq_pos ? a.seq_pos[g] : g % a.seq_len;
			for (int j = 0; j < N; j++) {
				const uint32_t m = a.map[k * N + j];
				if (m & kMapPunct)
					continue;
				const int back = (int)((m >> 21) & 3u);
				if (back > pos)
					continue;                            // before the first burst of the channel: zeros
				const size_t src = (size_t)(g - back);
				int v = a.ebits[src * 662 + (m & 0x3ffu)];
				bool flip = ((m >> 10) & 1u) != 0;
				if (a.ciph)
					flip ^= a.ciph[src * 658 + ((m >> 11) & 0x3ffu)] != 0;
				if (flip)
					v = (int8_t)(-v);
				c0[j] = (uint32_t)sbit_cost<ACC, SC>(v, 0);
				c1[j] = (uint32_t)sbit_cost<ACC, SC>(v, 1);
			}
		}
		// entries: costs of the coded bit pairs (0, 1), (2, 3) and of coded bit 4, `SH` bits wide
		constexpr int SH = 8 * (int)sizeof(CT);
		uint32_t *dst = reinterpret_cast<uint32_t *>(ct_all + ((size_t)q * S + k) * stride);
		const uint32_t a00 = c0[0] + c0[1], a01 = c0[0] + c1[1], a10 = c1[0] + c0[1], a11 = c1[0] + c1[1];
		if constexpr (ACC) {
			dst[0] = a00 | (a01 << SH); dst[1] = a10 | (a11 << SH);
			if (NW == 2) { dst[2] = c0[2] | (c1[2] << SH); dst[3] = 0; }           // N = 3 (N = 5 never runs with ACC)
		} else {
			dst[0] = a00 | (a01 << 8) | (a10 << 16) | (a11 << 24);
			if (NW == 2) {
				dst[1] = c0[2] | (c1[2] << 8);
			} else if (NW == 3) {
				dst[1] = (c0[2] + c0[3]) | ((c0[2] + c1[3]) << 8) | ((c1[2] + c0[3]) << 16) | ((c1[2] + c1[3]) << 24);
				dst[2] = c0[4] | (c1[4] << 8);
			}
		}
	}
	WSYNC();

	// ---- per-lane constants: table offsets of the own / partner coded words per phase
	const uint32_t stp = c_k5locs.st[loc], hit = c_k5locs.hi[loc];
	uint32_t o_own[4][3], o_par[4][3];
	bool hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		const uint32_t sp = (stp >> (4 * ph)) & 15u;
		const uint32_t b = sp >> 3;
		hi[ph] = b != 0;
		const uint32_t wo = coded_word(N, sp, b), wp = coded_word(N, sp ^ 8u, b);
		if (N == 2) {
			o_own[ph][0] = wo; o_par[ph][0] = wp;
			o_own[ph][1] = o_own[ph][2] = o_par[ph][1] = o_par[ph][2] = 0;
		} else if (N == 3) {
			o_own[ph][0] = wo >> 1; o_own[ph][1] = wo & 1u; o_own[ph][2] = 0;
			o_par[ph][0] = wp >> 1; o_par[ph][1] = wp & 1u; o_par[ph][2] = 0;
		} else {
			o_own[ph][0] = wo >> 3; o_own[ph][1] = (wo >> 1) & 3u; o_own[ph][2] = wo & 1u;
			o_par[ph][0] = wp >> 3; o_par[ph][1] = (wp >> 1) & 3u; o_par[ph][2] = wp & 1u;
		}
	}
	uint32_t T[16];
#pragma unroll
	for (int j = 0; j < 16; j++)
		T[j] = hit & (1u << j);

	// ---- forward pass: 4 steps (decisions u[-4..-1]), full windows of 16, a last window of `tail`
	// steps (16 or 12) whose final four are the flush
	const CT *ct = ct_all + (size_t)row * S * stride;
	const int n_win = (a.len + 15) / 16;             // windows after the first four steps
	const int tail = a.len - 16 * (n_win - 1);       // 16 or 12
	uint32_t off = 0;                                // what has been subtracted from the metrics so far
	// every other start state: unreachable / (ACC) behind state 0 by 127 * N * K correlation units = SC / 2 cost units each
	constexpr uint32_t kOther = ACC ? (uint32_t)(127 * (NW == 2 ? 3 : 2) * 5 * SC / 2) << 16 : kSent;
	uint32_t w = (loc ? kOther : 0u) | T[0];
	w = nt9_step<0, NW>(w, ct + 0 * stride, o_own[0], o_par[0]) + T[1];
	w = nt9_step<1, NW>(w, ct + 1 * stride, o_own[1], o_par[1]) + T[2];
	w = nt9_step<2, NW>(w, ct + 2 * stride, o_own[2], o_par[2]) + T[3];
	w = nt9_step<3, NW>(w, ct + 3 * stride, o_own[3], o_par[3]);
	w = (w & 0xffff0000u) | T[0];
#pragma unroll 1
	for (int wm = 0; wm < n_win - 1; wm++) {
		const CT *c = ct + (size_t)(4 + 16 * wm) * stride;
#pragma unroll
		for (int j = 0; j < 16; j += 4) {
			w = nt9_step<0, NW>(w, c + (j + 0) * stride, o_own[0], o_par[0]) + T[(j + 1) & 15];
			w = nt9_step<1, NW>(w, c + (j + 1) * stride, o_own[1], o_par[1]) + T[(j + 2) & 15];
			w = nt9_step<2, NW>(w, c + (j + 2) * stride, o_own[2], o_par[2]) + T[(j + 3) & 15];
			w = nt9_step<3, NW>(w, c + (j + 3) * stride, o_own[3], o_par[3]) + (j + 4 < 16 ? T[(j + 4) & 15] : 0u);
		}
		win[wm * 64 + lane] = (uint16_t)w;
		w = (w & 0xffff0000u) | T[0];
		if ((wm & 3) == 3) {
			// keep the metrics inside 16 bits: subtract the row minimum
			uint32_t mn = w >> 16;
			mn = row_xor_min<1>(mn);
			mn = row_xor_min<2>(mn);
			mn = row_xor_min<4>(mn);
			mn = row_xor_min<8>(mn);
			w -= mn << 16;
			off += mn;
		}
	}
	{
		const CT *c = ct + (size_t)(4 + 16 * (n_win - 1)) * stride;
		const int body = tail - 4;                   // steps of the last window before the flush (12 or 8)
#pragma unroll
		for (int j = 0; j < 8; j += 4) {
			w = nt9_step<0, NW>(w, c + (j + 0) * stride, o_own[0], o_par[0]) + T[j + 1];
			w = nt9_step<1, NW>(w, c + (j + 1) * stride, o_own[1], o_par[1]) + T[j + 2];
			w = nt9_step<2, NW>(w, c + (j + 2) * stride, o_own[2], o_par[2]) + T[j + 3];
			w = nt9_step<3, NW>(w, c + (j + 3) * stride, o_own[3], o_par[3]) + T[j + 4];
		}
		if (body == 12) {
			w = nt9_step<0, NW>(w, c + 8 * stride, o_own[0], o_par[0]) + T[9];
			w = nt9_step<1, NW>(w, c + 9 * stride, o_own[1], o_par[1]) + T[10];
			w = nt9_step<2, NW>(w, c + 10 * stride, o_own[2], o_par[2]) + T[11];
			w = nt9_step<3, NW>(w, c + 11 * stride, o_own[3], o_par[3]) + T[12];
		}
		// flush: only b = 0 transitions survive (lanes whose new state ends in 1 become unreachable)
		const uint32_t t1 = body == 12 ? T[13] : T[9], t2 = body == 12 ? T[14] : T[10], t3 = body == 12 ? T[15] : T[11];
		// (ACC: ordinary butterflies)
		w = nt9_step<0, NW>(w, c + (body + 0) * stride, o_own[0], o_par[0]);
		w = !ACC && hi[0] ? kSent : (w + t1);
		w = nt9_step<1, NW>(w, c + (body + 1) * stride, o_own[1], o_par[1]);
		w = !ACC && hi[1] ? kSent : (w + t2);
		w = nt9_step<2, NW>(w, c + (body + 2) * stride, o_own[2], o_par[2]);
		w = !ACC && hi[2] ? kSent : (w + t3);
		w = nt9_step<3, NW>(w, c + (body + 3) * stride, o_own[3], o_par[3]);
		w = !ACC && hi[3] ? kSent : w;
		win[(n_win - 1) * 64 + lane] = (uint16_t)w;
	}
	const uint32_t final_ae = ACC ? 0u : (w >> 16) + off;       // state 0 ends in location 0 of the row; osmo_conv_decode_acc returns 0
	WSYNC();

	// ---- survivor chain, one lane per row: window m's decisions at the survivor's location are decoded
	// bits u[16m ..]; their low nibble names the state at the window start (bit-reversed)
	if (loc == 0) {
		constexpr unsigned long long kLocOf =
			0x0ull | (0x8ull << 4) | (0x7ull << 8) | (0xFull << 12) | (0x2ull << 16) | (0xAull << 20) |
			(0x5ull << 24) | (0xDull << 28) | (0x1ull << 32) | (0x9ull << 36) | (0x6ull << 40) |
			(0xEull << 44) | (0x3ull << 48) | (0xBull << 52) | (0x4ull << 56) | (0xCull << 60);
		const uint16_t *d16 = win + row * 16;
		uint32_t L = 0, prev = 0;
		for (int wm = n_win - 1; wm >= 0; wm--) {
			uint32_t h = d16[wm * 64 + L];
			L = (uint32_t)(kLocOf >> (4 * (h & 15u))) & 15u;
			if (wm == n_win - 1 && tail == 12)
				h &= 0xfffu;
			if (wm & 1) {
				prev = h;
			} else {
				ub[row * 16 + (wm >> 1)] = h | (prev << 16);
				prev = 0;
			}
		}
	}
	WSYNC();

	// ---- outputs
	const int g = g0 + row;
	if (g >= a.n)
		return;
	const uint32_t *u = ub + row * 16;
	// L2 bytes: decoded bits LSB first (osmo_ubit2pbit_ext lsb mode) = the words as they stand
	for (int i = (int)loc; i < a.l2_bytes; i += 16) {
		uint32_t byte = (u[i >> 2] >> (8 * (i & 3))) & 0xffu;
		const int bits_left = a.len - (a.kind == 3 ? 16 : 0) - 8 * i;   // FACCH9: 300 message bits
		if (bits_left < 8)
			byte &= (1u << (bits_left > 0 ? bits_left : 0)) - 1u;
		a.l2[(size_t)g * a.l2_bytes + i] = (uint8_t)byte;
	}
	// status = e[52..55]; SACCH = bits_my[52..61] = e[56..65] after deciphering (facch9.c:118-131, tch9.c:152-165)
	if (a.status && loc < 4)
		a.status[(size_t)g * 4 + loc] = a.ebits[(size_t)g * 662 + 52 + loc];
	if (a.sacch && loc < 10) {
		int v = a.ebits[(size_t)g * 662 + 56 + loc];
		if (a.ciph && a.ciph[(size_t)g * 658 + 52 + loc])
			v = (int8_t)(-v);
		a.sacch[(size_t)g * 10 + loc] = (int8_t)v;
	}
	if (a.kind == 3 && a.crc) {
		// CRC16 over 300 + 16 bits: 20 bits per lane, XOR-reduced over the row
		uint32_t syn = 0;
		for (int q = 0; q < 20; q++) {
			const int k = (int)loc * 20 + q;
			if (k < 316 && ((u[k >> 5] >> (k & 31)) & 1u))
				syn ^= c_syn316.s[loc][q];
		}
		syn ^= dppu<0xB1>(syn);
		syn ^= dppu<0x4E>(syn);
		syn ^= dppu<0x1B>(dppu<0x141>(syn));
		syn ^= dppu<0x128>(syn);
		if (loc == 0)
			a.crc[g] = syn ? 1 : 0;
	}
	if (a.conv && loc == 0)
		a.conv[g] = (int32_t)final_ae;
}

// ---------------------------------------------------------------------------
// RACH (reference src/l1/rach.c:127-200): 494 soft bits -> 18 bytes.  K = 5 rate 1/4 over 159 bits
// (123 class-2 bits + CRC12, then 16 class-1 bits + CRC8 xor SB mask), bits 2 and 3 of the first 135
// steps punctured (rach.c:53-66); the class-1 part of the burst is sent twice and the two copies are
// averaged after descrambling (rach.c:161-162).  Same packed-word trellis, four bursts per wavefront.
// ---------------------------------------------------------------------------
static constexpr int kRachLen = 159, kRachSteps = 163, kRachWin = 10;     // last window: 15 steps

// map entry: bit 31 punctured; bits 0-9 index into the 494 e-bits, bit 10 the scrambler flips it;
// bit 22: a second copy exists at bits 11-20 (flip: bit 21)
struct RachMap { uint32_t m[4 * kRachSteps]; };
static constexpr uint32_t rach_x2e(int i)      // position in the de-multiplexed burst x -> e (rach.c:150-154)
{
	return (uint32_t)(i < 112 ? 136 + i : (i < 248 ? i - 112 : (i < 382 ? 360 + (i - 248) : 248 + (i - 382))));
}
static constexpr RachMap make_rach_map()
{
	RachMap t{};
	bool scr[494] = {};
	uint32_t r = 0x4d4bu;
	for (int i = 0; i < 494; i++) {
		const uint32_t b = ((r >> 14) ^ r) & 1u;
		r = ((r << 1) | b) & 0xffffu;
		scr[i] = b != 0;
	}
	for (int k = 0; k < kRachSteps; k++) {
		for (int j = 0; j < 4; j++) {
			uint32_t e = 0;
			if (k < 135 && j >= 2) {
				e = kMapPunct;
			} else if (k < 135) {
				const int q = 2 * k + j;                                   // bits_c[0..269] = e2p (N = 33, last 6 as sent)
				const int i2 = q < 264 ? 33 * ((5 * q) & 7) + (q >> 3) : q;
				const int xi = 112 + i2;
				e = rach_x2e(xi) | (scr[xi] ? 0x400u : 0u);
			} else {
				const int qq = 4 * (k - 135) + j;                          // bits_c[270..381] = e1p (N = 14)
				const int i1 = 14 * ((5 * qq) & 7) + (qq >> 3);
				const int xa = i1, xb = i1 + 382;
				e = rach_x2e(xa) | (scr[xa] ? 0x400u : 0u) | (rach_x2e(xb) << 11) | (scr[xb] ? 0x200000u : 0u) |
				    0x400000u;
			}
			t.m[4 * k + j] = e;
		}
	}
	return t;
}
__constant__ RachMap c_rach_map = make_rach_map();

// syndromes: bits 0-11 CRC12 over u[0..122] + u[123..134], bits 16-23 CRC8 over u[135..150] + u[151..158]
// (crc.c:36-54), 10 bits per lane of a row
struct SynRach { uint32_t s[16][10]; };
static constexpr SynRach make_syn_rach()
{
	SynRach t{};
	for (int k = 0; k < kRachLen; k++) {
		uint32_t v = 0;
		if (k < 123) {
			uint32_t crc = 0x800u;
			for (int i = k; i < 123; i++)
				crc = (crc & 0x800u) ? (((crc << 1) ^ 0x80fu) & 0xfffu) : ((crc << 1) & 0xfffu);
			v = crc;
		} else if (k < 135) {
			v = 1u << (11 - (k - 123));
		} else if (k < 151) {
			uint32_t crc = 0x80u;
			for (int i = k - 135; i < 16; i++)
				crc = (crc & 0x80u) ? (((crc << 1) ^ 0x9bu) & 0xffu) : ((crc << 1) & 0xffu);
			v = crc << 16;
		} else {
			v = (1u << (7 - (k - 151))) << 16;
		}
		t.s[k / 10][k % 10] = v;
	}
	return t;
}
__constant__ SynRach c_syn_rach = make_syn_rach();

template <bool ACC>
__global__ __launch_bounds__(64) void k_rach(RachArgs a)
{
	typedef typename std::conditional<ACC, uint16_t, uint8_t>::type CT;
	__shared__ __align__(16) CT s_ct[4 * kRachSteps * 8];
	__shared__ uint16_t s_win[kRachWin * 64];
	__shared__ uint32_t s_ub[4 * 8];
	const int lane = threadIdx.x;
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	const int g0 = blockIdx.x * 4;
	constexpr int stride = 8;

	if (lane < 32)
		s_ub[lane] = 0;
	// ---- cost tables: word A = coded bits (0, 1), word B = (2, 3)
	for (int it = lane; it < 4 * kRachSteps; it += 64) {
		const int q = it / kRachSteps, k = it % kRachSteps;
		const int g = g0 + q;
		uint32_t c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0};
		if (g < a.n) {
			const int8_t *e = a.ebits + (size_t)g * 494;
			for (int j = 0; j < 4; j++) {
				const uint32_t m = c_rach_map.m[4 * k + j];
				if (m & kMapPunct)
					continue;
				int v = e[m & 0x3ffu];
				if (m & 0x400u)
					v = (int8_t)(-v);
				if (m & 0x400000u) {
					int v2 = e[(m >> 11) & 0x3ffu];
					if (m & 0x200000u)
						v2 = (int8_t)(-v2);
					v = (int8_t)((v + v2) >> 1);
				}
				c0[j] = (uint32_t)sbit_cost<ACC>(v, 0);
				c1[j] = (uint32_t)sbit_cost<ACC>(v, 1);
			}
		}
		uint32_t *dst = reinterpret_cast<uint32_t *>(s_ct + ((size_t)q * kRachSteps + k) * stride);
		if constexpr (ACC) {
			dst[0] = (c0[0] + c0[1]) | ((c0[0] + c1[1]) << 16); dst[1] = (c1[0] + c0[1]) | ((c1[0] + c1[1]) << 16);
			dst[2] = (c0[2] + c0[3]) | ((c0[2] + c1[3]) << 16); dst[3] = (c1[2] + c0[3]) | ((c1[2] + c1[3]) << 16);
		} else {
			dst[0] = (c0[0] + c0[1]) | ((c0[0] + c1[1]) << 8) | ((c1[0] + c0[1]) << 16) | ((c1[0] + c1[1]) << 24);
			dst[1] = (c0[2] + c0[3]) | ((c0[2] + c1[3]) << 8) | ((c1[2] + c0[3]) << 16) | ((c1[2] + c1[3]) << 24);
		}
	}
	WSYNC();

	const uint32_t stp = c_k5locs.st[loc], hit = c_k5locs.hi[loc];
	uint32_t o_own[4][3], o_par[4][3];
	bool hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		const uint32_t sp = (stp >> (4 * ph)) & 15u;
		const uint32_t b = sp >> 3;
		hi[ph] = b != 0;
		const uint32_t wo = coded_word(4, sp, b), wp = coded_word(4, sp ^ 8u, b);
		o_own[ph][0] = wo >> 2; o_own[ph][1] = wo & 3u; o_own[ph][2] = 0;
		o_par[ph][0] = wp >> 2; o_par[ph][1] = wp & 3u; o_par[ph][2] = 0;
	}
	uint32_t T[16];
#pragma unroll
	for (int j = 0; j < 16; j++)
		T[j] = hit & (1u << j);

	// ---- forward pass: 4 steps (decisions u[-4..-1]), nine windows of 16, a last one of 15 steps whose
	// final four are the flush.  Max metric 163 * 4 * 126 does not fit 16 bits: renormalise every 4 windows.
	const CT *ct = s_ct + (size_t)row * kRachSteps * stride;
	uint32_t off = 0;
	constexpr uint32_t kOther = ACC ? (127u * 4u * 5u / 2u) << 16 : kSent;     // see k_nt9
	uint32_t w = (loc ? kOther : 0u) | T[0];
	w = nt9_step<0, 2>(w, ct + 0 * stride, o_own[0], o_par[0]) + T[1];
	w = nt9_step<1, 2>(w, ct + 1 * stride, o_own[1], o_par[1]) + T[2];
	w = nt9_step<2, 2>(w, ct + 2 * stride, o_own[2], o_par[2]) + T[3];
	w = nt9_step<3, 2>(w, ct + 3 * stride, o_own[3], o_par[3]);
	w = (w & 0xffff0000u) | T[0];
#pragma unroll 1
	for (int wm = 0; wm < kRachWin - 1; wm++) {
		const CT *c = ct + (size_t)(4 + 16 * wm) * stride;
#pragma unroll
		for (int j = 0; j < 16; j += 4) {
			w = nt9_step<0, 2>(w, c + (j + 0) * stride, o_own[0], o_par[0]) + T[(j + 1) & 15];
			w = nt9_step<1, 2>(w, c + (j + 1) * stride, o_own[1], o_par[1]) + T[(j + 2) & 15];
			w = nt9_step<2, 2>(w, c + (j + 2) * stride, o_own[2], o_par[2]) + T[(j + 3) & 15];
			w = nt9_step<3, 2>(w, c + (j + 3) * stride, o_own[3], o_par[3]) + (j + 4 < 16 ? T[(j + 4) & 15] : 0u);
		}
		s_win[wm * 64 + lane] = (uint16_t)w;
		w = (w & 0xffff0000u) | T[0];
		if ((wm & 3) == 3) {
			uint32_t mn = w >> 16;
			mn = row_xor_min<1>(mn);
			mn = row_xor_min<2>(mn);
			mn = row_xor_min<4>(mn);
			mn = row_xor_min<8>(mn);
			w -= mn << 16;
			off += mn;
		}
	}
	{
		const CT *c = ct + (size_t)(4 + 16 * (kRachWin - 1)) * stride;
		// 11 data steps ...
#pragma unroll
		for (int j = 0; j < 8; j += 4) {
			w = nt9_step<0, 2>(w, c + (j + 0) * stride, o_own[0], o_par[0]) + T[j + 1];
			w = nt9_step<1, 2>(w, c + (j + 1) * stride, o_own[1], o_par[1]) + T[j + 2];
			w = nt9_step<2, 2>(w, c + (j + 2) * stride, o_own[2], o_par[2]) + T[j + 3];
			w = nt9_step<3, 2>(w, c + (j + 3) * stride, o_own[3], o_par[3]) + T[j + 4];
		}
		w = nt9_step<0, 2>(w, c + 8 * stride, o_own[0], o_par[0]) + T[9];
		w = nt9_step<1, 2>(w, c + 9 * stride, o_own[1], o_par[1]) + T[10];
		w = nt9_step<2, 2>(w, c + 10 * stride, o_own[2], o_par[2]) + T[11];
		// ... and four flush steps (phases 3, 0, 1, 2): only b = 0 transitions survive
		w = nt9_step<3, 2>(w, c + 11 * stride, o_own[3], o_par[3]);
		w = !ACC && hi[3] ? kSent : (w + T[12]);
		w = nt9_step<0, 2>(w, c + 12 * stride, o_own[0], o_par[0]);
		w = !ACC && hi[0] ? kSent : (w + T[13]);
		w = nt9_step<1, 2>(w, c + 13 * stride, o_own[1], o_par[1]);
		w = !ACC && hi[1] ? kSent : (w + T[14]);
		w = nt9_step<2, 2>(w, c + 14 * stride, o_own[2], o_par[2]);
		w = !ACC && hi[2] ? kSent : w;
		s_win[(kRachWin - 1) * 64 + lane] = (uint16_t)w;
	}
	// 163 = 3 mod 4 steps: the layout is that of phase 3, where location 0 still holds state 0
	const uint32_t final_ae = ACC ? 0u : (w >> 16) + off;
	WSYNC();

	// ---- survivor chain, one lane per row.  The last window ends in the phase-3 layout, but the walk
	// starts from location 0 (state 0) and every window START is in the phase-0 layout.
	if (loc == 0) {
		constexpr unsigned long long kLocOf =
			0x0ull | (0x8ull << 4) | (0x7ull << 8) | (0xFull << 12) | (0x2ull << 16) | (0xAull << 20) |
			(0x5ull << 24) | (0xDull << 28) | (0x1ull << 32) | (0x9ull << 36) | (0x6ull << 40) |
			(0xEull << 44) | (0x3ull << 48) | (0xBull << 52) | (0x4ull << 56) | (0xCull << 60);
		const uint16_t *d16 = s_win + row * 16;
		uint32_t L = 0, prev = 0;
		for (int wm = kRachWin - 1; wm >= 0; wm--) {
			const uint32_t h = d16[wm * 64 + L];
			L = (uint32_t)(kLocOf >> (4 * (h & 15u))) & 15u;
			if (wm & 1) {
				prev = h;
			} else {
				s_ub[row * 8 + (wm >> 1)] = h | (prev << 16);
				prev = 0;
			}
		}
	}
	WSYNC();

	const int g = g0 + row;
	if (g >= a.n)
		return;
	const uint32_t *u = s_ub + row * 8;
	uint32_t syn = 0;
	for (int q = 0; q < 10; q++) {
		const int k = (int)loc * 10 + q;
		if (k < kRachLen && ((u[k >> 5] >> (k & 31)) & 1u))
			syn ^= c_syn_rach.s[loc][q];
	}
	syn ^= dppu<0xB1>(syn);
	syn ^= dppu<0x4E>(syn);
	syn ^= dppu<0x1B>(dppu<0x141>(syn));
	syn ^= dppu<0x128>(syn);
	// rach bits 0..15 = u[135..150], bits 16..138 = u[0..122] (rach.c:194-197), LSB first
	for (int i = (int)loc; i < 18; i += 16) {
		const int pos = i < 2 ? 135 + 8 * i : 8 * (i - 2);
		const unsigned long long two = (unsigned long long)u[pos >> 5] | ((unsigned long long)u[(pos >> 5) + 1] << 32);
		uint32_t byte = (uint32_t)(two >> (pos & 31)) & 0xffu;
		if (i == 17)
			byte &= 0x07u;
		a.rach[(size_t)g * 18 + i] = (uint8_t)byte;
	}
	if (loc == 0) {
		const uint32_t s12 = syn & 0xfffu, s8 = (syn >> 16) & 0xffu;
		// CRC8 is first checked as received, then with the SB mask removed (rach.c:176-184)
		int crc0 = s8 != 0;
		if (crc0)
			crc0 = (s8 ^ a.sb_mask[g]) != 0;
		const int crc1 = s12 != 0;
		a.rv[g] = crc0 || crc1;
		if (a.crc) {
			a.crc[2 * g] = crc0;
			a.crc[2 * g + 1] = crc1;
		}
		if (a.conv)
			a.conv[g] = (int32_t)final_ae;
	}
}

hipError_t launch_rach(const RachArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.conv_acc)
		hipLaunchKernelGGL(k_rach<true>, dim3((a.n + 3) / 4), dim3(64), 0, stream, a);
	else
		hipLaunchKernelGGL(k_rach<false>, dim3((a.n + 3) / 4), dim3(64), 0, stream, a);
	return hipGetLastError();
}

hipError_t launch_nt9(const Nt9Args &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	const int nw = a.N == 2 ? 1 : (a.N == 3 ? 2 : 3);
	const bool acc = a.conv_acc && a.N <= 4;         // osmo_conv_decode sends N = 5 to the generic decoder
	const size_t lds = (size_t)4 * kNt9MaxSteps * 4 * nw * (acc ? 2 : 1) + (size_t)kNt9MaxWin * 64 * 2 + 4 * 16 * 4;
	const dim3 grid((a.n + 3) / 4);
	if (nw == 1 && acc)
		hipLaunchKernelGGL((k_nt9<1, true>), grid, dim3(64), lds, stream, a);
	else if (nw == 1)
		hipLaunchKernelGGL((k_nt9<1, false>), grid, dim3(64), lds, stream, a);
	else if (nw == 2 && acc)
		hipLaunchKernelGGL((k_nt9<2, true>), grid, dim3(64), lds, stream, a);
	else if (nw == 2)
		hipLaunchKernelGGL((k_nt9<2, false>), grid, dim3(64), lds, stream, a);
	else
		hipLaunchKernelGGL((k_nt9<3, false>), grid, dim3(64), lds, stream, a);
	return hipGetLastError();
}

}  // namespace gmr1
