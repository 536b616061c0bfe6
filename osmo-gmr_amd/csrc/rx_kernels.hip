// rx_kernels.hip -- normal-burst receive kernels for gfx950 (MI355X).
//
//   k_rx<.., DECODE=false> : pi/4-CxPSK burst demodulation, one burst per wavefront
//                            (reference src/sdr/pi4cxpsk.c:520-602 gmr1_pi4cxpsk_demod)
//   k_rx<.., DECODE=true>  : the same, four bursts per wavefront back to back, followed
//                            by the BCCH/CCCH layer-1 chain for the four bursts at once:
//                            descramble + de-interleave folded into the branch-metric
//                            gather, 16-state K=5 rate-1/2 Viterbi with one burst per
//                            16-lane DPP row, survivor walk, CRC16, LSB-first packing
//                            (reference src/l1/bcch.c:83-103, src/l1/ccch.c:87-107 and
//                            libosmocore's generic osmo_conv_decode).
//   k_rx4                  : the default fused BCCH / CCCH kernel: the same arithmetic with the
//                            serial phases (timing bisection, sync-symbol terms) done once for
//                            the four bursts of a wave, one burst per 16-lane row.
//   k_rx_chain_pipe, k_rx_chain, k_rx_merge, k_rx_pack (rx_loop_kernels.inc, included below): the frame loop of gmr1_rx
//                            (process_bcch, src/gmr1_rx.c:852-895): the chain kernel walks the BCCH feedback chain of one
//                            chain per work-group -- rx4_body in its latency shape, cut into pipeline stages (PART) that
//                            run rounds apart on the work-group's waves -- and lists the CCCH bursts, k_rx4 takes those
//                            as one batch, k_rx_merge writes the records.
//   k_rx4g, k_rx4g_tch3    : demodulation only, four bursts per wave; with the TCH3 decoder behind it (tch3_body.h).
//   k_detect, k_mod_order  : gmr1_pi4cxpsk_detect / _mod_order (pi4cxpsk.c:617-729).
//   k_l1                   : the layer-1 chain alone on soft bits read from HBM.
//
// Design notes (DESIGN.md has the long form):
//   * Work-groups are single 64-lane wavefronts: every hand-off goes through the
//     wave's own LDS slice and needs no s_barrier (the loop's chain kernels: several waves, one barrier per round).
//   * Samples are loaded once from HBM with coalesced 8-byte-per-lane loads, DC /
//     power normalised in registers and parked in LDS; everything else reads LDS.
//   * Burst formats live in __constant__ memory and are read with scalar loads.
//   * The sync search never derotates the window: |sum conj(ref_n) x[.] e^{j th n}|
//     is evaluated with per-burst rotated coefficients, which drops ~1000
//     sincos per burst.  Only the 234 decimated symbols are derotated.
//   * The sinc interpolation of the early/late timing loop needs one sine per
//     point: sin(pi (k - f)) = -(-1)^k sin(pi f) for integer tap offsets k.
//   * The Viterbi butterfly is in place: the two predecessors of a state sit in two lanes
//     of the row that differ by an xor mask (8, 7, 2, 1 over the four phases: one DPP control
//     each), so the partner metric arrives folded into the add and no metric ever moves.  A
//     state's 32-bit word is [metric:16 | decisions of the current 16-step window:16]; v_min_u32
//     does compare, select, tie-break and decision recording at once (see decode4_k5_12).
#include <mutex>
#include <type_traits>

#include "gmr1_dev.h"
#include "tch3_body.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr float kPif = 3.14159265358979323846f;
typedef float v2f __attribute__((ext_vector_type(2)));   // (re, im) in a register pair: v_pk_add / v_pk_mul / v_pk_fma_f32
static constexpr int kSteps12 = 212;              // 208 data + 4 flush steps (BCCH/CCCH)
static constexpr int kEbRow = 448;                // LDS bytes per soft-bit row (>= 432, /16)
static constexpr int kEbitsLds = 704;             // single-burst soft-bit buffer (>= 662)

// ---------------------------------------------------------------------------
// constant memory
// ---------------------------------------------------------------------------
__constant__ DevBurst c_types[kNumTypes];


// Per trellis step of the BCCH / CCCH chain: where the two soft bits of the step
// sit in the burst's e-bit order and whether the scrambler flips them
//   bits  0..9  index of c[2k]   bit 10 its scrambling bit
//   bits 16..25 index of c[2k+1] bit 26 its scrambling bit
// (interleave.c:73-87 with N=53, scramb.c:39-73; CCCH: 4 leading pad bits, ccch.c:95-96)
struct StepTable { uint32_t w[2][kSteps12]; };
static constexpr StepTable make_steps()
{
	StepTable t{};
	for (int chain = 0; chain < 2; chain++) {
		const int off = chain ? 4 : 0;
		// scrambling sequence over the e-bit positions
		bool scr[448] = {};
		uint16_t r = 0x4d4b;
		for (int i = 0; i < 448; i++) {
			uint32_t b = ((r >> 14) ^ r) & 1u;
			r = (uint16_t)((r << 1) | b);
			scr[i] = b != 0;
		}
		for (int k = 0; k < kSteps12; k++) {
			uint32_t w = 0;
			for (int j = 0; j < 2; j++) {
				const int kc = 2 * k + j;
				const int ei = 53 * ((5 * kc) & 7) + (kc >> 3) + off;
				w |= ((uint32_t)ei | (scr[ei] ? 0x400u : 0u)) << (16 * j);
			}
			t.w[chain][k] = w;
		}
	}
	return t;
}
__constant__ StepTable c_steps = make_steps();

// Viterbi input cost of one soft bit, libosmocore's generic decoder: ((in -+ 127)^2) >> 9, and 0 for
// an erasure (in == 0).  Index = soft bit as uint8, + 256 when the scrambler flips it (the flipped
// value is (int8)(-v), so -128 stays -128 exactly as in gmr1_scramble_sbit, scramb.c:63-73).
// a[]: first coded bit of a step, cost replicated to the bytes of the words ov = 0..3 it belongs
// to (byte ov holds c0 for ov < 2, c1 otherwise); b[]: second coded bit (c0 for even ov, c1 for odd).
struct CostTable { uint32_t a[512], b[512]; };
static constexpr CostTable make_cost()
{
	CostTable t{};
	for (int idx = 0; idx < 512; idx++) {
		int v = (int)(int8_t)(uint8_t)(idx & 255);
		if (idx & 256)
			v = (int)(int8_t)(uint8_t)(-v);
		const int e0 = v - 127, e1 = v + 127;
		const uint32_t c0 = v ? (uint32_t)((e0 * e0) >> 9) : 0u;
		const uint32_t c1 = v ? (uint32_t)((e1 * e1) >> 9) : 0u;
		t.a[idx] = c0 | (c0 << 8) | (c1 << 16) | (c1 << 24);
		t.b[idx] = c0 | (c1 << 8) | (c0 << 16) | (c1 << 24);
	}
	return t;
}
__constant__ CostTable c_cost = make_cost();

// The same for libosmocore's accelerated decoder (osmo_conv_decode_acc, decision D1b: oracle/orc_3p_acc.c).  It MAXIMISES
// the correlation sum in * (+-1); minimising  sum over the coded bits that contradict the soft bit's sign of |in|  ranks
// every pair of paths identically ((sum |in| - correlation) / 2, an integer) and is non-negative, so the packed
// [metric | decisions] words and v_min_u32 serve both decoders.  |in| <= 127 on the fused path (the demodulator's soft
// bits); two soft bits of -128 in one step would overflow a byte lane -- k_l1 takes 16-bit lanes in this mode.
static constexpr CostTable make_cost_acc()
{
	CostTable t{};
	for (int idx = 0; idx < 512; idx++) {
		int v = (int)(int8_t)(uint8_t)(idx & 255);
		if (idx & 256)
			v = (int)(int8_t)(uint8_t)(-v);
		const uint32_t c0 = v < 0 ? (uint32_t)(-v) : 0u;
		const uint32_t c1 = v > 0 ? (uint32_t)v : 0u;
		t.a[idx] = c0 | (c0 << 8) | (c1 << 16) | (c1 << 24);
		t.b[idx] = c0 | (c1 << 8) | (c0 << 16) | (c1 << 24);
	}
	return t;
}
__constant__ CostTable c_cost_acc = make_cost_acc();
// what the accelerated decoder gives state 0 as a start: 127 * N * K in correlation units (conv_acc.c reset_decoder),
// halved like the costs
constexpr uint32_t kAccLeadK5r2 = 127u * 2u * 5u / 2u;

struct SynTable { uint16_t s[208]; };
static constexpr SynTable make_syn()
{
	// CRC16 (poly 0x1021, init 0; reference src/l1/crc.c:58-63) is linear: the
	// check word of 192 message bits is the XOR of s[k] over the set bits k.
	// s[192+i] folds the received CRC bit i (MSB first) in, so that the XOR over
	// all 208 decoded bits is zero iff the check passes.
	SynTable t{};
	for (int k = 0; k < 192; k++) {
		uint32_t crc = 0x8000u;
		for (int i = k; i < 192; i++)
			crc = (crc & 0x8000u) ? (((crc << 1) ^ 0x1021u) & 0xffffu) : ((crc << 1) & 0xffffu);
		t.s[k] = (uint16_t)crc;
	}
	for (int i = 0; i < 16; i++)
		t.s[192 + i] = (uint16_t)(1u << (15 - i));
	return t;
}
__constant__ SynTable c_syn = make_syn();

// the same table laid out for the decoder's CRC stage: lane `loc` of a row owns decoded bits
// 13 loc .. 13 loc + 12; w[loc][p] = s[13 loc + 2p] | s[13 loc + 2p + 1] << 16 (two 128-bit loads per lane)
struct SynRows { uint32_t w[16][8]; };
static constexpr SynRows make_syn_rows()
{
	const SynTable t = make_syn();
	SynRows r{};
	for (int loc = 0; loc < 16; loc++)
		for (int q = 0; q < 13; q++)
			r.w[loc][q >> 1] |= (uint32_t)t.s[13 * loc + q] << (16 * (q & 1));
	return r;
}
__constant__ __attribute__((aligned(16))) SynRows c_syn_rows = make_syn_rows();

// Soft bits of a pi/4-CQPSK symbol by table (pi4cxpsk.c:452-507): the two soft bits are a function of the symbol's
// phase quantised to 1/128 symbol (dq = round(|sv - round(sv)| * 128)) -- piecewise constant with every boundary on a
// multiple of 1/256 symbol.  Cell k of the table covers phases [k, k + 1) / 1024 turns (1 turn = 4 symbols) and holds
// what the arithmetic gives at the cell's midpoint (no ties there): nearest symbol sp (Gray bits p0 p1), its neighbour
// on the side of the phase, distance dq; the bit that differs between the two gets 127 - dq, the other 127 - dq/2.
// Entry = soft bit 0 | soft bit 1 << 8.  The arithmetic form and the table differ only for phases that are exactly a
// cell boundary in binary floating point.
struct SbLut { uint16_t v[1024]; };
static constexpr SbLut make_sb_lut()
{
	SbLut t{};
	for (int k = 0; k < 1024; k++) {
		int q = 2 * k + 1;                    // cell midpoint in 1/512 symbol; a turn is 2048
		if (q > 1024)
			q -= 2048;                        // (-2, 2] symbols
		const int n = (q + 256 + 2048) / 512 - 4;   // nearest symbol, floor((q + 256) / 512)
		const int dlq = 512 * n - q;          // round(sv) - sv, odd: never zero
		const int adl = dlq < 0 ? -dlq : dlq;
		const int dq = (adl + 2) / 4;         // round(|dl| * 128): adl / 4 = m + 1/4 or m + 3/4
		const unsigned sp = (unsigned)n & 3u;
		const unsigned neg = dlq < 0 ? 1u : 0u;
		const bool f0 = ((sp ^ neg ^ 1u) & 1u) != 0;
		const int m_near = 127 - dq, m_far = 127 - (dq >> 1);
		int v0 = f0 ? m_near : m_far;
		int v1 = f0 ? m_far : m_near;
		if (sp >> 1)
			v0 = -v0;
		if ((sp ^ (sp >> 1)) & 1u)
			v1 = -v1;
		t.v[k] = (uint16_t)(((unsigned)v0 & 0xffu) | (((unsigned)v1 & 0xffu) << 8));
	}
	return t;
}
__device__ __attribute__((aligned(16))) const SbLut g_sb_lut = make_sb_lut();
constexpr int kSbLutBytes = 2048;

// The same for pi/4-CBPSK (one bit per symbol, a turn is two symbols; pi4cxpsk.c:452-507 with nbits = 1): the symbol's
// one soft bit is 127 - dq (its neighbour always differs in that bit), negative for symbol 1.  Entry = the soft bit in
// the low byte; cells and boundaries as above (dq steps at odd multiples of 1/256 symbol, cell edges at multiples of 1/512).
static constexpr SbLut make_sb_lut1()
{
	SbLut t{};
	for (int k = 0; k < 1024; k++) {
		int q = 2 * k + 1;                    // cell midpoint in 1/1024 symbol; a turn is 2048
		if (q > 1024)
			q -= 2048;                        // (-1, 1] symbols
		const int n = (q + 512 + 2048) / 1024 - 2;  // nearest symbol, floor((q + 512) / 1024)
		const int dlq = 1024 * n - q;         // round(sv) - sv, odd: never zero
		const int adl = dlq < 0 ? -dlq : dlq;
		const int dq = (adl + 4) / 8;         // round(|dl| * 128): adl / 8 is never half an integer
		const unsigned sp = (unsigned)n & 1u;
		const int v0 = sp ? -(127 - dq) : (127 - dq);
		t.v[k] = (uint16_t)((unsigned)v0 & 0xffu);
	}
	return t;
}
__device__ __attribute__((aligned(16))) const SbLut g_sb_lut1 = make_sb_lut1();

// ---------------------------------------------------------------------------
// cross-lane helpers (DPP: no LDS traffic)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
	return __builtin_bit_cast(float, dpp<CTRL>(__builtin_bit_cast(uint32_t, v)));
}

// the partner's value in the steps 1, 2, 4, 8 of a reduction over a 16-lane row (every use in this file is such a
// reduction by a commutative operation, run in that order): lane l ^ X for X in {8, 2, 1}; for X = 4 lane 7 - (l & 7) of
// the half -- a lane of the half's OTHER quad, whose four lanes all hold that quad's value after steps 1 and 2 -- which is
// one DPP operand instead of the two moves an exact l ^ 4 takes, with the same result bit for bit
template <int X>
__device__ __forceinline__ uint32_t row_xor(uint32_t v)
{
	if constexpr (X == 8) return dpp<0x128>(v);                    // row_ror:8
	else if constexpr (X == 4) return dpp<0x141>(v);               // row_half_mirror
	else if constexpr (X == 2) return dpp<0x4E>(v);                // quad_perm [2,3,0,1]
	else return dpp<0xB1>(v);                                      // quad_perm [1,0,3,2]
}
template <int X>
__device__ __forceinline__ float row_xorf(float v)
{
	return __builtin_bit_cast(float, row_xor<X>(__builtin_bit_cast(uint32_t, v)));
}

// every lane gets the sum over its 16-lane row
__device__ __forceinline__ float row_sum(float v)
{
	v += row_xorf<1>(v);
	v += row_xorf<2>(v);
	v += row_xorf<4>(v);
	v += row_xorf<8>(v);
	return v;
}

__device__ __forceinline__ float lane_val(float v, int l)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// (r0 + r16) + (r32 + r48) of the four row sums, wave-uniform.  The two cross-row steps are DPP row broadcasts: lane 15 of
// every row into the next row (row 1 then holds r16 + r0, row 3 r48 + r32), then lane 31 into rows 2 and 3 (row 3:
// (r48 + r32) + (r16 + r0)) -- the same three additions, operands swapped, so the same float; one readlane instead of four
// and no moves back from scalar registers.
__device__ __forceinline__ float wave_sum(float v)
{
	v = row_sum(v);
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xf, 0xf, true));   // row_bcast:15
	v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xf, 0xf, true));   // row_bcast:31 (rows 0, 1: + 0)
	return lane_val(v, 63);
}

// ---------------------------------------------------------------------------
// math helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// sin / cos for |x| up to a few thousand radians: two-constant Cody-Waite reduction
// by pi/2 (exact to ~1e-10 thanks to fma) and the classic single-precision minimax
// polynomials on [-pi/4, pi/4]; ~1 ulp, so results track libm's to the last bit or two.
__device__ __forceinline__ void sincos_fast(float x, float &s, float &c)
{
	const float k = rintf(x * 0.636619772367581343f);
	float r = fmaf(-k, 1.57079637050628662109375f, x);
	r = fmaf(-k, -4.37113900018624283e-8f, r);
	const float z = r * r;
	float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
	sp = fmaf(sp, z, -1.6666654611e-1f);
	sp = fmaf(sp * z, r, r);
	float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
	cp = fmaf(cp, z, 4.166664568298827e-2f);
	cp = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
	const int q = (int)k;
	const float ss = (q & 1) ? cp : sp;
	const float cc = (q & 1) ? sp : cp;
	s = (q & 2) ? -ss : ss;
	c = ((q + 1) & 2) ? -cc : cc;
}

// atan2 with ~1.5e-7 absolute error: octant folding + one reciprocal + degree-9 minimax
__device__ __forceinline__ float atan2_fast(float y, float x)
{
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	const bool big = mn > 0.41421356237f * mx;             // tan(pi/8)
	const float num = big ? (mn - mx) : mn;
	const float den = big ? (mn + mx) : mx;
	const float t = num * __builtin_amdgcn_rcpf(den);
	const float z = t * t;
	float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
	p = fmaf(p, z, 1.99777106478e-1f);
	p = fmaf(p, z, -3.33329491539e-1f);
	float a = fmaf(p * z, t, t);
	a += big ? 0.785398163397448309f : 0.0f;
	a = (ay > ax) ? (1.57079632679489662f - a) : a;
	a = (x < 0.0f) ? (kPif - a) : a;
	a = (mx == 0.0f) ? 0.0f : a;
	return (y < 0.0f) ? -a : a;
}

// atan2(y, x) / (2 pi), same minimax polynomial as atan2_fast with the coefficients in turns;
// atan2_turns(0, 0) = 0
__device__ __forceinline__ float atan2_turns(float y, float x)
{
	const float ax = fabsf(x), ay = fabsf(y);
	const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	const bool big = mn > 0.41421356237f * mx;             // tan(pi/8)
	const float num = big ? (mn - mx) : mn;
	const float den = big ? (mn + mx) : mx;
	const float t = num * __builtin_amdgcn_rcpf(den);
	const float z = t * t;
	float p = fmaf(z, 1.28179325e-2f, -2.20870226e-2f);     // atan2_fast's coefficients / (2 pi)
	p = fmaf(p, z, 3.17955140e-2f);
	p = fmaf(p, z, -5.30510363e-2f);
	p = fmaf(p, z, 1.59154943e-1f);
	float a = p * t;
	a += big ? 0.125f : 0.0f;
	a = (ay > ax) ? (0.25f - a) : a;
	a = (x < 0.0f) ? (0.5f - a) : a;
	a = (mx == 0.0f) ? 0.0f : a;
	return __builtin_copysignf(a, y);
}

// conj(ref) * v for ref = modulating value of sync symbol `sym` (exact: ref is +-1 / +-j)
__device__ __forceinline__ float2 conj_ref_mul(int nbits, int sym, float2 v)
{
	if (nbits == 2) {
		// sym0: ( x, y)  sym1: ( y,-x)  sym2: (-x,-y)  sym3: (-y, x)
		const bool odd = (sym & 1) != 0;
		const float a = odd ? v.y : v.x, b = odd ? v.x : v.y;
		return make_float2((sym & 2) ? -a : a, ((sym + 1) & 2) ? -b : b);
	}
	return (sym & 1) ? make_float2(-v.x, -v.y) : v;
}

// K=5 rate-1/2 code (g0 = 1+D^3+D^4, g1 = 1+D+D^2+D^4; reference src/l1/conv.c:123-145)
__device__ __forceinline__ uint32_t out_k5_12(uint32_t s, uint32_t b)
{
	uint32_t reg = (s << 1) | b;
	return ((uint32_t)(__popc(reg & 0x19u) & 1) << 1) | (uint32_t)(__popc(reg & 0x17u) & 1);
}
__device__ __forceinline__ uint32_t rotl4(uint32_t x, int r) { return ((x << r) | (x >> (4 - r))) & 15u; }

// ---------------------------------------------------------------------------
// LDS carve-up of one wavefront
//   [x | aux | eb]   aux = corr + coef during the sync search, y afterwards
//   after the 4 demods of a fused wave, bm and surv overlay x
// ---------------------------------------------------------------------------
struct Lds {
	float2 *x;        // normalised input window           [max_in_len]
	float *corr;      // accumulated sync correlation      [kMaxWindow]      (aux)
	float2 *coef;     // rotated sync reference            [kMaxCoef]        (aux + 1 KiB)
	float2 *y;        // decimated symbols                 [max_len]         (aux)
	int8_t *eb;       // soft bits: 4 rows (fused) or one buffer
	uint32_t *bm;     // branch metrics 4 x 212            (overlays x)
	uint64_t *surv;   // 13 x 64 halfwords of window decisions (overlays x, after bm)
	uint32_t *ubits;  // decoded bits, 4 rows x 8 words    (overlays x, after surv)
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline size_t lds_corr_bytes(int max_len, bool decode)
{
	if (decode)
		return align16((size_t)max_len * 4);
	return align16((size_t)(max_len > kMaxWindow ? max_len : kMaxWindow) * 4);
}

__host__ __device__ inline size_t lds_layout(int max_in_len, int max_len, bool decode, size_t *off)
{
	size_t o = 0;
	size_t xbytes = align16((size_t)max_in_len * 8);
	const size_t dec_bytes = 4 * kSteps12 * 4 + kSteps12 * 8 + 4 * 8 * 4;
	if (decode && xbytes < dec_bytes)
		xbytes = align16(dec_bytes);
	off[0] = o; o += xbytes;
	// aux = correlation accumulator + rotated sync reference.  The fused BCCH / CCCH path knows
	// its formats (<= max_len lags, 17 sync symbols), which keeps 15 wavefronts per CU resident
	// (demodulation only: the caller's lag count when it passes the 256 the layout has always had room for)
	const size_t corr_bytes = lds_corr_bytes(max_len, decode);
	const size_t coef_bytes = decode ? 32 * 8 : (size_t)kMaxCoef * 8;
	off[1] = o; o += corr_bytes + coef_bytes;
	off[2] = o; o += decode ? 4 * kEbRow : kEbitsLds;
	return align16(o);
}

__device__ __forceinline__ Lds lds_carve(unsigned char *raw, int max_in_len, int max_len, bool decode)
{
	size_t off[3];
	lds_layout(max_in_len, max_len, decode, off);
	Lds L;
	L.x = reinterpret_cast<float2 *>(raw + off[0]);
	L.corr = reinterpret_cast<float *>(raw + off[1]);
	L.coef = reinterpret_cast<float2 *>(raw + off[1] + lds_corr_bytes(max_len, decode));
	L.y = reinterpret_cast<float2 *>(raw + off[1]);
	L.eb = reinterpret_cast<int8_t *>(raw + off[2]);
	L.bm = reinterpret_cast<uint32_t *>(raw + off[0]);
	L.surv = reinterpret_cast<uint64_t *>(raw + off[0] + 4 * kSteps12 * 4);
	L.ubits = reinterpret_cast<uint32_t *>(raw + off[0] + 4 * kSteps12 * 4 + kSteps12 * 8);
	return L;
}

// ---------------------------------------------------------------------------
// building blocks of the demodulator, one burst per wavefront
// ---------------------------------------------------------------------------

// window HBM -> registers -> LDS, DC and power normalised
template <int NPL>
__device__ __forceinline__ void load_normalise_stats(const float2 *__restrict__ in, int in_len, const Lds &L, int lane,
                                                     float &avr_o, float &avi_o, float &inv_o)
{
	// ---- load + normalise (osmo_cxvec_sig_normalize, decim 1) ------------------
	// rows k < nfull are whole (no lane test); row nfull is the ragged tail
	float2 v[NPL];
	float sr = 0.f, si = 0.f;
	const int nfull = in_len >> 6;
	const bool tail = (lane + 64 * nfull) < in_len;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		if (k < nfull)
			v[k] = in[lane + 64 * k];
		else if (k == nfull && tail)
			v[k] = in[lane + 64 * k];
		else
			v[k] = make_float2(0.f, 0.f);
		sr += v[k].x;
		si += v[k].y;
	}
	sr = wave_sum(sr);
	si = wave_sum(si);
	// mean / sigma only fix the DC offset and an overall scale that nothing downstream depends
	// on, so reciprocals (1 ulp) stand in for the reference's divisions and square root
	const float inv_n = __builtin_amdgcn_rcpf((float)in_len);
	// (the mean keeps the true division: a constant window must normalise to exactly zero)
	const float avr = sr / (float)in_len, avi = si / (float)in_len;
	float acc = 0.f;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		if (k < nfull || (k == nfull && tail)) {
			v[k].x -= avr;
			v[k].y -= avi;
			acc = fmaf(v[k].x, v[k].x, fmaf(v[k].y, v[k].y, acc));
		}
	}
	float sigma = wave_sum(acc) * inv_n;
	float stddev = __builtin_amdgcn_sqrtf(sigma);
	if (stddev == 0.0f)
		stddev = 1.0f;
	const float inv = __builtin_amdgcn_rcpf(stddev);
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		if (k < nfull || (k == nfull && tail))
			L.x[lane + 64 * k] = make_float2(v[k].x * inv, v[k].y * inv);
	}
	avr_o = avr;
	avi_o = avi;
	inv_o = inv;
}

// window statistics only (mean, 1/sigma); the samples stay in registers and are dropped
// NFULL >= 0: the caller knows in_len >> 6 at compile time (the fused sps = 4 path: 1016 and 976
// samples both have 15 whole rows), which removes the per-row branches
// RS = 64: `in` is the window, lane l takes samples l + 64 k.  RS = 16 (polyphase-planar array at 4 samples per symbol,
// rx4_body's PL): `in` is already this lane's first sample in its plane, sample l + 64 k is 16 k places further on.
template <int NPL, int NFULL = -1, int RS = 64>
__device__ __forceinline__ void window_fetch(const float2 *__restrict__ in, int in_len, int lane, float2 (&v)[NPL])
{
	const int nfull = NFULL >= 0 ? NFULL : (in_len >> 6);
	const bool tail = (lane + 64 * nfull) < in_len;
	const int l0 = RS == 64 ? lane : 0;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		if (k < nfull)
			v[k] = in[l0 + RS * k];
		else if (k == nfull && tail)
			v[k] = in[l0 + RS * k];
		else
			v[k] = make_float2(0.f, 0.f);
	}
}

// perm_src >= 0 (rx4_body's PL): this lane holds the samples of ANOTHER lane of the usual assignment (`lane` names that
// one); the per-lane partial sums -- formed over the same samples in the same order -- are first moved to the lane that
// usually forms them (every lane fetches from lane perm_src), so the cross-lane sums add the same numbers in the same
// order and the statistics come out bit-identical.
template <int NPL, int NFULL = -1>
__device__ __forceinline__ void window_stats(const float2 (&v)[NPL], int in_len, int lane,
                                             float &avr_o, float &avi_o, float &inv_o, int perm_src = -1, int odd_src = 1)
{
	auto home = [&](float x) {
		return perm_src < 0 ? x : __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm_src << 2, __builtin_bit_cast(int, x)));
	};
	// (re, im) pairs through the packed FP32 pipe: one v_pk_add_f32 per sample for the sums, one v_pk_add_f32 and
	// one v_pk_fma_f32 for the variance (re and im are summed in separate chains, as they are in the mean)
	v2f s2 = {0.f, 0.f};
	const int nfull = NFULL >= 0 ? NFULL : (in_len >> 6);
	const bool tail = (lane + 64 * nfull) < in_len;
#pragma unroll
	for (int k = 0; k < NPL; k++)
		s2 += (v2f){v[k].x, v[k].y};
	const float sr = wave_sum(home(s2.x));
	const float si = wave_sum(home(s2.y));
	const float inv_n = __builtin_amdgcn_rcpf((float)in_len);
	// true division, see load_normalise -- ONE division sequence for the two wave-uniform sums: lanes with an even `lane`
	// divide the real sum, those with an odd one (wave lane odd_src is one) the imaginary sum
	const float quot = ((lane & 1) ? si : sr) / (float)in_len;
	const float avr = lane_val(quot, 0), avi = lane_val(quot, odd_src);
	const v2f av = {avr, avi};
	v2f acc2 = {0.f, 0.f};
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		if (k < nfull || (k == nfull && tail)) {
			const v2f d = (v2f){v[k].x, v[k].y} - av;
			acc2 = __builtin_elementwise_fma(d, d, acc2);
		}
	}
	float stddev = __builtin_amdgcn_sqrtf(wave_sum(home(acc2.x + acc2.y)) * inv_n);
	if (stddev == 0.0f)
		stddev = 1.0f;
	avr_o = avr;
	avi_o = avi;
	inv_o = __builtin_amdgcn_rcpf(stddev);
}

template <int NPL, int NFULL = -1>
__device__ __forceinline__ void load_stats(const float2 *__restrict__ in, int in_len, int lane,
                                           float &avr_o, float &avi_o, float &inv_o)
{
	float2 v[NPL];
	window_fetch<NPL, NFULL>(in, in_len, lane, v);
	window_stats<NPL, NFULL>(v, in_len, lane, avr_o, avi_o, inv_o);
}

// ---- the QUAD layout of a window in registers (rx4_body's QL: the fused batch kernel at 4 samples per symbol) ----
// Lane l holds window samples 256 b + 4 l + c as v[4 b + c] (b = 0..3, c = 0..3): four CONSECUTIVE samples of each quarter
// of the window, fetched as two 16-byte loads a quarter (a wave instruction covers 1 KB, every line asked for whole).
// What it buys: the samples pass 2 keeps -- d, d + 4, d + 8, ... (pi4cxpsk.c:292-295) -- are sub-slot c = d & 3 of EVERY lane
// of every quarter, one per lane and quarter, in lane order: kept sample i sits in lane (i + (d >> 2)) & 63 of quarter
// (i + (d >> 2)) >> 6.  A lane ROTATION by d >> 2 (ds_bpermute, no LDS memory) puts kept sample l + 64 r into lane l --
// pass 2's own assignment -- while the window is still in registers: no second trip to memory for it (rx4_body, QX).
// With the samples stored polyphase-planar the same assignment is lane l <- place l + 64 b of plane c: a coalesced 512-byte
// load, and the per-lane partial sums below are formed over the same samples in the same order, so the planar call's
// statistics equal the interleaved call's bit for bit.
typedef float v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ void window_fetch_q(const float2 *__restrict__ in, int in_len, int lane, float2 (&v)[16])
{
	const v4f_a8 *__restrict__ p = reinterpret_cast<const v4f_a8 *>(in + 4 * lane);
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const int s0 = 256 * b + 4 * lane;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			float2 lo = make_float2(0.f, 0.f), hi = make_float2(0.f, 0.f);
			if (b < 3 || s0 + 2 * h + 1 < in_len) {        // (in_len >= 960: the first three quarters are whole)
				const v4f_a8 u = p[128 * b + h];
				lo = make_float2(u.x, u.y);
				hi = make_float2(u.z, u.w);
			} else if (s0 + 2 * h < in_len) {
				lo = in[s0 + 2 * h];
			}
			v[4 * b + 2 * h] = lo;
			v[4 * b + 2 * h + 1] = hi;
		}
	}
}

// the same assignment out of a polyphase-planar array: `pl` = the array, o = the window's first sample (flat count)
__device__ __forceinline__ void window_fetch_q_planar(const float2 *__restrict__ pl, long long plane_stride, uint64_t o, int in_len,
                                                      int lane, float2 (&v)[16])
{
#pragma unroll
	for (int c = 0; c < 4; c++) {
		const uint64_t oc = o + (uint64_t)c;
		const float2 *__restrict__ src = pl + (long long)(oc & 3) * plane_stride + (long long)(oc >> 2) + lane;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int sidx = 256 * b + 4 * lane + c;
			v[4 * b + c] = (b < 3 || sidx < in_len) ? src[64 * b] : make_float2(0.f, 0.f);
		}
	}
}

// mean and 1 / sigma of a window in the quad layout (osmo_cxvec_sig_normalize's statistics), ONE sweep: sum x and sum |x|^2
// together -- sum |x - m|^2 = sum |x|^2 - n |m|^2, never below zero -- as the small formats' pass 1 has always had it: sigma only
// sets a scale nothing downstream depends on (every consumer takes an angle, a ratio of energies or the place of a peak), and a
// second sweep over sixteen register pairs for the variance about the mean is a third of the statistics' instructions.  Packed
// sums, ONE true division sequence for the two means (a constant window must normalise to exactly zero), reciprocals elsewhere.
__device__ __forceinline__ void window_stats_q(const float2 (&v)[16], int in_len, int lane, float &avr_o, float &avi_o, float &inv_o)
{
	v2f s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const v2f x = {v[k].x, v[k].y};                     // (samples beyond the window are zeros)
		s2 += x;
		q2 = __builtin_elementwise_fma(x, x, q2);
	}
	const float sr = wave_sum(s2.x);
	const float si = wave_sum(s2.y);
	const float sq = wave_sum(q2.x + q2.y);
	const float inv_n = __builtin_amdgcn_rcpf((float)in_len);
	const float quot = ((lane & 1) ? si : sr) / (float)in_len;
	const float avr = lane_val(quot, 0), avi = lane_val(quot, 1);
	const float var = fmaxf(fmaf(-(float)in_len, fmaf(avr, avr, avi * avi), sq), 0.0f) * inv_n;
	float stddev = __builtin_amdgcn_sqrtf(var);
	if (stddev == 0.0f)
		stddev = 1.0f;
	avr_o = avr;
	avi_o = avi;
	inv_o = __builtin_amdgcn_rcpf(stddev);
}

// burst_energy() of the caller (gmr1_rx.c:172-182): sum |x|^2 over [len>>5, len - len>>5) of the RAW
// window, divided by len.  Only the receive driver asks for it (RxArgs::energy); the window was
// read a moment ago, so this second read is served by L1 / L2.
template <int NPL>
__device__ __noinline__ float window_energy(const float2 *__restrict__ in, int in_len, int lane)
{
	const int bd = in_len >> 5;
	float e = 0.f;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		const int idx = lane + 64 * k;
		if (idx >= bd && idx < in_len - bd) {
			const float2 v = in[idx];
			e = fmaf(v.x, v.x, fmaf(v.y, v.y, e));
		}
	}
	return wave_sum(e) / (float)in_len;
}

// the same sum over a window that is still in registers (window_fetch layout): identical operations in
// identical order, without the second read
template <int NPL>
__device__ __forceinline__ float window_energy_regs(const float2 (&v)[NPL], int in_len, int lane)
{
	const int bd = in_len >> 5;
	float e = 0.f;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		const int idx = lane + 64 * k;
		if (idx >= bd && idx < in_len - bd)
			e = fmaf(v[k].x, v[k].x, fmaf(v[k].y, v[k].y, e));
	}
	return wave_sum(e) / (float)in_len;
}

template <int NPL>
__device__ __forceinline__ void load_normalise(const float2 *__restrict__ in, int in_len, const Lds &L, int lane)
{
	float a, b, c;
	load_normalise_stats<NPL>(in, in_len, L, lane, a, b, c);
}

// sync sequence search over the normalised window in L.x with derotation step fs (rad/sample).
// Returns the winning sequence (-1: none has power), its fractional TOA and power.
template <int SPS>
__device__ int sync_search(int type, int in_len, int sps_rt, float fs, const Lds &L, int lane,
                           int dbg_stop, float &toa_o, float &pwr_o)
{
	const DevBurst &bt = c_types[type];
	const int sps = SPS ? SPS : sps_rt;
	const int nbits = bt.nbits;
	const int w = in_len - bt.len * sps + 1;
	WSYNC();
	for (int j = lane; j < w; j += 64)
		L.corr[j] = 0.f;

	// ---- sync search (pi4cxpsk.c:184-268) --------------------------------------
	float p_toa = 0.f, p_pwr = 0.f;
	int p_idx = -1;
	const int win = w < 3 ? w : 3;
	const int nsync = bt.n_sync;

	for (int sq = 0; sq < nsync; sq++) {
		const int tl = bt.sync_tl[sq];
		const int nch = bt.n_chunks[sq];

		// rotated reference: conj(ref_n) * e^{j fs sps n}; the common phase of a lag
		// drops out under |.|, so the window itself is never derotated here
		WSYNC();
		for (int n = lane; n < tl; n += 64) {
			int ch = 0, base = 0, cum = 0;
			for (int c = 0; c < nch - 1; c++) {
				cum += bt.sync[sq][c].len;
				if (n >= cum) { base = cum; ch = c + 1; }
			}
			const int nn = n - base;
			const int sym = bt.sync[sq][ch].syms[nn];
			float s, c;
			sincos_fast(fs * (float)(nn * sps), s, c);
			L.coef[n] = conj_ref_mul(nbits, sym, make_float2(c, s));
		}
		WSYNC();

		for (int j = lane; j < w; j += 64) {
			float cj = L.corr[j];
			int base = 0;
			for (int ch = 0; ch < nch; ch++) {
				const int pos = bt.sync[sq][ch].pos, len = bt.sync[sq][ch].len;
				const float2 *xp = L.x + pos * sps + j;
				const float2 *cp = L.coef + base;
				float ar = 0.f, ai = 0.f;
				for (int n = 0; n < len; n++) {
					const float2 x = xp[n * sps];
					const float2 cf = cp[n];
					ar = fmaf(cf.x, x.x, fmaf(-cf.y, x.y, ar));
					ai = fmaf(cf.x, x.y, fmaf(cf.y, x.x, ai));
				}
				base += len;
				cj += sqrtf(fmaf(ar, ar, ai * ai));
			}
			L.corr[j] = cj;
		}
		WSYNC();
		if (dbg_stop == 2) return -100;

		// ---- osmo_cxvec_peak_energy_find(corr, 3, PEAK_EARLY_LATE, &peak) ----------
		// key = (energy bits << 32) | ~index : max key = highest energy, lowest index on ties
		unsigned long long key = 0;
		for (int m = lane; m + win <= w; m += 64) {
			float e = 0.f;
			for (int k = 0; k < win; k++) {
				const float c = L.corr[m + k];
				e += c * c;
			}
			const unsigned long long kk =
				((unsigned long long)__builtin_bit_cast(uint32_t, e) << 32) | (uint32_t)(~m);
			key = kk > key ? kk : key;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const unsigned long long ok = __shfl_xor(key, o);
			key = ok > key ? ok : key;
		}
		int mi = (int)(~(uint32_t)key);
		if (mi < 0 || mi + win > w)
			mi = 0;
		int p = mi;
		{
			float pe = -1.f;
			for (int k = 0; k < win; k++) {
				const float c = L.corr[mi + k];
				const float e = c * c;
				if (e > pe) { pe = e; p = mi + k; }
			}
		}

		// sinc-interpolated corr at `pos` (libosmo-dsp interpolate_point, 21 taps):
		// lanes t = 0..20 of each 32-lane half hold one tap of that half's position.
		// tap weight sinc(pi (i - pos)) with i - pos = k - f  ->  -(-1)^k sin(pi f) / (pi (k - f))
		const int t = lane & 31;
		auto interp_term = [&](float pos) -> float {
			const float fl = floorf(pos);
			const int i0 = (int)fl;
			const float f = pos - fl;
			int b = i0 - 10, e = i0 + 11;
			if (b < 0) b = 0;
			if (e >= w) e = w - 1;
			const int i = i0 - 10 + t;
			const bool valid = t < 21 && i >= b && i < e;
			const float xx = kPif * ((float)i - pos);
			const float S = __builtin_amdgcn_sinf(0.5f * f);       // sin(pi f), argument in turns
			const float sg = (t & 1) ? S : -S;                     // k = t - 10 has the parity of t
			const float wgt = (xx >= 0.01f || xx <= -0.01f) ? sg * __builtin_amdgcn_rcpf(xx) : 1.0f;
			const float c = L.corr[valid ? i : 0];
			return valid ? c * wgt : 0.0f;
		};
		auto half_total = [&](float v, int half) -> float {
			v = row_sum(v);
			return lane_val(v, 32 * half) + lane_val(v, 32 * half + 16);
		};

		// Early / late bisection (incr = 1/2 ... 1/512), THREE levels per evaluation: each group of 8 lanes
		// interpolates the correlation at one candidate position and two samples later (same fractional part,
		// so the same 21 weights; lane sub holds taps k = 3 sub - 10 + {0,1,2}) -- group 0 at the current point,
		// groups 1 / 2 where the search goes if the early / late side wins, groups 3..6 one level further down.
		// The candidates are formed by the same float operations the level-by-level walk performs, so it takes
		// the same decisions; the walk itself is scalar work on two ballots.
		const int grp = lane >> 3, isub = lane & 7;
		auto interp_pair = [&](float pos, float &se, float &sl) {
			const float fl = floorf(pos);
			const int ib = (int)fl;
			const float f = pos - fl;
			const float S = __builtin_amdgcn_sinf(0.5f * f);       // sin(pi f); sin(pi (k - f)) = -(-1)^k sin(pi f)
			int be = ib - 10, ee = ib + 11, bl = ib - 8, el = ib + 13;
			if (be < 0) be = 0;
			if (bl < 0) bl = 0;
			if (ee >= w) ee = w - 1;
			if (el >= w) el = w - 1;
			float ae = 0.f, al = 0.f;
#pragma unroll
			for (int tt = 0; tt < 3; tt++) {
				const int k = 3 * isub - 10 + tt;
				const float sg = ((isub + tt) & 1) ? S : -S;
				const float xx = kPif * ((float)k - f);
				const float wgt = (xx >= 0.01f || xx <= -0.01f) ? sg * __builtin_amdgcn_rcpf(xx) : 1.0f;
				const int ie = ib + k, il = ib + 2 + k;
				const bool ve = k <= 10 && ie >= be && ie < ee;
				const bool vl = k <= 10 && il >= bl && il < el;
				const float ce = L.corr[ve ? ie : 0], cl = L.corr[vl ? il : 0];
				ae += ve ? ce * wgt : 0.0f;
				al += vl ? cl * wgt : 0.0f;
			}
			ae += row_xorf<1>(ae);
			ae += row_xorf<2>(ae);
			ae += row_xorf<4>(ae);
			al += row_xorf<1>(al);
			al += row_xorf<2>(al);
			al += row_xorf<4>(al);
			se = ae;
			sl = al;
		};
		float early = (float)p - 1.0f, incr = 0.5f;
#pragma unroll 1
		for (int it = 0; it < 3; it++) {
			const float half = incr * 0.5f, quarter = incr * 0.25f;
			float pos = early;
			if (grp == 1) {
				pos = early - incr;
			} else if (grp == 2) {
				pos = early + incr;
			} else if (grp >= 3 && grp <= 6) {
				const float a1 = grp < 5 ? early - incr : early + incr;
				pos = (grp & 1) ? a1 - half : a1 + half;           // 3: - -, 4: - +, 5: + -, 6: + +
			}
			float se, sl;
			interp_pair(pos, se, sl);
			const float ee = se * se, le = sl * sl;
			const unsigned long long m_neg = __ballot(ee > le), m_pos = __ballot(ee < le);
			auto dec = [&](int g) -> int { return ((m_neg >> (8 * g)) & 1ull) ? -1 : (((m_pos >> (8 * g)) & 1ull) ? 1 : 0); };
			const int d0 = dec(0);
			if (d0 == 0) break;
			early = d0 < 0 ? early - incr : early + incr;
			const int d1 = dec(d0 < 0 ? 1 : 2);
			if (d1 == 0) break;
			early = d1 < 0 ? early - half : early + half;
			const int d2 = dec(3 + (d0 > 0 ? 2 : 0) + (d1 > 0 ? 1 : 0));
			if (d2 == 0) break;
			early = d2 < 0 ? early - quarter : early + quarter;
			incr *= 0.125f;
		}
		const float s_toa = early + 1.0f;
		float pk = half_total(interp_term(s_toa), 0);
		pk = pk * __builtin_amdgcn_rcpf((float)tl);     // only ranked and tested against 0
		const float s_pwr = pk * pk;
		if (s_pwr > p_pwr) {
			p_pwr = s_pwr;
			p_toa = s_toa;
			p_idx = sq;
		}
	}
	toa_o = p_toa;
	pwr_o = p_pwr;
	return p_idx;
}

// ---------------------------------------------------------------------------
// demodulation of one burst by one wavefront
// returns the reference's rv (0, or -1 when no sync sequence has power)
// ---------------------------------------------------------------------------
template <int NPL, int SPS>
__device__ int demod_one(int type, const float2 *__restrict__ in, int in_len, int sps_rt,
                         float freq_shift, const Lds &L, int8_t *__restrict__ eb, int lane,
                         int dbg_stop, int &sync_id_o, float &toa_o, float &ferr_o,
                         float *__restrict__ g_ssyms)
{
	const DevBurst &bt = c_types[type];
	const int sps = SPS ? SPS : sps_rt;
	const int nbits = bt.nbits;
	const int blen = bt.len;

	load_normalise<NPL>(in, in_len, L, lane);
	if (dbg_stop == 1) return -100;

	// per-sample derotation step (pi4cxpsk.c:539)
	const float fs = (freq_shift - bt.rotation) / (float)sps;
	float p_toa = 0.f, p_pwr = 0.f;
	const int p_idx = sync_search<SPS>(type, in_len, sps_rt, fs, L, lane, dbg_stop, p_toa, p_pwr);
	if (p_idx == -100 || dbg_stop == 3) return -100;

	sync_id_o = p_idx;
	toa_o = p_toa;
	if (p_idx < 0) {
		ferr_o = 0.f;
		return -1;
	}
	const int sq = p_idx;
	const int nch = bt.n_chunks[sq];

	// ---- everything after the sync search works in the PHASE domain ----------------
	// The reference rotates the decimated burst three times (derotation e^{j fs n},
	// fine frequency e^{-j f i}, carrier conj(phasor)) and then takes cargf() of each
	// symbol (pi4cxpsk.c:286-297,574-581,442-460).  arg() of that product is
	//     arg(x[i sps + d]) + fs (i sps + d) - f i - arg(phasor)      (mod 2 pi)
	// so only the <= 17 sync symbols are ever rotated as complex numbers; the 234
	// symbols cost one atan2 and a few adds each.  Soft bits only depend on the phase.
	const int d = (int)roundf(p_toa);
	const int row = lane >> 4, col = lane & 15;

	// align (pi4cxpsk.c:280-348): at sps >= 4 symbol i is sample i*sps + d.  Below 4 samples per
	// symbol the reference first applies a 21-tap sinc fractional delay (osmo_cxvec_convolve,
	// CONV_NO_DELAY) when |toa - d| > 0.1.  It does so on the DEROTATED burst; with
	// g[m] = x[m] e^{j fs m} the delayed sample is e^{j fs n} sum_k (p_k e^{j fs (10-k)}) x[n+10-k], so the
	// rotation moves into 21 complex taps and the common e^{j fs n} stays in the phase domain.
	const float ofs_frac = p_toa - (float)d;
	const bool frac_on = (sps < 4) && (fabsf(ofs_frac) > 0.1f);
	if (frac_on) {
		WSYNC();
		if (lane < 21) {
			const float xx = kPif * ((float)(lane - 10) + ofs_frac);
			const float pv = (xx >= 0.01f || xx <= -0.01f) ? (sinf(xx) / xx) : 1.0f;
			float s, c;
			sincos_fast(fs * (float)(10 - lane), s, c);
			L.coef[lane] = make_float2(pv * c, pv * s);
		}
		WSYNC();
	}
	auto pick = [&](int j) -> float2 {
		if (j < 0 || j >= in_len)
			return make_float2(0.f, 0.f);
		if (!frac_on)
			return L.x[j];
		float2 acc = make_float2(0.f, 0.f);
		for (int k = 0; k < 21; k++) {
			const int m = j + 10 - k;
			if (m >= 0 && m < in_len) {
				const float2 q = L.coef[k], x = L.x[m];
				acc.x = fmaf(q.x, x.x, fmaf(-q.y, x.y, acc.x));
				acc.y = fmaf(q.x, x.y, fmaf(q.y, x.x, acc.y));
			}
		}
		return acc;
	};

	auto reduce_2pi = [](float a) -> float {
		const float k = rintf(a * 0.159154943091895336f);
		a = fmaf(-k, 6.2831854820251465f, a);
		return fmaf(-k, -1.7484555e-7f, a);
	};
	// conj(ref) * derotated sample of sync symbol j of chunk c (pi4cxpsk.c:386-388)
	auto sync_term = [&](int c, int j) -> float2 {
		const int idx = (bt.sync[sq][c].pos + j) * sps + d;
		float2 x = pick(idx);
		float s, cc;
		sincos_fast(fs * (float)idx, s, cc);
		x = cmul(x, make_float2(cc, s));
		return conj_ref_mul(nbits, bt.sync[sq][c].syms[j], x);
	};

	// ---- fine frequency error from the sync chunks (pi4cxpsk.c:360-406) ---------
	// one chunk per 16-lane row, one sync symbol per lane; chunk sums by DPP
	float ffe = 0.f;
	if (nch > 1) {
		float sumr[kMaxChunks], sumi[kMaxChunks];
#pragma unroll
		for (int c0 = 0; c0 < kMaxChunks; c0 += 4) {
			if (c0 < nch) {
				const int c = c0 + row;
				float tr = 0.f, ti = 0.f;
				if (c < nch) {
					const int len = bt.sync[sq][c].len;
					for (int j = col; j < len; j += 16) {
						const float2 tt = sync_term(c, j);
						tr += tt.x;
						ti += tt.y;
					}
				}
				tr = row_sum(tr);
				ti = row_sum(ti);
#pragma unroll
				for (int r = 0; r < 4; r++) {
					sumr[c0 + r] = lane_val(tr, 16 * r);
					sumi[c0 + r] = lane_val(ti, 16 * r);
				}
			}
		}
		float f = 0.f;
#pragma unroll
		for (int i = 1; i < kMaxChunks; i++) {
			if (i < nch) {
				const float ppos = (float)bt.sync[sq][i - 1].pos + (float)bt.sync[sq][i - 1].len / 2.0f;
				const float cpos = (float)bt.sync[sq][i].pos + (float)bt.sync[sq][i].len / 2.0f;
				// corr[i] * conj(corr[i-1])
				const float re = sumr[i] * sumr[i - 1] - sumi[i] * (-sumi[i - 1]);
				const float im = sumr[i] * (-sumi[i - 1]) + sumi[i] * sumr[i - 1];
				f += atan2_fast(im, re) / (cpos - ppos);
			}
		}
		f /= (float)(nch - 1);
		ffe = f;
	}
	ferr_o = ffe;
	const float rps = -ffe;            // pi4cxpsk.c:574-575
	if (dbg_stop == 4) return -100;

	// ---- carrier phase from the (frequency-corrected) sync symbols (pi4cxpsk.c:415-433)
	float tr = 0.f, ti = 0.f;
#pragma unroll
	for (int c0 = 0; c0 < kMaxChunks; c0 += 4) {
		if (c0 < nch) {
			const int c = c0 + row;
			if (c < nch) {
				const int pos = bt.sync[sq][c].pos, len = bt.sync[sq][c].len;
				for (int j = col; j < len; j += 16) {
					float2 tt = sync_term(c, j);
					if (ffe != 0.0f) {
						float s, cc;
						sincos_fast(rps * (float)(pos + j), s, cc);
						tt = cmul(tt, make_float2(cc, s));
					}
					tr += tt.x;
					ti += tt.y;
				}
			}
		}
	}
	const float phr = wave_sum(tr), phi = wave_sum(ti);
	const float psi = atan2_fast(phi, phr);      // arg(phasor); |phasor| never matters
	if (dbg_stop == 5) return -100;

	// ---- soft symbols + soft bits (pi4cxpsk.c:442-503) ------------------------------
	constexpr int NSYM = NPL > 16 ? 8 : 4;       // 4 x 64 >= 234, 8 x 64 >= 468
	const float inv_dd = (float)(1 << nbits) / (2.0f * kPif);
	const int mask = (1 << nbits) - 1;
#pragma unroll
	for (int r = 0; r < NSYM; r++) {
		const int i = lane + 64 * r;
		if (i >= blen)
			continue;
		const int j = i * sps + d;
		const float2 x = pick(j);
		float th = atan2_fast(x.y, x.x) + reduce_2pi(fs * (float)j);
		th = reduce_2pi(fmaf(rps, (float)i, th) - psi);
		const float sv = (x.x == 0.0f && x.y == 0.0f) ? 0.0f : th * inv_dd;   // cargf(0) = 0
		if (g_ssyms)
			g_ssyms[i] = sv;
		const int ord = bt.ord_of_sym[i];
		if (ord >= 0) {
			const float svr2 = roundf(sv);
			const int sp = (int)svr2 & mask;
			const int ss = (svr2 > sv ? (sp - 1) : (sp + 1)) & mask;
			const int dq = (int)roundf((2.0f * fabsf(svr2 - sv)) * 64.0f);
			if (nbits == 2) {
				// symbol -> bits 0:00 1:01 2:11 3:10 (pi4cxpsk.c:95-100)
				const int p0 = sp >> 1, p1 = (sp ^ (sp >> 1)) & 1;
				const int s0 = ss >> 1, s1 = (ss ^ (ss >> 1)) & 1;
				const int v0 = 127 - ((p0 ^ s0) ? dq : (dq >> 1));
				const int v1 = 127 - ((p1 ^ s1) ? dq : (dq >> 1));
				const uint32_t pk2 = (uint32_t)(uint8_t)(int8_t)(p0 ? -v0 : v0) |
				                     ((uint32_t)(uint8_t)(int8_t)(p1 ? -v1 : v1) << 8);
				*reinterpret_cast<uint16_t *>(eb + 2 * ord) = (uint16_t)pk2;
			} else {
				const int p0 = sp & 1, s0 = ss & 1;
				const int v0 = 127 - ((p0 ^ s0) ? dq : (dq >> 1));
				eb[ord] = (int8_t)(p0 ? -v0 : v0);
			}
		}
	}
	WSYNC();
	return 0;
}

// ---------------------------------------------------------------------------
// branch metrics of one burst into bm[0..212): byte ov = cost of coded word ov
// (descramble + de-interleave folded into the gather via c_steps)
//   bcch.c:91-92 / ccch.c:95-96, interleave.c:73-87, scramb.c:63-73
// ---------------------------------------------------------------------------
template <bool ACC = false>
__device__ __forceinline__ void branch_metrics_k5_12(const int8_t *__restrict__ eb, int chain,
                                                     uint32_t *__restrict__ bm, int lane)
{
	const CostTable &ct = ACC ? c_cost_acc : c_cost;
	for (int k = lane; k < kSteps12; k += 64) {
		const uint32_t st = c_steps.w[chain][k];
		// the four byte sums c(a) + c(b) of a step come out of one add of two table words
		const uint32_t ia = (uint32_t)(uint8_t)eb[st & 0x3ffu] | ((st >> 2) & 0x100u);
		const uint32_t ib = (uint32_t)(uint8_t)eb[(st >> 16) & 0x3ffu] | ((st >> 18) & 0x100u);
		bm[k] = ct.a[ia] + ct.b[ib];
	}
}

// the four bursts of a fused wave at once: every lane owns steps lane + 64 it of each burst, and the
// three dependent fetches (step descriptor -> soft bits -> cost words) are each issued for all 16
// (burst, step) pairs before anything waits -- three memory round trips per wave instead of 48
template <bool ACC = false>
__device__ __forceinline__ void branch_metrics4_k5_12(const int8_t *__restrict__ eb, int eb_stride, int row_ok,
                                                      int row_chain, uint32_t *__restrict__ bm, int lane)
{
	const CostTable &ct = ACC ? c_cost_acc : c_cost;
	uint32_t st[2][4];
#pragma unroll
	for (int c = 0; c < 2; c++)
#pragma unroll
		for (int it = 0; it < 4; it++) {
			const int k = lane + 64 * it;
			st[c][it] = k < kSteps12 ? c_steps.w[c][k] : 0u;
		}
	uint32_t ia[4][4], ib[4][4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const bool ch = ((row_chain >> q) & 1) != 0;
		const int8_t *e = eb + q * eb_stride;
#pragma unroll
		for (int it = 0; it < 4; it++) {
			const uint32_t s = ch ? st[1][it] : st[0][it];
			ia[q][it] = (uint32_t)(uint8_t)e[s & 0x3ffu] | ((s >> 2) & 0x100u);
			ib[q][it] = (uint32_t)(uint8_t)e[(s >> 16) & 0x3ffu] | ((s >> 18) & 0x100u);
		}
	}
	uint32_t va[4][4], vb[4][4];
#pragma unroll
	for (int q = 0; q < 4; q++)
#pragma unroll
		for (int it = 0; it < 4; it++) {
			va[q][it] = ct.a[ia[q][it]];
			vb[q][it] = ct.b[ib[q][it]];
		}
#pragma unroll
	for (int q = 0; q < 4; q++) {
		const bool ok = ((row_ok >> q) & 1) != 0;
#pragma unroll
		for (int it = 0; it < 4; it++) {
			const int k = lane + 64 * it;
			if (k < kSteps12)
				bm[q * kSteps12 + k] = ok ? va[q][it] + vb[q][it] : 0u;
		}
	}
}

// ---------------------------------------------------------------------------
// 4 x (K=5, rate 1/2, 208 bits + flush) Viterbi, one burst per 16-lane row
//
// In-place butterfly: the two predecessors of a state always sit in two lanes of the row that
// differ by an xor mask, and the two successor states are written back to the same two lanes.
// The masks of the four phases are 8, 7, 2, 1 -- each ONE DPP control (row_ror:8,
// row_half_mirror, quad_perm), so the partner's metric arrives folded into the add.  With
// loc = c0*8 ^ c1*7 ^ c2*2 ^ c3*1, the predecessor state held by a lane in phase ph has bit i =
// c[(3 - i + ph) & 3]; after 4 steps the layout is back where it started.
//
// One 32-bit word per state carries everything the step needs:
//     [ path metric : 16 | decisions of the current 16-step window : 16 ]
// The metric never exceeds 212 * 252 = 53 424; unreachable states carry 0xF000 (libosmocore's
// MAX_AE plays the same role).  Before step j of a window the word of a lane that is the HIGH
// predecessor ((t >> 1) + 8) of its butterfly has bit j set (tb).  The two candidates of a new
// state are  own word + (cost << 16)  and  partner word + (cost << 16); v_min_u32 then (a) picks
// the smaller metric, (b) on equal metrics keeps the LOW predecessor (osmo_conv_decode: strict
// '>' on ascending states), and (c) leaves the decision in bit j of the winner's history --
// four VALU instructions per trellis step (add, add with DPP, min, add next tb).  The cost byte
// is fetched by every lane straight from the branch-metric words in LDS into the HIGH half of a
// register (ds_read_u8_d16_hi; with SRAM-ECC the low half reads back as zero, which is what the
// add wants), 8 steps ahead.
//
// The decision of step k is the oldest bit of the winning predecessor = input bit u[k-4].
// Windows start at k = 4 + 16 m, so window m's 16 decisions ARE the decoded bits
// u[16 m .. 16 m + 15], and its low 4 bits name the survivor's state at the start of the
// window: the "traceback" is 13 dependent 16-bit LDS reads per burst.
// ---------------------------------------------------------------------------
constexpr uint32_t kSentinel = 0xF0000000u;

// per row location: bits 0-7 own cost byte (2 bits per phase), 8-15 partner cost byte, 16-31 the
// 16-step tb pattern (bit j set when the lane holds a HIGH predecessor in phase j & 3)
struct DecTable { uint32_t v[16]; };
static constexpr uint32_t dec_out(uint32_t s, uint32_t b)
{
	const uint32_t reg = (s << 1) | b;
	uint32_t p0 = reg & 0x19u, p1 = reg & 0x17u;
	p0 ^= p0 >> 4; p0 ^= p0 >> 2; p0 ^= p0 >> 1;
	p1 ^= p1 >> 4; p1 ^= p1 >> 2; p1 ^= p1 >> 1;
	return ((p0 & 1u) << 1) | (p1 & 1u);
}
static constexpr DecTable make_dec()
{
	DecTable t{};
	for (uint32_t loc = 0; loc < 16; loc++) {
		// loc in the basis {8, 7, 2, 1}
		uint32_t c[4] = {0, 0, 0, 0};
		c[0] = (loc >> 3) & 1u;
		uint32_t x = loc & 7u;
		c[1] = (x >> 2) & 1u;
		x ^= c[1] ? 7u : 0u;
		c[2] = (x >> 1) & 1u;
		c[3] = x & 1u;
		uint32_t e = 0;
		for (int ph = 0; ph < 4; ph++) {
			uint32_t sp = 0;
			for (int i = 0; i < 4; i++)
				sp |= c[(3 - i + ph) & 3] << i;
			const uint32_t b = sp >> 3;
			e |= dec_out(sp, b) << (2 * ph);
			e |= dec_out(sp ^ 8u, b) << (8 + 2 * ph);
			for (int j = ph; j < 16; j += 4)
				e |= b << (16 + j);
		}
		t.v[loc] = e;
	}
	return t;
}
__constant__ DecTable c_dec = make_dec();

#define GMR1_DPP_PH0 "row_ror:8"
#define GMR1_DPP_PH1 "row_half_mirror"
#define GMR1_DPP_PH2 "quad_perm:[2,3,0,1]"
#define GMR1_DPP_PH3 "quad_perm:[1,0,3,2]"

#define ACS_CORE(PH)                                                                               \
	"s_waitcnt lgkmcnt(%[wt])\n\t"                                                                  \
	"v_add_u32 %[t1], %[w], %[r]\n\t"                                                               \
	"v_add_u32_dpp %[t2], %[w], %[q] " GMR1_DPP_PH##PH " row_mask:0xf bank_mask:0xf\n\t"            \
	"v_min_u32 %[w], %[t1], %[t2]\n\t"
// step with the operands of position J, prefetching the cost bytes of step J + 8; TN = tb of the next position
#define ACS_PF(J, PH, WAIT, TN)                                                                    \
	asm volatile(ACS_CORE(PH)                                                                      \
	             "v_add_u32 %[w], %[w], %[tn]\n\t"                                                  \
	             "ds_read_u8_d16_hi %[rn], %[ao] offset:%[off]\n\t"                                 \
	             "ds_read_u8_d16_hi %[qn], %[ap] offset:%[off]\n\t"                                 \
	             : [w] "+v"(w), [rn] "+v"(R[((J) + 8) & 15]), [qn] "+v"(Q[((J) + 8) & 15]),        \
	               [t1] "=&v"(t1), [t2] "=&v"(t2)                                                    \
	             : [r] "v"(R[J]), [q] "v"(Q[J]), [ao] "v"(ao[PH]), [ap] "v"(ap[PH]), [tn] "v"(TN),   \
	               [off] "i"(4 * ((J) + 8)), [wt] "i"(WAIT))
// step without prefetch
#define ACS_NP(J, PH, WAIT, TN)                                                                    \
	asm volatile(ACS_CORE(PH)                                                                      \
	             "v_add_u32 %[w], %[w], %[tn]\n\t"                                                  \
	             : [w] "+v"(w), [t1] "=&v"(t1), [t2] "=&v"(t2)                                       \
	             : [r] "v"(R[J]), [q] "v"(Q[J]), [tn] "v"(TN), [wt] "i"(WAIT))
// last step of a window: the caller clears the decisions and sets the first tb itself
#define ACS_PF_END(J, PH, WAIT)                                                                    \
	asm volatile(ACS_CORE(PH)                                                                      \
	             "ds_read_u8_d16_hi %[rn], %[ao] offset:%[off]\n\t"                                 \
	             "ds_read_u8_d16_hi %[qn], %[ap] offset:%[off]\n\t"                                 \
	             : [w] "+v"(w), [rn] "+v"(R[((J) + 8) & 15]), [qn] "+v"(Q[((J) + 8) & 15]),        \
	               [t1] "=&v"(t1), [t2] "=&v"(t2)                                                    \
	             : [r] "v"(R[J]), [q] "v"(Q[J]), [ao] "v"(ao[PH]), [ap] "v"(ap[PH]),                 \
	               [off] "i"(4 * ((J) + 8)), [wt] "i"(WAIT))
#define ACS_NP_END(J, PH, WAIT)                                                                    \
	asm volatile(ACS_CORE(PH)                                                                      \
	             : [w] "+v"(w), [t1] "=&v"(t1), [t2] "=&v"(t2)                                       \
	             : [r] "v"(R[J]), [q] "v"(Q[J]), [wt] "i"(WAIT))
// cost bytes of step K (relative to the address registers) into the operands of position J
#define ACS_LOAD(J, PH, K)                                                                         \
	asm volatile("ds_read_u8_d16_hi %[rn], %[ao] offset:%[off]\n\t"                                 \
	             "ds_read_u8_d16_hi %[qn], %[ap] offset:%[off]\n\t"                                 \
	             : [rn] "+v"(R[J]), [qn] "+v"(Q[J])                                                  \
	             : [ao] "v"(ao[PH]), [ap] "v"(ap[PH]), [off] "i"(4 * (K)))

struct DecPre;
__device__ __forceinline__ void k5_12_survivors_crc(uint64_t *__restrict__ surv, uint32_t *__restrict__ ubits, int lane,
                                                    uint32_t &syn_o, const DecPre *dp = nullptr);
__device__ __forceinline__ void k5_12_survivors_crc_lat(uint64_t *__restrict__ surv, uint32_t *__restrict__ ubits, int lane,
                                                        uint32_t &syn_o, const DecPre *dp);

// bm: 4 rows x 212 words; surv: 13 x 64 halfwords of window decisions; ubits: 4 rows x 8 words
// (decoded bits, LSB first)
//
// ACC = libosmocore's accelerated decoder instead of its generic one (decision D1b, oracle/orc_3p_acc.c; costs from
// c_cost_acc): every start state is allowed, state 0 leading by 127 * N * K; the four flush steps are ordinary
// butterflies (the survivor walk still starts in state 0); no path metric is returned.  Ties between the two paths into
// a state fall to the same (lower) predecessor in both decoders.
struct DecPre {                                    // the decoder's per-lane constants, when the caller keeps them (receive loop)
	uint32_t dc;
	uint4 sy0, sy1;
#ifdef GMR1_HIP_PROFILE
	unsigned long long *stamp = nullptr;
#endif
};
#ifdef GMR1_HIP_PROFILE
#define GMR1_DSTAMP(dp, k, lane)                                             \
	do {                                                                    \
		if ((dp)->stamp && (lane) == 0)                                     \
			(dp)->stamp[k] = __builtin_readcyclecounter();                  \
	} while (0)
#else
#define GMR1_DSTAMP(dp, k, lane) do { } while (0)
#endif

template <bool ACC = false>
__device__ void decode4_k5_12(const uint32_t *__restrict__ bm, uint64_t *__restrict__ surv,
                              uint32_t *__restrict__ ubits, int lane, uint32_t &syn_o, uint32_t &final_ae,
                              const DecPre *dp = nullptr)
{
	typedef __attribute__((address_space(3))) const unsigned char lds_cbyte;
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	// per-location constants (c_dec): cost byte of the own / partner transition per phase, tb pattern
	const uint32_t dc = dp ? dp->dc : c_dec.v[loc];
	const uint32_t row_base = (uint32_t)(uintptr_t)(lds_cbyte *)(bm + row * kSteps12);
	uint32_t ao[4], ap[4];      // LDS byte address of this lane's own / partner cost in step 0 of the phase
	bool hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		ao[ph] = row_base + ((dc >> (2 * ph)) & 3u);
		ap[ph] = row_base + ((dc >> (8 + 2 * ph)) & 3u);
		hi[ph] = ((dc >> (16 + ph)) & 1u) != 0;
	}
	// cost << 16 of the own / partner transition, per window position (low halves stay zero whether
	// or not the d16 load preserves them)
	uint32_t R[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, Q[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	uint32_t T[16];             // tie-break / decision bit of the position: set in HIGH-predecessor lanes
#pragma unroll
	for (int j = 0; j < 16; j++)
		T[j] = (dc >> 16) & (1u << j);
	uint32_t w = (loc ? (ACC ? kAccLeadK5r2 << 16 : kSentinel) : 0u) | T[0];
	uint32_t t1, t2;
	uint16_t *dump = reinterpret_cast<uint16_t *>(surv) + lane;

	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	// steps 0..3: the decisions are u[-4..-1], dropped
	ACS_LOAD(0, 0, 0); ACS_LOAD(1, 1, 1); ACS_LOAD(2, 2, 2); ACS_LOAD(3, 3, 3);
	ACS_NP(0, 0, 6, T[1]); ACS_NP(1, 1, 4, T[2]); ACS_NP(2, 2, 2, T[3]); ACS_NP_END(3, 3, 0);
	w = (w & 0xffff0000u) | T[0];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		ao[ph] += 16;
		ap[ph] += 16;
	}
	// window pipeline: the costs of 8 steps are always in flight
	ACS_LOAD(0, 0, 0); ACS_LOAD(1, 1, 1); ACS_LOAD(2, 2, 2); ACS_LOAD(3, 3, 3);
	ACS_LOAD(4, 0, 4); ACS_LOAD(5, 1, 5); ACS_LOAD(6, 2, 6); ACS_LOAD(7, 3, 7);
#pragma unroll 1
	for (int m = 0; m < 12; m++) {
		ACS_PF(0, 0, 14, T[1]); ACS_PF(1, 1, 14, T[2]); ACS_PF(2, 2, 14, T[3]); ACS_PF(3, 3, 14, T[4]);
		ACS_PF(4, 0, 14, T[5]); ACS_PF(5, 1, 14, T[6]); ACS_PF(6, 2, 14, T[7]); ACS_PF(7, 3, 14, T[8]);
		ACS_PF(8, 0, 14, T[9]); ACS_PF(9, 1, 14, T[10]); ACS_PF(10, 2, 14, T[11]); ACS_PF(11, 3, 14, T[12]);
		ACS_PF(12, 0, 14, T[13]); ACS_PF(13, 1, 14, T[14]); ACS_PF(14, 2, 14, T[15]); ACS_PF_END(15, 3, 14);
		dump[m * 64] = (uint16_t)w;
		w = (w & 0xffff0000u) | T[0];
#pragma unroll
		for (int ph = 0; ph < 4; ph++) {
			ao[ph] += 64;
			ap[ph] += 64;
		}
	}
	// window 12: steps 196..211, the last four are the flush (b = 0 transitions only: the lanes
	// whose new state ends in 1 become unreachable)
	ACS_PF(0, 0, 14, T[1]); ACS_PF(1, 1, 14, T[2]); ACS_PF(2, 2, 14, T[3]); ACS_PF(3, 3, 14, T[4]);
	ACS_PF(4, 0, 14, T[5]); ACS_PF(5, 1, 14, T[6]); ACS_PF(6, 2, 14, T[7]); ACS_PF(7, 3, 14, T[8]);
	ACS_NP(8, 0, 14, T[9]); ACS_NP(9, 1, 12, T[10]); ACS_NP(10, 2, 10, T[11]); ACS_NP(11, 3, 8, T[12]);
	if constexpr (ACC) {
		ACS_NP(12, 0, 6, T[13]); ACS_NP(13, 1, 4, T[14]); ACS_NP(14, 2, 2, T[15]); ACS_NP_END(15, 3, 0);
		(void)hi;
	} else {
		ACS_NP_END(12, 0, 6);
		w = hi[0] ? kSentinel : (w + T[13]);
		ACS_NP_END(13, 1, 4);
		w = hi[1] ? kSentinel : (w + T[14]);
		ACS_NP_END(14, 2, 2);
		w = hi[2] ? kSentinel : (w + T[15]);
		ACS_NP_END(15, 3, 0);
		w = hi[3] ? kSentinel : w;
	}
	dump[12 * 64] = (uint16_t)w;
	// state 0 ends in location 0 of the row; osmo_conv_decode_acc returns 0, not a metric
	final_ae = ACC ? 0u : w >> 16;
	k5_12_survivors_crc(surv, ubits, lane, syn_o, dp);
}

// The K=5 rate-1/2 decoder shaped for the LATENCY of one burst (the receive loop: a wave alone on its SIMD issues one
// instruction every four to five cycles whatever the dependences, so a round costs what its instruction count says).
// The batch decoder above spends 7 instructions per trellis step (wait, add, add-dpp, min, add of the next tie-break bit, two
// cost-byte reads).  Here a step's operands come ready-made from a table the branch-metric phase expands once per burst:
// per step and code word two 8-byte entries (this lane the HIGH predecessor or not) -- both generators have the D^0 and D^4
// taps, so the partner's code word is the own one's complement -- holding  own cost << 16 | tie-break bit if this lane is the
// HIGH predecessor,  partner's cost << 16 | tie-break bit if the partner is.  One ds_read_b64 and three VALU per step: 5
// instructions.  212 x 64 B of LDS, which only the loop (one burst per work-group) can afford.  Arithmetic, ties and
// decisions are the batch decoder's.
static constexpr bool dec_partner_is_complement()
{
	const DecTable t = make_dec();
	for (int loc = 0; loc < 16; loc++)
		for (int ph = 0; ph < 4; ph++)
			if (((t.v[loc] >> (8 + 2 * ph)) & 3u) != (((t.v[loc] >> (2 * ph)) & 3u) ^ 3u))
				return false;
	return true;
}
static_assert(dec_partner_is_complement(), "the partner transition's code word must be the own one's complement");
typedef uint32_t lat_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kLatTabBytes = kSteps12 * 64;
#define ACSL_CORE(PH)                                                                              \
	".if %[wt] >= 0\n\t"                                                                            \
	"s_waitcnt lgkmcnt(%[wt])\n\t"                                                                  \
	".endif\n\t"                                                                                    \
	"v_add_u32 %[t1], %[w], %[r]\n\t"                                                               \
	"v_add_u32_dpp %[t2], %[w], %[q] " GMR1_DPP_PH##PH " row_mask:0xf bank_mask:0xf\n\t"            \
	"v_min_u32 %[w], %[t1], %[t2]\n\t"
#define ACSL_PF(J, PH, WAIT)                                                                       \
	asm volatile(ACSL_CORE(PH)                                                                     \
	             "ds_read_b64 %[rqn], %[a] offset:%[off]\n\t"                                       \
	             : [w] "+v"(w), [rqn] "=v"(RQ[((J) + 8) & 15]), [t1] "=&v"(t1), [t2] "=&v"(t2)        \
	             : [r] "v"(RQ[J].x), [q] "v"(RQ[J].y), [a] "v"(A[PH]), [off] "i"(16 * ((J) + 8)), [wt] "i"(WAIT))
#define ACSL_NP(J, PH, WAIT)                                                                       \
	asm volatile(ACSL_CORE(PH)                                                                     \
	             : [w] "+v"(w), [t1] "=&v"(t1), [t2] "=&v"(t2)                                       \
	             : [r] "v"(RQ[J].x), [q] "v"(RQ[J].y), [wt] "i"(WAIT))
#define ACSL_LOAD(J, PH, K)                                                                        \
	asm volatile("ds_read_b64 %[rqn], %[a] offset:%[off]\n\t" : [rqn] "=v"(RQ[J]) : [a] "v"(A[PH]), [off] "i"(16 * (K)))

// one step's eight entries from its cost word (byte j = cost of code word j): cls = 2 * code word + HIGH
__device__ __forceinline__ void lat_expand_step(uint32_t *__restrict__ tab, int k, uint32_t word)
{
	const uint32_t bit = 1u << (k < 4 ? k : ((k - 4) & 15));
	const uint32_t c0 = (word << 16) & 0x00ff0000u, c1 = (word << 8) & 0x00ff0000u, c2 = word & 0x00ff0000u,
	               c3 = (word >> 8) & 0x00ff0000u;
	// code-word-major ([j][step], 16 bytes each: not HIGH {c_j, c_(3-j) | bit}, HIGH {c_j | bit, c_(3-j)}): lanes own
	// consecutive steps, so a wave's 16-byte writes are consecutive in LDS (step-major they were 64 bytes apart: eight-way
	// bank conflicts, 1 400 cycles)
	uint4 *d = reinterpret_cast<uint4 *>(tab) + k;
	d[0 * kSteps12] = make_uint4(c0, c3 | bit, c0 | bit, c3);
	d[1 * kSteps12] = make_uint4(c1, c2 | bit, c1 | bit, c2);
	d[2 * kSteps12] = make_uint4(c2, c1 | bit, c2 | bit, c1);
	d[3 * kSteps12] = make_uint4(c3, c0 | bit, c3 | bit, c0);
}

// TAIL = false: the forward pass alone -- window words to `surv`, the final metric returned; the survivor walk and the CRC
// (k5_12_survivors_crc_lat) are the caller's, on another wave (k_rx_chain_pipe)
template <bool ACC = false, bool TAIL = true>
__device__ void decode1_k5_12_lat(const uint32_t *__restrict__ tab, uint64_t *__restrict__ surv,
                                  uint32_t *__restrict__ ubits, int lane, uint32_t &syn_o, uint32_t &final_ae,
                                  const DecPre *dp)
{
	typedef __attribute__((address_space(3))) const unsigned char lds_cbyte;
	const uint32_t loc = (uint32_t)lane & 15u;
	const uint32_t dc = dp->dc;
	const uint32_t base = (uint32_t)(uintptr_t)(lds_cbyte *)tab;
	uint32_t A[4];              // LDS byte address of this lane's entry in step 0 of the phase
	bool hi[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		hi[ph] = ((dc >> (16 + ph)) & 1u) != 0;
		A[ph] = base + (uint32_t)(kSteps12 * 16) * ((dc >> (2 * ph)) & 3u) + (hi[ph] ? 8u : 0u);
	}
	lat_u32x2 RQ[16];
	uint32_t w = loc ? (ACC ? kAccLeadK5r2 << 16 : kSentinel) : 0u;
	uint32_t t1, t2;
	uint16_t *dump = reinterpret_cast<uint16_t *>(surv) + lane;

	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	// steps 0..3: the decisions are u[-4..-1], dropped
	ACSL_LOAD(0, 0, 0); ACSL_LOAD(1, 1, 1); ACSL_LOAD(2, 2, 2); ACSL_LOAD(3, 3, 3);
	ACSL_NP(0, 0, 3); ACSL_NP(1, 1, 2); ACSL_NP(2, 2, 1); ACSL_NP(3, 3, 0);
	w &= 0xffff0000u;
#pragma unroll
	for (int ph = 0; ph < 4; ph++)
		A[ph] += 4 * 16;
	// window pipeline: the operands of 8 steps are always in flight
	ACSL_LOAD(0, 0, 0); ACSL_LOAD(1, 1, 1); ACSL_LOAD(2, 2, 2); ACSL_LOAD(3, 3, 3);
	ACSL_LOAD(4, 0, 4); ACSL_LOAD(5, 1, 5); ACSL_LOAD(6, 2, 6); ACSL_LOAD(7, 3, 7);
#pragma unroll 1
	for (int m = 0; m < 12; m++) {
		// (one wait per four steps: eight loads are in flight, the four oldest must have landed)
		ACSL_PF(0, 0, 4); ACSL_PF(1, 1, -1); ACSL_PF(2, 2, -1); ACSL_PF(3, 3, -1);
		ACSL_PF(4, 0, 4); ACSL_PF(5, 1, -1); ACSL_PF(6, 2, -1); ACSL_PF(7, 3, -1);
		ACSL_PF(8, 0, 4); ACSL_PF(9, 1, -1); ACSL_PF(10, 2, -1); ACSL_PF(11, 3, -1);
		ACSL_PF(12, 0, 4); ACSL_PF(13, 1, -1); ACSL_PF(14, 2, -1); ACSL_PF(15, 3, -1);
		dump[m * 64] = (uint16_t)w;
		w &= 0xffff0000u;
#pragma unroll
		for (int ph = 0; ph < 4; ph++)
			A[ph] += 16 * 16;
	}
	// window 12: steps 196..211, the last four are the flush (generic decoder: b = 0 transitions only -- the lanes whose new
	// state ends in 1 become unreachable)
	ACSL_PF(0, 0, 4); ACSL_PF(1, 1, -1); ACSL_PF(2, 2, -1); ACSL_PF(3, 3, -1);
	ACSL_PF(4, 0, 4); ACSL_PF(5, 1, -1); ACSL_PF(6, 2, -1); ACSL_PF(7, 3, -1);
	ACSL_NP(8, 0, 4); ACSL_NP(9, 1, -1); ACSL_NP(10, 2, -1); ACSL_NP(11, 3, -1);
	if constexpr (ACC) {
		ACSL_NP(12, 0, 0); ACSL_NP(13, 1, -1); ACSL_NP(14, 2, -1); ACSL_NP(15, 3, -1);
		(void)hi;
	} else {
		ACSL_NP(12, 0, 0);
		w = hi[0] ? kSentinel : w;
		ACSL_NP(13, 1, -1);
		w = hi[1] ? kSentinel : w;
		ACSL_NP(14, 2, -1);
		w = hi[2] ? kSentinel : w;
		ACSL_NP(15, 3, -1);
		w = hi[3] ? kSentinel : w;
	}
	dump[12 * 64] = (uint16_t)w;
	final_ae = ACC ? 0u : w >> 16;
	GMR1_DSTAMP(dp, 12, lane);
	if constexpr (TAIL)
		k5_12_survivors_crc_lat(surv, ubits, lane, syn_o, dp);
	else
		syn_o = 0;
}

// Tail of the decoder shaped for the LATENCY of one burst (the receive loop: one burst per wave, nothing to overlap with):
// the 13 window words of every location are read at once and the survivor chain is walked with v_readlane on scalars --
// 13 dependent LDS round trips become one.  Row 0 only; the CRC as in k5_12_survivors_crc.
__device__ __forceinline__ void k5_12_survivors_crc_lat(uint64_t *__restrict__ surv, uint32_t *__restrict__ ubits, int lane,
                                                        uint32_t &syn_o, const DecPre *dp)
{
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	WSYNC();
	// survivor chain of row 0: every location's 13 window words at once, then the walk on scalars
	{
		constexpr unsigned long long kLocOf =
			0x0ull | (0x8ull << 4) | (0x7ull << 8) | (0xFull << 12) | (0x2ull << 16) | (0xAull << 20) |
			(0x5ull << 24) | (0xDull << 28) | (0x1ull << 32) | (0x9ull << 36) | (0x6ull << 40) |
			(0xEull << 44) | (0x3ull << 48) | (0xBull << 52) | (0x4ull << 56) | (0xCull << 60);
		const uint16_t *d16 = reinterpret_cast<const uint16_t *>(surv) + loc;      // (rows 1-3 read row 0's words too)
		uint32_t H[13];
#pragma unroll
		for (int m = 0; m < 13; m++)
			H[m] = d16[m * 64];
		uint32_t L = 0, hv[13];
#pragma unroll
		for (int m = 12; m >= 0; m--) {
			hv[m] = (uint32_t)__builtin_amdgcn_readlane((int)H[m], (int)L);
			L = (uint32_t)(kLocOf >> (4 * (hv[m] & 15u))) & 15u;
		}
		if (lane == 0) {
#pragma unroll
			for (int m = 0; m < 13; m += 2)
				ubits[m >> 1] = hv[m] | (m == 12 ? 0u : (hv[m + 1] << 16));
		}
	}
	WSYNC();
	GMR1_DSTAMP(dp, 13, lane);
	// CRC16 over the 208 decoded bits, 13 bits per lane of the row, XOR-reduced with DPP (as in k5_12_survivors_crc)
	uint32_t syn = 0;
	{
		const uint32_t *ub = ubits + row * 8;
		const uint32_t k0 = loc * 13u;
		const uint32_t lo = ub[k0 >> 5], hi2 = ub[(k0 >> 5) + 1];
		const uint32_t cbits = __builtin_amdgcn_alignbit(hi2, lo, k0 & 31u);
		const uint32_t sy[7] = {dp->sy0.x, dp->sy0.y, dp->sy0.z, dp->sy0.w, dp->sy1.x, dp->sy1.y, dp->sy1.z};
		uint32_t acc = 0;
#pragma unroll
		for (int pq = 0; pq < 7; pq++) {
			const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)cbits, 2 * pq, 1);
			const uint32_t m1 = pq < 6 ? (uint32_t)__builtin_amdgcn_sbfe((int)cbits, 2 * pq + 1, 1) : 0u;
			acc ^= sy[pq] & ((m0 & 0xffffu) | (m1 & 0xffff0000u));
		}
		syn = (acc ^ (acc >> 16)) & 0xffffu;
		syn ^= row_xor<1>(syn);
		syn ^= row_xor<2>(syn);
		syn ^= row_xor<4>(syn);
		syn ^= row_xor<8>(syn);
	}
	syn_o = syn;
}

// second half of the decoder: survivor chain and CRC16 of the four rows (shared with the 16-bit-lane forward pass below)
__device__ __forceinline__ void k5_12_survivors_crc(uint64_t *__restrict__ surv, uint32_t *__restrict__ ubits, int lane,
                                                    uint32_t &syn_o, const DecPre *dp)
{
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	// this lane's CRC syndrome words travel while the survivor chain is walked
	const uint4 sy0 = dp ? dp->sy0 : *reinterpret_cast<const uint4 *>(&c_syn_rows.w[loc][0]);
	const uint4 sy1 = dp ? dp->sy1 : *reinterpret_cast<const uint4 *>(&c_syn_rows.w[loc][4]);
	WSYNC();

	// survivor chain, one lane per row: window m's decisions at the survivor's location are the
	// decoded bits u[16 m ..]; their low nibble (u[16m-4 .. 16m-1] seen from window m) is the state
	// at the start of the window, bit-reversed: h0 -> state bit 3 -> basis vector 8, h1 -> 7,
	// h2 -> 2, h3 -> 1  (osmo_conv_decode_get_output, end state 0 after flush)
	if (loc == 0) {
		const uint16_t *d16 = reinterpret_cast<const uint16_t *>(surv) + row * 16;
		// location of the state whose reversed nibble is x, x = 0..15
		constexpr unsigned long long kLocOf =
			0x0ull | (0x8ull << 4) | (0x7ull << 8) | (0xFull << 12) | (0x2ull << 16) | (0xAull << 20) |
			(0x5ull << 24) | (0xDull << 28) | (0x1ull << 32) | (0x9ull << 36) | (0x6ull << 40) |
			(0xEull << 44) | (0x3ull << 48) | (0xBull << 52) | (0x4ull << 56) | (0xCull << 60);
		uint32_t L = 0;
		uint32_t prev = 0;
#pragma unroll
		for (int m = 12; m >= 0; m--) {
			const uint32_t h = d16[m * 64 + L];
			L = (uint32_t)(kLocOf >> (4 * (h & 15u))) & 15u;
			if (m & 1)
				prev = h;
			else
				ubits[row * 8 + (m >> 1)] = h | (m == 12 ? 0u : (prev << 16));
		}
	}
	WSYNC();

	// CRC16 over the 208 decoded bits, 13 bits per lane of the row, XOR-reduced with DPP
	uint32_t syn = 0;
	{
		const uint32_t *ub = ubits + row * 8;
		const uint32_t k0 = loc * 13u;
		const uint32_t lo = ub[k0 >> 5], hi2 = ub[(k0 >> 5) + 1];      // word 7 of a row is never a data word
		const uint32_t cbits = __builtin_amdgcn_alignbit(hi2, lo, k0 & 31u);
		const uint32_t sy[7] = {sy0.x, sy0.y, sy0.z, sy0.w, sy1.x, sy1.y, sy1.z};
		uint32_t acc = 0;
#pragma unroll
		for (int pq = 0; pq < 7; pq++) {
			const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)cbits, 2 * pq, 1);
			const uint32_t m1 = pq < 6 ? (uint32_t)__builtin_amdgcn_sbfe((int)cbits, 2 * pq + 1, 1) : 0u;
			acc ^= sy[pq] & ((m0 & 0xffffu) | (m1 & 0xffff0000u));
		}
		syn = (acc ^ (acc >> 16)) & 0xffffu;
		syn ^= row_xor<1>(syn);
		syn ^= row_xor<2>(syn);
		syn ^= row_xor<4>(syn);
		syn ^= row_xor<8>(syn);
	}
	syn_o = syn;
}

__device__ __forceinline__ void store_l2(uint8_t *l2, const uint32_t *ub)
{
	uint32_t *l2w = reinterpret_cast<uint32_t *>(l2);
#pragma unroll
	for (int i = 0; i < 6; i++)
		l2w[i] = ub[i];
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int NPL, int SPS, bool DECODE, bool ACC = false>
__global__ __launch_bounds__(64) void k_rx(RxArgs a, int max_in_len, int max_len)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const Lds L = lds_carve(lds_raw, max_in_len, max_len, DECODE);

	constexpr int PER = DECODE ? 4 : 1;
	int g0 = blockIdx.x * PER;
	int n_end = a.n;
	if (DECODE && a.seg_count) {
		// the receive loop's CCCH lists (see k_rx4): segments with unused slots, one time slice of each per launch
		int sg = g0 / a.seg_stride;
		int lo = 0;
		if (a.seg_first) {
			if (a.seg_groups > 0) {
				sg = (int)blockIdx.x / a.seg_groups;
				lo = (a.seg_first[sg] + 3) & ~3;
				g0 = sg * a.seg_stride + lo + ((int)blockIdx.x % a.seg_groups) * PER;
			} else {
				lo = (a.seg_first[sg] + 3) & ~3;
			}
		}
		const int base = sg * a.seg_stride;
		n_end = min(min(a.n, g0 + PER), base + min(a.seg_count[sg], a.seg_stride));
		if (g0 >= n_end || g0 < base + lo)
			return;
	}
	int row_ok = 0;       // bit q: burst q of this wave demodulated fine
	int row_chain = 0;    // bit q: burst q is CCCH

	for (int q = 0; q < PER; q++) {
		const int g = g0 + q;
		if (g >= n_end)
			break;
		int type, in_len;
		if (DECODE) {
			const int kind = a.kind[g] ? 1 : 0;
			type = kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH;
			in_len = a.in_len[kind];
			row_chain |= kind << q;
		} else {
			type = a.fixed_type;
			in_len = a.in_len[0];
		}
		type = __builtin_amdgcn_readfirstlane(type);
		in_len = __builtin_amdgcn_readfirstlane(in_len);
		const float fsh = a.freq_shift ? a.freq_shift[g] : 0.0f;
		int sid = -1;
		float toa = 0.f, fe = 0.f;
		float *gss = a.ssyms ? a.ssyms + (size_t)g * a.ssyms_stride : nullptr;
		int8_t *eb = L.eb + (DECODE ? q * kEbRow : 0);

		WSYNC();
		const int rv = demod_one<NPL, SPS>(type, a.iq + a.offset[g], in_len, a.sps, fsh, L, eb, lane,
		                                   a.dbg_stop, sid, toa, fe, gss);
		if (rv == -100)
			continue;    // profiling build-out: phase cut-off
		if (a.energy) {
			const float e = window_energy<NPL>(a.iq + a.offset[g], in_len, lane);
			if (lane == 0)
				a.energy[g] = e;
		}

		if (lane == 0) {
			a.rv[g] = rv;
			if (a.sync_id) a.sync_id[g] = sid;
			if (a.toa) a.toa[g] = rv ? 0.f : toa;
			if (a.freq_err) a.freq_err[g] = rv ? 0.f : fe;
		}
		if (a.ebits) {
			const int neb = c_types[type].ebits;
			int8_t *ge = a.ebits + (size_t)g * a.ebits_stride;
			for (int i = lane; i < a.ebits_stride; i += 64)
				ge[i] = (rv == 0 && i < neb) ? eb[i] : (int8_t)0;
		}
		if (rv && gss)
			for (int i = lane; i < c_types[type].len; i += 64)
				gss[i] = 0.f;
		if (rv == 0)
			row_ok |= 1 << q;
	}

	if (DECODE) {
		if (a.dbg_stop && a.dbg_stop < 7)
			return;
		WSYNC();     // x is dead from here on: bm / surv / ubits overlay it
		for (int q = 0; q < 4; q++) {
			if ((row_ok >> q) & 1) {
				branch_metrics_k5_12<ACC>(L.eb + q * kEbRow, (row_chain >> q) & 1, L.bm + q * kSteps12, lane);
			} else {
				for (int k = lane; k < kSteps12; k += 64)
					L.bm[q * kSteps12 + k] = 0;
			}
		}
		WSYNC();
		if (a.dbg_stop == 7)
			return;
		uint32_t syn, fae;
		decode4_k5_12<ACC>(L.bm, L.surv, L.ubits, lane, syn, fae);
		const int row = lane >> 4;
		const int g = g0 + row;
		if ((lane & 15) == 0 && g < n_end) {
			if ((row_ok >> row) & 1) {
				store_l2(a.l2 + (size_t)g * 24, L.ubits + row * 8);
				a.crc[g] = syn ? 1 : 0;
				a.conv[g] = (int32_t)fae;
			} else {
				uint32_t *l2w = reinterpret_cast<uint32_t *>(a.l2 + (size_t)g * 24);
#pragma unroll
				for (int i = 0; i < 6; i++)
					l2w[i] = 0;
				a.crc[g] = -1;
				a.conv[g] = 0;
			}
		}
	}
}

// ---------------------------------------------------------------------------
// k_rx4 -- fused BCCH / CCCH receive, four bursts per wavefront, with the serial phases
// (timing bisection, sync-symbol arithmetic) done ONCE for the four bursts, one burst per
// 16-lane row, instead of once per burst with most lanes idle:
//
//   pass 1, per burst : load -> statistics in registers, sync-chunk windows -> LDS (normalised),
//                       sync correlation                              -> corr[q][.]
//   rows              : peak window argmax, early/late bisection (21 taps, early point on lanes
//                       0-7 and late point on lanes 8-15 of the row), interpolated peak power
//                                                                     -> toa, rv per row
//   rows              : sync symbols re-read from L2 / Infinity Cache, chunk sums,
//                       fine frequency error, carrier phase           -> ffe, psi per row
//   pass 2, per burst : 234 symbols re-read (stride sps), soft symbols from the phase in turns,
//                       soft bits by Gray-boundary arithmetic
//   rows              : branch metrics (table words, batched fetches), Viterbi, survivor walk,
//                       CRC (decode4_k5_12)
//
// The second read of a burst happens a few microseconds after the first and is served by the
// L2 / Infinity Cache; it buys back ~650 VALU instructions per burst.
// Single-sequence burst formats only (BCCH, DC6), which is all the fused path handles.
// ---------------------------------------------------------------------------
// rotated reference value of sync symbol n of a format's first training sequence: conj(ref_n) e^{j fs (n' sps)},
// n' = position of the symbol inside its chunk (pi4cxpsk.c:125-171 + the derotation of :539 folded in)
__device__ __forceinline__ float2 sync_coef0(const DevBurst &bt, int n, int sps, float fs)
{
	const int nch = bt.n_chunks[0];
	int ch = 0, base = 0, cum = 0;
	for (int c = 0; c < nch - 1; c++) {
		cum += bt.sync[0][c].len;
		if (n >= cum) { base = cum; ch = c + 1; }
	}
	const int nn = n - base;
	float s, c;
	sincos_fast(fs * (float)(nn * sps), s, c);
	return conj_ref_mul(bt.nbits, bt.sync[0][ch].syms[nn], make_float2(c, s));
}

// the same for symbol n of training sequence sq of a format whose sequences have ONE chunk each (NT3 FACCH)
__device__ __forceinline__ float2 sync_coef_seq1(const DevBurst &bt, int sq, int n, int sps, float fs)
{
	float s, c;
	sincos_fast(fs * (float)(n * sps), s, c);
	return conj_ref_mul(bt.nbits, bt.sync[sq][0].syms[n], make_float2(c, s));
}

// the same for freq_shift = 0, every burst format, sps 1..16: [sps][type][n < 32]
constexpr int kCoef0MaxSps = 16;
__device__ float2 g_coef0[kCoef0MaxSps + 1][kNumTypes][32];
// ... and a copy in the constant address space (made on the device after every k_coef0 run), which a wave-uniform index
// reads with scalar loads -- from the writable array above the compiler issues one vector load per value
__constant__ float2 c_coef0[kCoef0MaxSps + 1][kNumTypes][32];

__global__ __launch_bounds__(64) void k_coef0(int first, int count)
{
	const int type = first + (int)blockIdx.x, sps = (int)blockIdx.y + 1, n = (int)threadIdx.x;
	if (type >= first + count || n >= 32)
		return;
	const DevBurst &bt = c_types[type];
	float2 v = make_float2(0.f, 0.f);
	if (bt.n_sync > 0 && n < bt.sync_tl[0]) {
		const float fs = (0.0f - bt.rotation) / (float)sps;               // pi4cxpsk.c:539 with freq_shift = 0
		v = sync_coef0(bt, n, sps, fs);
	}
	g_coef0[sps][type][n] = v;
}

hipError_t upload_types(const DevBurst *host, int first, int count, hipStream_t stream)
{
	hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_types), host, sizeof(DevBurst) * (size_t)count,
	                                      sizeof(DevBurst) * (size_t)first, hipMemcpyHostToDevice, stream);
	if (e != hipSuccess)
		return e;
	// the zero-shift sync references of the uploaded formats (same stream: ordered before any burst kernel)
	hipLaunchKernelGGL(k_coef0, dim3((unsigned)count, kCoef0MaxSps), dim3(64), 0, stream, first, count);
	e = hipGetLastError();
	if (e != hipSuccess)
		return e;
	void *src = nullptr;
	e = hipGetSymbolAddress(&src, HIP_SYMBOL(g_coef0));
	if (e != hipSuccess)
		return e;
	return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_coef0), src, sizeof(g_coef0), 0, hipMemcpyDeviceToDevice, stream);
}

// Sync correlation of the fused path's two formats with everything static: chunks of T0, T1, T2 training symbols
// (BCCH 11 + 3 + 3, DC6 7 + 3 + 3; nb.c:36-41, 94-99).  The rotated reference values sit in scalar registers, one pair
// (re, im) per tap -- read straight from the table with scalar loads when no frequency shift was given (`ctab`), else
// out of lane n of `cfl` (v_readlane) -- and feed the packed FMAs as scalar operands, so a tap costs one LDS read and two
// v_pk_fma_f32, all reads of a lag issued back to back.  xs: the staged chunk windows, window c = samples
// [pos_c sps, pos_c sps + T_c sps + w - 1).  corr[j] = sum over chunks of |sum_n c_n x[j + n sps]|.
//
// (ar, ai) += c x as two packed FMAs: (-c.im x.im, c.im x.re) first, then c.re (x.re, x.im) -- the order of the scalar
// chains ar = fma(c.re, x.re, fma(-c.im, x.im, ar)), ai = fma(c.re, x.im, fma(c.im, x.re, ai)).  Written out with the
// operand selects and the sign on the ONE pair: the compiler builds (-c.im, c.im) and (c.re, c.re) as pairs of their own,
// four scalar registers a tap, and spills what they displace.  (s_nop: a packed result needs one wait state before its
// next use, which the compiler's own sequences carry as well.)
__device__ __forceinline__ void pk_cmac(v2f &acc, unsigned long long c, v2f x)
{
	asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\ts_nop 0\n\t"
	    "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\ts_nop 0"
	    : "+v"(acc)
	    : "s"(c), "v"(x));
}

// (lags [j_begin, j_end) of the w there are; j_end < 0: all)
// (best / best_j, optional: the largest magnitude among this lane's lags and its lag -- rx4_body's QX speculates on it)
template <int SPS, int T0, int T1, int T2>
__device__ __forceinline__ void corr_fixed(const float2 *__restrict__ xs, int sps_rt, int w, int lane,
                                           const float2 *__restrict__ ctab, float2 cfl, float *__restrict__ corr,
                                           int j_begin = 0, int j_end = -1, uint32_t *best = nullptr, int *best_j = nullptr)
{
	constexpr int T[3] = {T0, T1, T2};
	constexpr int NT = T0 + T1 + T2;
	const int sps = SPS ? SPS : sps_rt;
	unsigned long long cf[NT];
	if (ctab) {
		const unsigned long long *__restrict__ ct = reinterpret_cast<const unsigned long long *>(ctab);
#pragma unroll
		for (int n = 0; n < NT; n++)
			cf[n] = ct[n];
	} else {
#pragma unroll
		for (int n = 0; n < NT; n++)
			cf[n] = (unsigned long long)__builtin_bit_cast(uint32_t, lane_val(cfl.x, n)) |
			        ((unsigned long long)__builtin_bit_cast(uint32_t, lane_val(cfl.y, n)) << 32);
	}
	const int j_stop = j_end < 0 ? w : j_end;
	for (int j = j_begin + lane; j < j_stop; j += 64) {
		float cj = 0.f;
		int wb = 0, base = 0;
#pragma unroll
		for (int ch = 0; ch < 3; ch++) {
			const float2 *xp = xs + wb + j;
			v2f acc = {0.f, 0.f};
#pragma unroll
			for (int n = 0; n < T[ch]; n++) {
				const float2 x = xp[n * sps];
				pk_cmac(acc, cf[base + n], (v2f){x.x, x.y});
			}
			base += T[ch];
			wb += T[ch] * sps + w - 1;
			cj += __builtin_amdgcn_sqrtf(fmaf(acc.x, acc.x, acc.y * acc.y));
		}
		corr[j] = cj;
		if (best) {
			// (a sum of square roots is not negative: its bits order like its value)
			const uint32_t cb = __builtin_bit_cast(uint32_t, cj);
			if (cb > *best) { *best = cb; *best_j = j; }
		}
	}
}

struct Lds4 {
	float2 *x;        // normalised sync-chunk windows of the burst being correlated [stage_samples]
	float *corr;      // 4 x cw correlation magnitudes
	float2 *coef;     // 32 rotated sync reference values
	int8_t *eb;       // 4 soft-bit rows
	uint32_t *bm;     // overlays x after pass 1
	uint64_t *surv;
	uint32_t *ubits;
};

// (ebrow, gen only: bytes between the four soft-bit rows -- rx4_body's EBROW)
__host__ __device__ inline size_t lds4_layout(int stage_samples, int cw, size_t *off, bool gen = false, bool ub_over = false,
                                              int ebrow = 432)
{
	// pass 1 keeps only the sync-chunk windows of the burst in LDS (everything else it needs is in
	// registers; pass 2 takes its kept samples out of the window registers -- the fused sps-4 kernel -- or re-reads them).
	// Decode-time data overlays all of it:
	//   [stage | corr 4 x cw | coef]   during pass 1 and the timing rows
	//   [bm | ubits | 4 soft-bit rows / window decisions]   from pass 2 on; during pass 2 itself the soft-bit
	//   table (2 KB, g_sb_lut) sits where the branch metrics will go
	// (the 13 x 64 halfwords of window decisions overlay the soft-bit rows, which are dead once the
	// branch metrics exist)
	// (ub_over, the batch kernel: the decoded words go where the branch metrics were -- all of them are consumed by the time
	// the survivor walk writes -- which brings the headline shape to 5 120 B, the LDS of eight waves per SIMD)
	const size_t dec_bytes = 4 * kSteps12 * 4 + (ub_over ? 0 : 4 * 8 * 4);
	const size_t stage_bytes = align16((size_t)stage_samples * 8);
	const size_t corr_bytes = align16((size_t)4 * cw * 4);
	off[0] = 0;
	off[1] = stage_bytes;
	off[2] = stage_bytes + corr_bytes;
	off[3] = dec_bytes;            // soft-bit rows
	if (gen) {
		// demodulation only (k_rx4g): no layer-1 data; the soft-bit rows overlay the pass-1 data, which is dead by then
		off[3] = 0;
		const size_t p1 = stage_bytes + corr_bytes + 18 * 8;
		const size_t p2 = 4 * (size_t)ebrow + kSbLutBytes;      // pass 2: soft-bit rows, then the soft-bit table
		// the small formats' pass 1 (one burst per row): 4 x 64 staged samples, 4 x <= 128 correlation values, 4 x 16 coefficients
		const size_t p3 = 4 * 64 * 8 + 4 * 128 * 4 + 4 * 16 * 8;
		const size_t m = p1 > p2 ? p1 : p2;
		return align16(m > p3 ? m : p3);
	}
	size_t total = stage_bytes + corr_bytes + 18 * 8;
	if (total < dec_bytes + 4 * 432)
		total = dec_bytes + 4 * 432;
	return align16(total);
}

// samples of the sync-chunk windows a burst type needs staged: sum over chunks of len*sps + w - 1
__host__ __device__ inline int stage_samples_of(const DevBurst &bt, int sps, int in_len)
{
	const int w = in_len - bt.len * sps + 1;
	int n = 0;
	for (int c = 0; c < bt.n_chunks[0]; c++)
		n += bt.sync[0][c].len * sps + w - 1;
	return n;
}

// 64-bit max within each 16-lane row
template <int X>
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long k)
{
	const uint32_t lo = row_xor<X>((uint32_t)k), hi = row_xor<X>((uint32_t)(k >> 32));
	const unsigned long long o = ((unsigned long long)hi << 32) | lo;
	return o > k ? o : k;
}

// The body works on bursts g0 .. n_end-1 (at most four) of `a` with one wavefront and its own LDS slice;
// k_rx4 is the batch kernel around it, k_rx_chain (below) the receive loop's feedback chain that calls it round after round.
// per-burst arrays of a launch: the batch kernel takes them from its arguments, the receive loop points them
// at the current round's log blocks (by value: they stay in scalar registers)
#ifdef GMR1_HIP_PROFILE
// cycle stamps of ONE burst of the receive loop (chain 0, the BCCH burst of round kStampRound): tools/loop_stamps.py
__device__ unsigned long long g_stamp[32];
__device__ int g_prof_flag;              // experiments of the profiling build (gmr1_hip_prof_flag)
__device__ int g_prof_miss;              // bursts of k_rx4's QX path whose speculated pick did not hold (gmr1_hip_prof_miss)
constexpr int kStampRound = 55;    // (late in a time slice: its first rounds share the CU with the previous slice's CCCH batch)
#define GMR1_STAMP(k)                                                        \
	do {                                                                    \
		if (LAT && io.stamp && lane == 0)                                   \
			io.stamp[k] = __builtin_readcyclecounter();                     \
	} while (0)
#else
#define GMR1_STAMP(k) do { } while (0)
#endif

struct RxIo {
#ifdef GMR1_HIP_PROFILE
	unsigned long long *stamp;
#endif
	const uint64_t *offset;
	const uint8_t *kind;
	const float *freq_shift;
	uint8_t *l2;
	int32_t *crc, *conv, *rv, *sync_id;
	float *toa, *freq_err, *energy;
	int8_t *ebits;
	float *ssyms;
};

// Burst-format parameters as the body reads them.  GEN (any one format per launch): the descriptor table.  The fused BCCH /
// DC6 path: the two formats' numbers themselves (nb.c:36-62, 94-120) -- one sync chunk of 11 / 7 symbols at 28, two of 3
// at 119 and 197, 234 pi/4-CQPSK symbols --, so that no phase waits for a table entry (a per-row descriptor lookup is a
// vector load from constant memory: three dependent ones sat in front of the sync-symbol terms); the host refuses to start
// the fused kernels unless the tables say exactly this (fused_formats_match, capi.cpp).
constexpr unsigned long long kFusedSymsBcch =      // [0 2 2 0 0 0 2 0 2 2 2 | 2 2 0 | 2 2 0], two bits each, first symbol lowest
	0ull | 2ull << 2 | 2ull << 4 | 0ull << 6 | 0ull << 8 | 0ull << 10 | 2ull << 12 | 0ull << 14 | 2ull << 16 | 2ull << 18 | 2ull << 20 |
	2ull << 22 | 2ull << 24 | 0ull << 26 | 2ull << 28 | 2ull << 30 | 0ull << 32;
constexpr unsigned long long kFusedSymsDc6 =       // [0 0 0 2 2 0 2 | 0 3 0 | 3 1 1]
	0ull | 0ull << 2 | 0ull << 4 | 2ull << 6 | 2ull << 8 | 0ull << 10 | 2ull << 12 |
	0ull << 14 | 3ull << 16 | 0ull << 18 | 3ull << 20 | 1ull << 22 | 1ull << 24;
template <bool GEN>
struct Fmt {
	static __device__ __forceinline__ int len(const DevBurst &b) { return GEN ? b.len : 234; }
	static __device__ __forceinline__ int nbits(const DevBurst &b) { return GEN ? b.nbits : 2; }
	static __device__ __forceinline__ float rotation(const DevBurst &b) { return GEN ? b.rotation : kPif / 4.0f; }
	static __device__ __forceinline__ int tl(const DevBurst &b, int kind) { return GEN ? b.sync_tl[0] : (kind ? 13 : 17); }
	static __device__ __forceinline__ int nch(const DevBurst &b) { return GEN ? b.n_chunks[0] : 3; }
	static __device__ __forceinline__ int clen(const DevBurst &b, int kind, int c) { return GEN ? b.sync[0][c].len : (c == 0 ? (kind ? 7 : 11) : 3); }
	static __device__ __forceinline__ int cpos(const DevBurst &b, int c) { return GEN ? b.sync[0][c].pos : (c == 0 ? 28 : (c == 1 ? 119 : 197)); }
	// training symbol n (counted through the chunks) of the first sequence; GEN: symbol nn of chunk ch of sequence sq
	static __device__ __forceinline__ int sym(const DevBurst &b, int kind, int sq, int ch, int nn, int n)
	{
		if constexpr (GEN)
			return b.sync[sq][ch].syms[nn];
		else
			return (int)(((kind ? kFusedSymsDc6 : kFusedSymsBcch) >> (2 * n)) & 3ull);
	}
};

// What the receive loop hands its burst body (LAT).
// (1) What depends only on WHERE the burst sits -- the window's mean and deviation, its energy, the normalised samples under
// the sync chunks -- prepared a round ahead by the chain's second wave (lat_prepare, on another SIMD) for the place the
// chain's schedule puts the next BCCH burst, eight frames on; a burst that turns out to sit elsewhere (the feedback moved
// the chain) is prepared by the body itself, as every burst of the batch kernels is.
// (2) What every burst reads from constant tables, kept where a wave of the loop gets at it in an LDS access instead of a
// trip to the L2 (each of these sat at the head of a phase of every burst of every round): the soft-bit table, the
// trellis-step descriptors and the cost words of the branch metrics (work-group copies), the decoder's per-lane
// constants and CRC syndrome words (registers).
struct LoopCo {                    // front wave <-> helper wave, within a tick (flags carry the tick's job id)
	int job;                       // F -> S, before the tick's barrier: id (> 0) of the job, 0: none
	float fsh;                     // the burst: frequency shift ...
	uint64_t off;                  // ... and first sample (S helps only if that is the window prepared)
	int p_id, p;                   // F -> S: the coarse peak (p < 0: never mind)
	int p_pred;                    // F -> S with the job: where the peak is expected (e_toa) -- S works that case out ahead
	int c_id;                      // S -> F: the correlation's tail is in place
	int s_id;                      // S -> F: ffe / psi of the three candidates are in place
	float ffe[3], psi[3];
	float tail[64];                // the correlation of the last partial round of lags (F copies it in: a front that was
	                               // squashed meanwhile never looks at it)
};
// (bounded: a wave that gives up does the work itself -- no hand-shake can hang the work-group)
__device__ __forceinline__ bool lds_wait_eq(const int *flag, int want)
{
	for (int i = 0; i < (1 << 16); i++) {
		if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == want)
			return true;
		__builtin_amdgcn_s_sleep(1);
	}
	return false;
}
__device__ __forceinline__ void lds_post(int *flag, int v)
{
	__hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NPL, int SPS>
struct LatPre {
	const uint64_t *h_off = nullptr;       // LDS: first sample of the window prepared (~0: none)
	const int *h_kind = nullptr;
	const float *h_stat = nullptr;         // LDS: mean re, mean im, 1 / deviation, burst energy
	const float2 *h_x = nullptr;           // LDS: the staged sync-chunk windows
	const unsigned char *lut = nullptr;
	const uint32_t *steps = nullptr;       // [2][kSteps12]
	const uint32_t *cost_a = nullptr, *cost_b = nullptr;   // [512] each
	uint32_t *vtab = nullptr;              // LDS: the decoder's per-step operand table (decode1_k5_12_lat), kLatTabBytes
	uint32_t dc = 0;
	uint4 sy0, sy1;
	// (3) The pipelined loop (k_rx_chain_pipe, PART): the burst is cut where its feedback exists -- after the sync-symbol
	// terms (pi4cxpsk.c:547-575: toa, freq_err) --, the FRONT on one wave, the rest (soft bits, layer 1) on others, rounds
	// later.  What crosses the cut: `cut`; the burst's raw window stays in LDS for the kept samples of pass 2.
	struct Cut { int found, d, sid; float toa, ffe, psi, avr, avi; };
	Cut *cut = nullptr;                    // LDS
	Cut out;                               // front: the same in (scalar) registers, for the walk
	float out_energy = 0.f;                // front: burst_energy() of the window
	// (4) The front's helper wave (k_rx_chain_pipe's S): for a burst whose window was prepared it correlates the last
	// partial round of lags while the front wave does the whole rounds, and, once the front has posted the coarse peak p,
	// forms the sync-symbol terms for the three values round(toa) can take (p - 1, p, p + 1: the bisection starts at p - 1
	// and moves less than one lag) while the front wave bisects.
	LoopCo *co = nullptr;                  // LDS; null: the front works alone
	int co_id = 0;
	// (5) PART 1 / 2: the one burst's operands in registers of the caller (a BCCH burst) -- not through the arrays of `io`,
	// whose every read would be an LDS round trip in front of the phase that needs it
	uint64_t b_off = 0;
	float b_fsh = 0.f;
	float2 *win_w = nullptr;               // LDS, front half: where the window goes if the front had to fetch it itself
	const float2 *win_r = nullptr;         // LDS, back half: the burst's raw window (lane l's samples l + 64 k as fetched)
};

// rotated reference of a fused format's training sequence under a caller-supplied frequency shift: value n in lane n
__device__ __forceinline__ float2 lat_coef(int kind, int lane, float fs, int sps)
{
	float2 cfl = make_float2(0.f, 0.f);
	if (lane < (kind ? 13 : 17)) {
		const int l0 = kind ? 7 : 11;
		const int nn = lane < l0 ? lane : (lane < l0 + 3 ? lane - l0 : lane - l0 - 3);
		float sn, cs;
		sincos_fast(fs * (float)(nn * sps), sn, cs);
		cfl = conj_ref_mul(2, (int)(((kind ? kFusedSymsDc6 : kFusedSymsBcch) >> (2 * lane)) & 3ull), make_float2(cs, sn));
	}
	return cfl;
}

// The sync-symbol terms of a fused-format burst whose sync-chunk windows are staged in LDS (the receive loop's bursts): fine
// frequency from the chunk sums, then the carrier phase (pi4cxpsk.c:381-433, 574-575) -- one burst, or one candidate
// timing d of a burst, per 16-lane row; lane col holds sync symbols n = col and n = col + 16.  The window scale 1 / sigma
// is irrelevant to every angle.
__device__ __forceinline__ void lat_sync_terms(const float2 *__restrict__ xst, int kind, int d, bool live, float fs, int sps, int w,
                                               int in_len, int col, float &ffe_o, float &psi_o)
{
	typedef Fmt<false> F;
	const DevBurst &bt = c_types[GMR1_HIP_BCCH];      // (Fmt<false> carries the numbers itself)
	const int tl = F::tl(bt, kind);
	float2 t0[2], xr[2];
	int chn[2], spos[2], idxv[2];
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int n = col + 16 * h;
		t0[h] = xr[h] = make_float2(0.f, 0.f);
		chn[h] = -1;
		spos[h] = idxv[h] = 0;
		if (n < tl && live) {
			int ch = 0, base = 0, cum = 0, wb = 0;
			for (int c = 0; c < 2; c++) {
				cum += F::clen(bt, kind, c);
				if (n >= cum) { base = cum; ch = c + 1; wb += F::clen(bt, kind, c) * sps + w - 1; }
			}
			const int nn = n - base;
			const int sp = F::cpos(bt, ch) + nn;
			const int idx = sp * sps + d;
			if (idx >= 0 && idx < in_len)
				xr[h] = xst[wb + nn * sps + d];
			chn[h] = ch;
			spos[h] = sp;
			idxv[h] = idx;
		}
	}
#pragma unroll
	for (int h = 0; h < 2; h++) {
		if (chn[h] >= 0) {
			float sn, cs;
			sincos_fast(fs * (float)idxv[h], sn, cs);
			const float2 x = cmul(xr[h], make_float2(cs, sn));
			t0[h] = conj_ref_mul(2, F::sym(bt, kind, 0, 0, 0, col + 16 * h), x);
		}
	}
	float ffe = 0.f;
	{
		float sumr[3], sumi[3];
#pragma unroll
		for (int c = 0; c < 3; c++) {
			const float pr = (chn[0] == c ? t0[0].x : 0.f) + (chn[1] == c ? t0[1].x : 0.f);
			const float pi = (chn[0] == c ? t0[0].y : 0.f) + (chn[1] == c ? t0[1].y : 0.f);
			sumr[c] = row_sum(pr);
			sumi[c] = row_sum(pi);
		}
		float f = 0.f;
#pragma unroll
		for (int i = 1; i < 3; i++) {
			const float ppos = (float)F::cpos(bt, i - 1) + (float)F::clen(bt, kind, i - 1) / 2.0f;
			const float cpos = (float)F::cpos(bt, i) + (float)F::clen(bt, kind, i) / 2.0f;
			const float re = sumr[i] * sumr[i - 1] - sumi[i] * (-sumi[i - 1]);
			const float im = sumr[i] * (-sumi[i - 1]) + sumi[i] * sumr[i - 1];
			f += atan2_fast(im, re) / (cpos - ppos);
		}
		ffe = f / (float)(3 - 1);
	}
	float tr = 0.f, ti = 0.f;
#pragma unroll
	for (int h = 0; h < 2; h++) {
		float2 tt = t0[h];
		if (ffe != 0.0f) {
			float sn, cs;
			sincos_fast(-ffe * (float)spos[h], sn, cs);
			tt = cmul(tt, make_float2(cs, sn));
		}
		tr += tt.x;
		ti += tt.y;
	}
	ffe_o = ffe;
	psi_o = atan2_fast(row_sum(ti), row_sum(tr));
}

// What the front's helper wave does for one BCCH burst of the loop whose window was prepared (LatPre (4)).
template <int SPS>
__device__ __forceinline__ void loop_front_helper_corr(const RxArgs &a, LoopCo *co, int id, const float2 *__restrict__ xst, int lane)
{
	const int sps = SPS ? SPS : a.sps;
	const int in_len = __builtin_amdgcn_readfirstlane(a.in_len[0]);
	const int w = in_len - 234 * sps + 1;
	const float fs = (co->fsh - kPif / 4.0f) / (float)sps;              // as rx4_body: (shift - rotation) / sps
	const float2 cfl = lat_coef(0, lane, fs, sps);
	const int j0 = ((w - 1) >> 6) << 6;                                 // the last, partial round of lags
	corr_fixed<SPS, 11, 3, 3>(xst, sps, w, lane, nullptr, cfl, co->tail - j0, j0, w);
	WSYNC();
	if (lane == 0)
		lds_post(&co->c_id, id);
}
// the sync-symbol terms of the three timing candidates p - 1, p, p + 1 -> co->ffe / psi
template <int SPS>
__device__ __forceinline__ void loop_front_helper_sync(const RxArgs &a, LoopCo *co, const float2 *__restrict__ xst, int lane, int p)
{
	const int sps = SPS ? SPS : a.sps;
	const int in_len = __builtin_amdgcn_readfirstlane(a.in_len[0]);
	const int w = in_len - 234 * sps + 1;
	const float fs = (co->fsh - kPif / 4.0f) / (float)sps;
	const int row = lane >> 4, col = lane & 15;
	float ffe, psi;
	lat_sync_terms(xst, 0, p - 1 + row, row < 3, fs, sps, w, in_len, col, ffe, psi);
	if (col == 0 && row < 3) {
		co->ffe[row] = ffe;
		co->psi[row] = psi;
	}
	WSYNC();
}

// The position-only part of a fused-format burst's pass 1 (rx4_body does the same, operation for operation): window,
// statistics, burst energy, normalised sync-chunk windows.
template <int NPL, int SPS>
__device__ __forceinline__ void lat_prepare(const RxArgs &a, uint64_t off, int kind, int lane, float2 *__restrict__ hx,
                                            float *__restrict__ hs, float2 *__restrict__ win = nullptr)
{
	const int sps = SPS ? SPS : a.sps;
	const int in_len = __builtin_amdgcn_readfirstlane(a.in_len[kind]);
	typedef Fmt<false> F;
	const DevBurst &bt = c_types[kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH];
	const int w = in_len - 234 * sps + 1;
	const float2 *__restrict__ in = a.iq + off;
	constexpr int NFULL = (SPS == 4 && NPL == 16) ? 15 : -1;
	constexpr int SIT = SPS == 4 ? 2 : 4;
	float2 wv[NPL];
	float2 sv[3][SIT];
	window_fetch<NPL, NFULL>(in, in_len, lane, wv);
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const int wl = F::clen(bt, kind, c) * sps + w - 1;
		const float2 *__restrict__ src = in + F::cpos(bt, c) * sps;
#pragma unroll
		for (int h = 0; h < SIT; h++) {
			const int sidx = lane + 64 * h;
			sv[c][h] = sidx < wl ? src[sidx] : make_float2(0.f, 0.f);
		}
	}
	if (win) {
		// the whole raw window, as fetched (lane l's samples l + 64 k): the back half of the burst takes its kept samples here
#pragma unroll
		for (int k = 0; k < NPL; k++)
			win[lane + 64 * k] = wv[k];
	}
	float avr, avi, inv;
	window_stats<NPL, NFULL>(wv, in_len, lane, avr, avi, inv);
	const float e = window_energy_regs<NPL>(wv, in_len, lane);
	int wb = 0;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const int wl = F::clen(bt, kind, c) * sps + w - 1;
#pragma unroll
		for (int h = 0; h < SIT; h++) {
			const int sidx = lane + 64 * h;
			if (sidx < wl) {
				const v2f nv = ((v2f){sv[c][h].x, sv[c][h].y} - (v2f){avr, avi}) * (v2f){inv, inv};
				hx[wb + sidx] = make_float2(nv.x, nv.y);
			}
		}
		wb += wl;
	}
	if (lane == 0) {
		hs[0] = avr;
		hs[1] = avi;
		hs[2] = inv;
		hs[3] = e;
	}
}

// LAT: the caller cares about the latency of ONE burst (the receive loop), not about throughput
// GEN: demodulation only, every burst of the one format a.fixed_type (one training sequence, QPSK, <= 3 sync chunks,
// <= 18 sync symbols, <= 256 symbols: NT3 speech, DC2, BCCH, DC6) -- the batch form of gmr1_pi4cxpsk_demod for large n
// FAC (with GEN, NPL = 8): a format with TWO training sequences of one chunk each at the same place, BPSK - NT3 FACCH.
// Both sequences are correlated over the one staged window; the second is ranked and timed on the SUM of both
// correlations, as the reference's uncleared accumulator has it (pi4cxpsk.c:207-237); the timing rows run once per
// sequence; `cw` holds both correlation arrays of a burst (first cw / 2 lags: sequence 0, then the sum).
// EN (fused batch kernel only): the caller wants burst_energy() of every window (RxArgs::energy; the receive loop's CCCH batch).
// A template parameter because keeping the window registers alive for it costs the headline instantiation its sixth wave
// (80 VGPRs with 2 spilled against 78 with none: 0.257 -> 0.265 ms per 100 k bursts, and the spill stores tripled the kernel's
// write traffic -- which is how it was found).
// PL (fused batch kernel at 4 samples per symbol only): the sample array is stored POLYPHASE-PLANAR -- sample s of the flat
// array sits at iq[(s & 3) * plane_stride + (s >> 2)] (include/gmr1_hip.h, gmr1_hip_rx_bcch_ccch_batch_planar_dev).  Only
// addresses change: the window sits in registers in the same QUAD layout as the interleaved call's (lane l holds samples
// 256 b + 4 l + c: place l + 64 b of plane c, a coalesced 512-byte load), so every sum is formed over the same samples in the
// same order and every result is bit-identical; pass 2's 234 samples at stride 4 from sample d (pi4cxpsk.c:292-295) are 234
// CONSECUTIVE samples of plane (offset + d) & 3 -- 15 lines of 128 bytes instead of every line of the window.
// EBROW (GEN only): bytes between the four bursts' soft-bit rows in LDS.  432 holds any format; the kernel that decodes NT3
// speech bursts right behind the demodulator (k_rx4g_tch3) packs its 212-byte rows at 216 to leave the decoder its tables.
// PART (LAT only): 0 the whole burst; 1 its front -- pass 1, timing, sync-symbol terms; rv / toa / freq_err / energy to
// `io`, the rest of what the middle needs to pre->cut --; 2 its middle -- pass 2 out of pre->win_r, then the decoder's
// operand table into pre->vtab; the decoder itself is the caller's third stage (see LatPre (3))
template <int NPL, int SPS, bool LAT = false, bool GEN = false, bool FAC = false, bool ACC = false, bool EN = true, bool PL = false,
          int EBROW = 432, int PART = 0>
__device__ __forceinline__ void rx4_body(const RxArgs &a, const RxIo io, int stage_samples, int cw, int g0, int n_end,
                                         unsigned char *__restrict__ lds_raw, int lane, LatPre<NPL, SPS> *pre = nullptr)
{
	const int row = lane >> 4, col = lane & 15;
	const int sps = SPS ? SPS : a.sps;
	// the small generic variant (NPL = 8; the host launches it for formats of <= 128 symbols with one sync chunk of
	// <= 16 symbols whose window is <= 64 samples: NT3 speech, DC2) drops the unrolled work the long bursts need
	constexpr bool SMALL = GEN && NPL == 8;
	static_assert(!FAC || SMALL, "the two-sequence variant builds on the small generic one");
	static_assert(!PL || (SPS == 4 && !GEN && !LAT && !EN), "the planar layout exists for the fused batch kernel at sps 4");
	static_assert(PART == 0 || LAT, "only the receive loop's burst is cut in two");
	const int cwh = FAC ? cw / 2 : cw;                // lags per correlation array
	// The next burst's window in flight during this burst's correlation costs 32 registers at the body's peak.  The fused
	// kernel does without: 78 instead of 87 VGPRs is the step from five to six waves per SIMD, and the sixth wave hides more
	// latency than the prefetch did (0.288 -> 0.274 ms per 100 k bursts).  The other instantiations keep it.
	constexpr bool PREFETCH_NEXT = GEN || LAT;
	constexpr int NSYM = SMALL ? 2 : 4;               // 64-symbol pieces of a burst
	constexpr int NCHK = SMALL ? 1 : 3;               // sync chunks
	constexpr int NSH = SMALL ? 1 : 2;                // 16-symbol pieces of the sync sequence
	size_t off[4];
	constexpr bool UB_OVER = !LAT && !GEN;
	lds4_layout(stage_samples, cw, off, GEN, UB_OVER, EBROW);
	Lds4 L;
	L.x = reinterpret_cast<float2 *>(lds_raw + off[0]);
	L.corr = reinterpret_cast<float *>(lds_raw + off[1]);
	L.coef = reinterpret_cast<float2 *>(lds_raw + off[2]);
	L.eb = reinterpret_cast<int8_t *>(lds_raw + off[3]);
	L.bm = reinterpret_cast<uint32_t *>(lds_raw + off[0]);
	L.surv = reinterpret_cast<uint64_t *>(lds_raw + off[3]);
	L.ubits = reinterpret_cast<uint32_t *>(lds_raw + off[0] + (UB_OVER ? 0 : 4 * kSteps12 * 4));

	const int g_row = g0 + row;                       // this row's burst
	const bool row_live = g_row < n_end;

	float avr_r = 0.f, avi_r = 0.f;                    // window mean of this row's burst
	const float2 *xst_lat = L.x;                       // LAT: where the one burst's sync-chunk windows are staged

	// QL (the fused batch kernel at 4 samples per symbol, both sample layouts): the window sits in registers in the QUAD layout
	// (window_fetch_q).  QX (interleaved samples): pass 2's kept samples never make a second trip to memory.  Right behind a
	// burst's correlation its pick d = round(toa) is SPECULATED as the lag of the largest correlation magnitude (the timing
	// rows below start their early / late walk on the strongest lag p of the best three-lag window and end within a lag of it,
	// pi4cxpsk.c:240, so round(toa) is p - 1, p or p + 1, and p is the largest magnitude itself whenever the peak is clean);
	// the 234 kept samples d + 4 i are then one sub-slot of every lane of the window registers, a lane rotation away from
	// pass 2's assignment, and what pass 2 takes from a kept sample -- its phase -- is formed on the spot: four registers a
	// burst instead of a window.  The timing rows then compute the pick as ever; a burst whose pick is NOT the speculated one
	// (a few per cent: noise decides where toa lies half-way between two samples) takes the old route -- kept samples and sync
	// symbols re-read from memory, the same operations on the same numbers -- so no output depends on the speculation.
	constexpr bool QL = !GEN && !LAT && !EN && SPS == 4 && NPL == 16;
	constexpr bool QX = QL && !PL;
	static_assert(!PL || QL, "the planar layout exists for the fused batch kernel");
	float th_q[QX ? 4 : 1][4];                         // QX: phase (turns) of kept sample lane + 64 r of burst q
	uint32_t zm_q = 0xff000000u;                       // bit 4 q + r: that sample counts as 0 + 0j (outside the window, or zero);
	                                                   // bits 24-31 (row q's lanes): burst q's speculated pick (0 ... 80; 255: none)
	float2 xs_q[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};   // row q: normalised samples of sync symbols col, col + 16 at the speculated pick
	float inv_r = 1.0f;                                // row q: 1 / sigma of burst q's window
	// a burst's operands: the arrays of `io`, or (the loop's pipeline stages) the caller's registers (LatPre (5))
	auto op_kind = [&](int g) -> int {
		if constexpr (LAT && PART != 0) return 0; else return io.kind[g] ? 1 : 0;
	};
	auto op_fsh = [&](int g) -> float {
		if constexpr (LAT && PART != 0) return pre->b_fsh; else return io.freq_shift ? io.freq_shift[g] : 0.0f;
	};
	auto op_off = [&](int g) -> uint64_t {
		if constexpr (LAT && PART != 0) return pre->b_off; else return io.offset[g];
	};
	float2 wv_own[NPL];
	bool co_on = false;                                // (LAT, PART 1) the helper wave shares this burst's front (LatPre (4))
	int co_p = 0;
	if constexpr (PART == 2)
		GMR1_STAMP(14);
	else
		GMR1_STAMP(0);
	// =========================== pass 1: correlation magnitudes ===========================
	if constexpr (PART != 2) {
	if constexpr (SMALL) {
		// The short formats (<= 512 samples, one sync chunk of <= 16 symbols, <= 64 lags) take pass 1 with ONE BURST PER ROW
		// as well: lane `col` of a row reads samples col, col + 16, ... of the row's burst (16 lanes x 8 B = one 128-byte
		// line per row and load), so window sums, the division, the rotated reference, the staged sync window and the
		// correlation are formed once per wave for its four bursts instead of once per burst with the other rows' lanes
		// idle or duplicating.
		const DevBurst &bt = c_types[a.fixed_type];
		const int in_len = __builtin_amdgcn_readfirstlane(a.in_len[0]);
		const int w = in_len - bt.len * sps + 1;
		const int tl = bt.sync_tl[0];
		const int len0 = bt.sync[0][0].len;
		const int wl = len0 * sps + w - 1;                 // samples under the sync chunk for every lag (<= 64)
		const int gq = row_live ? g_row : g0;              // a dead row shadows the wave's first burst
		const float2 *__restrict__ in = a.iq + io.offset[gq];
		const float fsh = io.freq_shift ? io.freq_shift[gq] : 0.0f;
		const float fs = (fsh - bt.rotation) / (float)sps;
		float2 *xs = reinterpret_cast<float2 *>(lds_raw) + row * 64;
		L.corr = reinterpret_cast<float *>(lds_raw + 2048);
		float2 *coef = reinterpret_cast<float2 *>(lds_raw + 4096) + row * 16;
		// what needs memory is asked for first: the samples under the sync chunk ...
		float2 sv[4];
		{
			const float2 *__restrict__ src = in + bt.sync[0][0].pos * sps;
#pragma unroll
			for (int h = 0; h < 4; h++) {
				const int sidx = col + 16 * h;
				sv[h] = sidx < wl ? src[sidx] : make_float2(0.f, 0.f);
			}
		}
		// ... and the whole window, ONE sweep: sum x and sum |x|^2 (a second sweep for the variance about the mean would
		// be a third trip through the memory system for a kernel whose two -- this one and pass 2's -- already load it
		// fully; sigma only sets a scale nothing downstream depends on)
		const int nit = in_len >> 4, rem = in_len & 15;
		const bool want_en = io.energy != nullptr;
		v2f s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll 8
		for (int t = 0; t < nit; t++) {
			const float2 x = in[col + 16 * t];
			s2 += (v2f){x.x, x.y};
			q2 = __builtin_elementwise_fma((v2f){x.x, x.y}, (v2f){x.x, x.y}, q2);
		}
		if (col < rem) {
			const float2 x = in[col + 16 * nit];
			s2 += (v2f){x.x, x.y};
			q2 = __builtin_elementwise_fma((v2f){x.x, x.y}, (v2f){x.x, x.y}, q2);
		}
		float en = 0.f;
		if (want_en) {
			// burst_energy() (gmr1_rx.c:172-182): the inner 30 / 32 of the raw window, served by the caches
			const int bd = in_len >> 5;
			for (int idx = col; idx < in_len; idx += 16)
				if (idx >= bd && idx < in_len - bd) {
					const float2 x = in[idx];
					en = fmaf(x.x, x.x, fmaf(x.y, x.y, en));
				}
		}
		// rotated reference of the training sequence(s), one value per lane of the row, while the samples travel
		if constexpr (FAC) {
			if (col < 2 * tl)
				coef[col] = sync_coef_seq1(bt, col >= tl ? 1 : 0, col >= tl ? col - tl : col, sps, fs);
		} else {
			float2 cfl = make_float2(0.f, 0.f);
			if (io.freq_shift == nullptr)
				cfl = g_coef0[sps][a.fixed_type][col];
			else if (col < tl)
				cfl = sync_coef0(bt, col, sps, fs);
			coef[col] = cfl;
		}
		const float inv_n = __builtin_amdgcn_rcpf((float)in_len);
		const float avr = row_sum(s2.x) / (float)in_len, avi = row_sum(s2.y) / (float)in_len;     // true division, see load_normalise
		const v2f av = {avr, avi};
		// sum |x - m|^2 = sum |x|^2 - n |m|^2 (never below zero)
		const float var = fmaxf(fmaf(-(float)in_len, fmaf(avr, avr, avi * avi), row_sum(q2.x + q2.y)), 0.0f) * inv_n;
		float stddev = __builtin_amdgcn_sqrtf(var);
		if (stddev == 0.0f)
			stddev = 1.0f;
		const float inv = __builtin_amdgcn_rcpf(stddev);
		avr_r = avr;
		avi_r = avi;
		if (want_en) {
			const float e = row_sum(en) / (float)in_len;
			if (col == 0 && row_live)
				io.energy[g_row] = e;
		}
#pragma unroll
		for (int h = 0; h < 4; h++) {
			const int sidx = col + 16 * h;
			if (sidx < wl) {
				const v2f nv = ((v2f){sv[h].x, sv[h].y} - av) * (v2f){inv, inv};
				xs[sidx] = make_float2(nv.x, nv.y);
			}
		}
		WSYNC();
		float *corr = L.corr + row * cw;
		for (int j = col; j < w; j += 16) {
			const float2 *xp = xs + j;
			v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
			for (int n = 0; n < len0; n++) {
				const float2 x = xp[n * sps];
				const float2 c0 = coef[n];
				acc0 = __builtin_elementwise_fma((v2f){-c0.y, c0.y}, (v2f){x.y, x.x}, acc0);
				acc0 = __builtin_elementwise_fma((v2f){c0.x, c0.x}, (v2f){x.x, x.y}, acc0);
				if constexpr (FAC) {
					const float2 c1 = coef[tl + n];
					acc1 = __builtin_elementwise_fma((v2f){-c1.y, c1.y}, (v2f){x.y, x.x}, acc1);
					acc1 = __builtin_elementwise_fma((v2f){c1.x, c1.x}, (v2f){x.x, x.y}, acc1);
				}
			}
			float cj = 0.f;
			cj += sqrtf(fmaf(acc0.x, acc0.x, acc0.y * acc0.y));
			corr[j] = cj;                                   // FAC: sequence 0
			if constexpr (FAC) {
				cj += sqrtf(fmaf(acc1.x, acc1.x, acc1.y * acc1.y));
				corr[cwh + j] = cj;                         // what sequence 1 is ranked and timed on
			}
		}
	} else {
	float2 (&wv)[NPL] = wv_own;
	const int lane_outer = lane;
	// QX: where, in the staged sync-chunk windows of a BCCH / of a DC6 burst, the sample under sync symbol `col` lies for pick 0
	// (chunk c's window starts at staged sample wb_c; symbol nn of it, at pick d, is staged sample wb_c + 4 nn + d); -1: none
	int xsp_bc = 0xffff;                               // (the two places in one register, a byte each; 255: none)
	if constexpr (QX) {
		auto place = [&](int kind) -> int {
			const int w = a.in_len[kind] - 234 * 4 + 1;
			const int c0 = kind ? 7 : 11;
			const int n = col;
			const int ch = n < c0 ? 0 : (n < c0 + 3 ? 1 : 2);
			const int nn = n - (ch == 0 ? 0 : (ch == 1 ? c0 : c0 + 3));
			const int wb = ch == 0 ? 0 : (ch == 1 ? c0 * 4 + w - 1 : c0 * 4 + w - 1 + 3 * 4 + w - 1);
			return n < c0 + 6 ? wb + nn * 4 : -1;
		};
		xsp_bc = (place(0) & 0xff) | ((place(1) & 0xff) << 8);      // (places are below 240)
	}
	// (QX: unrolled, the bursts' phase registers th_q[q] are picked at compile time and come to life one burst at a time)
#pragma clang loop unroll_count(QX ? 4 : 1)
	for (int q = 0; q < 4; q++) {
		const int g = g0 + q;
		if (g >= n_end)
			break;
		// (QX: the lane number is made opaque once per iteration, so that nothing derived from it -- addresses, staging
		// predicates, tap places -- is hoisted out of the loop to sit in registers beside the bursts' kept phases)
		int lane_l = lane_outer;
		if constexpr (QX)
			asm volatile("" : "+v"(lane_l));
		const int lane = lane_l;
		const int row = lane >> 4, col = lane & 15;
		const int kind = GEN ? 0 : __builtin_amdgcn_readfirstlane(op_kind(g));
		const int type = GEN ? a.fixed_type : (kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH);
		const int in_len = __builtin_amdgcn_readfirstlane(a.in_len[kind]);
		const DevBurst &bt = c_types[type];
		typedef Fmt<GEN> F;
		const int w = in_len - F::len(bt) * sps + 1;
		const float fsh = op_fsh(g);
		const float fs = (fsh - F::rotation(bt)) / (float)sps;

		const float2 *__restrict__ in = a.iq + op_off(g);
		const int sl = lane;
		const int tl = F::tl(bt, kind);
		const int nch = F::nch(bt);
		constexpr int NFULL = (!GEN && SPS == 4 && NPL == 16) ? 15 : -1;
		// everything that needs memory is asked for first -- the whole window (statistics) and, again,
		// the ~300 samples under the sync chunks (they go to LDS; the second request hits the lines the
		// first one is fetching) -- and what needs no data (the rotated reference) is computed while it
		// travels
		constexpr int SIT = SMALL ? 1 : (SPS == 4 ? 2 : 4);   // 64-sample pieces per chunk window
		float2 sv_own[NCHK][SIT];
		float2 (&sv)[NCHK][SIT] = sv_own;
		auto fetch_window = [&](const float2 *__restrict__ from) {
			window_fetch<NPL, NFULL>(from, in_len, lane, wv);
#pragma unroll
			for (int c = 0; c < NCHK; c++) {
				const int wl = c < nch ? F::clen(bt, kind, c) * sps + w - 1 : 0;
				const float2 *__restrict__ src = from + (c < nch ? F::cpos(bt, c) * sps : 0);
#pragma unroll
				for (int h = 0; h < SIT; h++) {
					const int sidx = lane + 64 * h;
					sv[c][h] = sidx < wl ? src[sidx] : make_float2(0.f, 0.f);
				}
			}
		};
		// LAT: prepared a round ago by the chain's helper wave, if the burst sits where it was expected (LatPre)
		bool prepared = false;
		const float2 *__restrict__ xst = L.x;
		if constexpr (LAT) {
			prepared = *pre->h_off == op_off(g) && *pre->h_kind == kind;
			if (prepared)
				xst = xst_lat = pre->h_x;
			if constexpr (PART == 1)
				co_on = prepared && pre->co != nullptr && kind == 0;
		}
		if (LAT) {
			if (!prepared)
				fetch_window(in);
		} else {
			if constexpr (PL) {
				window_fetch_q_planar(a.iq, a.plane_stride, io.offset[g], in_len, lane, wv);
			} else {
			if constexpr (QL)
				window_fetch_q(in, in_len, lane, wv);
			else if (q == 0 || !PREFETCH_NEXT)
				window_fetch<NPL, NFULL>(in, in_len, lane, wv);
#pragma unroll
			for (int c = 0; c < NCHK; c++) {
				const int wl = c < nch ? F::clen(bt, kind, c) * sps + w - 1 : 0;
				const float2 *__restrict__ src = in + (c < nch ? F::cpos(bt, c) * sps : 0);
#pragma unroll
				for (int h = 0; h < SIT; h++) {
					const int sidx = lane + 64 * h;
					sv[c][h] = sidx < wl ? src[sidx] : make_float2(0.f, 0.f);
				}
			}
			}
		}
		// rotated reference of the (single) sync sequence: without a caller-supplied frequency shift it only depends on
		// the burst format and sps -- a table built once with this same arithmetic (g_coef0)
		// (lane n keeps value n: tl <= 32 for every format this body is launched for)
		float2 cfl = make_float2(0.f, 0.f);
		// (!GEN: the table row itself, read by scalar loads in corr_fixed -- a wave-uniform pointer, so set outside the
		// lane-dependent branches below)
		const float2 *__restrict__ ctab = (!GEN && io.freq_shift == nullptr) ? c_coef0[sps][type] : nullptr;
		if (io.freq_shift == nullptr) {
			if (GEN && lane < 32)
				cfl = g_coef0[sps][type][lane];
		} else if (lane < tl) {
			if constexpr (GEN) {
				cfl = sync_coef0(bt, lane, sps, fs);
			} else {
				// the same with the fused formats' numbers (Fmt<false>): no table walk
				cfl = lat_coef(kind, lane, fs, sps);
			}
		}
		WSYNC();
		if constexpr (FAC) {
			if (lane < 2 * tl)
				L.coef[lane] = sync_coef_seq1(bt, lane >= tl ? 1 : 0, lane >= tl ? lane - tl : lane, sps, fs);
		} else if (GEN && lane < tl)
			L.coef[lane] = cfl;
		float avr, avi, inv;
		if (prepared) {
			avr = pre->h_stat[0];
			avi = pre->h_stat[1];
			inv = pre->h_stat[2];
			if constexpr (PART == 1)
				pre->out_energy = pre->h_stat[3];
			if (io.energy && lane == 0)
				io.energy[g] = pre->h_stat[3];
		} else {
		if constexpr (QL)
			window_stats_q(wv, in_len, lane, avr, avi, inv);
		else
			window_stats<NPL, NFULL>(wv, in_len, sl, avr, avi, inv, -1, 1);
		if constexpr (LAT && PART == 1) {
			if (pre->win_w) {
#pragma unroll
				for (int k = 0; k < NPL; k++)
					pre->win_w[lane + 64 * k] = wv[k];
			}
		}
		if constexpr (LAT && PART == 1) {
			const float e = window_energy_regs<NPL>(wv, in_len, lane);
			pre->out_energy = e;
			if (io.energy && lane == 0)
				io.energy[g] = e;
		} else if ((LAT || !PREFETCH_NEXT) && EN && io.energy) {
			// burst_energy() while the window is still in registers
			const float e = window_energy_regs<NPL>(wv, in_len, lane);
			if (lane == 0)
				io.energy[g] = e;
		}
		}
		GMR1_STAMP(1);
		if (row == q) { avr_r = avr; avi_r = avi; inv_r = inv; }
		if (PREFETCH_NEXT && !LAT && q + 1 < 4 && g + 1 < n_end) {
			// the next burst's window travels during this burst's correlation
			const int kind1 = GEN ? 0 : __builtin_amdgcn_readfirstlane(io.kind[g + 1] ? 1 : 0);
			window_fetch<NPL, NFULL>(a.iq + io.offset[g + 1], __builtin_amdgcn_readfirstlane(a.in_len[kind1]), lane, wv);
		}
		if (!LAT && PREFETCH_NEXT && EN && io.energy) {
			// second read (L2): the registers already hold the next burst's window
			const float e = window_energy<NPL>(in, in_len, lane);
			if (lane == 0)
				io.energy[g] = e;
		}
		// stage the sync-chunk windows, normalised: window c = samples [pos_c sps, pos_c sps + len_c sps + w - 1)
		if constexpr (PL) {
			// The quad layout stages out of the window registers (the planar call; the interleaved one asks for the samples under
			// the sync chunks a second time, as the other instantiations do -- staging out of registers that have to last through
			// the correlation for the kept samples costs it more registers than it has): a lane's four consecutive samples 256 b + 4 l .. + 3 lie wholly
			// inside a chunk window or wholly outside it (the windows start on a symbol, 4 | pos sps, and are len sps + w - 1
			// samples long with w - 1 = 80 or 40), so a lane whose quad is inside writes it with two 16-byte stores.  Chunk 0
			// ([112, 236) at most) lies in quarter 0, chunk 1 ([476, 568)) in quarters 1 and 2, chunk 2 ([788, 880)) in quarter 3.
			const int wl0 = F::clen(bt, kind, 0) * 4 + w - 1, wl1 = 3 * 4 + w - 1;
			const v2f av = {avr, avi}, iv = {inv, inv};
			auto stage_quad = [&](int b, int start, int wl, int wb) {
				const int rel = 256 * b + 4 * lane - start;
				if (rel >= 0 && rel < wl) {
					v2f n0 = ((v2f){wv[4 * b].x, wv[4 * b].y} - av) * iv, n1 = ((v2f){wv[4 * b + 1].x, wv[4 * b + 1].y} - av) * iv;
					v2f n2 = ((v2f){wv[4 * b + 2].x, wv[4 * b + 2].y} - av) * iv, n3 = ((v2f){wv[4 * b + 3].x, wv[4 * b + 3].y} - av) * iv;
					float4 *dst = reinterpret_cast<float4 *>(L.x + wb + rel);
					dst[0] = make_float4(n0.x, n0.y, n1.x, n1.y);
					dst[1] = make_float4(n2.x, n2.y, n3.x, n3.y);
				}
			};
			stage_quad(0, 28 * 4, wl0, 0);
			stage_quad(1, 119 * 4, wl1, wl0);
			stage_quad(2, 119 * 4, wl1, wl0);
			stage_quad(3, 197 * 4, wl1, wl0 + wl1);
		} else
		if (!prepared) {
			int wb = 0;
#pragma unroll
			for (int c = 0; c < NCHK; c++) {
				const int wl = c < nch ? F::clen(bt, kind, c) * sps + w - 1 : 0;
#pragma unroll
				for (int h = 0; h < SIT; h++) {
					const int sidx = sl + 64 * h;
					if (sidx < wl) {
						const v2f nv = ((v2f){sv[c][h].x, sv[c][h].y} - (v2f){avr, avi}) * (v2f){inv, inv};
						L.x[wb + sidx] = make_float2(nv.x, nv.y);
					}
				}
				wb += wl;
			}
		}
		WSYNC();
		float *corr = L.corr + q * cw;
		if constexpr (!GEN) {
			// BCCH / DC6: static tap structure (the host refuses to start this kernel if the tables say otherwise)
			if (LAT && PART == 1 && co_on) {
				// the whole rounds of lags here, the last partial one on the helper wave
				const int j0 = ((w - 1) >> 6) << 6;
				corr_fixed<SPS, 11, 3, 3>(xst, sps, w, lane, ctab, cfl, corr, 0, j0);
				if (lds_wait_eq(&pre->co->c_id, pre->co_id)) {
					if (lane < w - j0)
						corr[j0 + lane] = pre->co->tail[lane];
				} else
					corr_fixed<SPS, 11, 3, 3>(xst, sps, w, lane, ctab, cfl, corr, j0, w);
			} else if constexpr (QX) {
				uint32_t cbest = 0;
				int cbj = 0;
				if (kind == 0)
					corr_fixed<SPS, 11, 3, 3>(xst, sps, w, lane, ctab, cfl, corr, 0, -1, &cbest, &cbj);
				else
					corr_fixed<SPS, 7, 3, 3>(xst, sps, w, lane, ctab, cfl, corr, 0, -1, &cbest, &cbj);
				// the lag of the largest magnitude (any lane that holds it): the speculated pick
				uint32_t mx = cbest, o;
				o = dpp<0xB1>(mx); mx = o > mx ? o : mx;
				o = dpp<0x4E>(mx); mx = o > mx ? o : mx;
				o = dpp<0x141>(mx); mx = o > mx ? o : mx;
				o = dpp<0x140>(mx); mx = o > mx ? o : mx;
				const uint32_t m01 = max((uint32_t)__builtin_amdgcn_readlane((int)mx, 0), (uint32_t)__builtin_amdgcn_readlane((int)mx, 16));
				const uint32_t m23 = max((uint32_t)__builtin_amdgcn_readlane((int)mx, 32), (uint32_t)__builtin_amdgcn_readlane((int)mx, 48));
				const uint32_t M = max(m01, m23);
				const unsigned long long holders = __ballot(cbest == M);
				const int d = __builtin_amdgcn_readlane(cbj, holders ? __builtin_ctzll(holders) : 0);     // 0 <= d < w <= 81
				// row q keeps what the rows below need of this burst at that pick: the normalised samples under its sync symbols
				// (still staged) ...
				if (row == q) {
					zm_q = (zm_q & 0x00ffffffu) | ((uint32_t)d << 24);
					const int pl = (xsp_bc >> (kind ? 8 : 0)) & 0xff;
					if (pl != 0xff)
						xs_q[0] = xst[pl + d];
					// (the one symbol beyond the sixteenth: BCCH's n = 16, symbol 2 of the third chunk, on the row's first lane)
					if (kind == 0 && col == 0)
						xs_q[1] = xst[11 * 4 + w - 1 + 3 * 4 + w - 1 + 2 * 4 + d];
				}
				// ... and every lane the phases of its four kept samples i = lane + 64 r: sample 4 i + d is sub-slot d & 3 of lane
				// (i + (d >> 2)) & 63 in quarter (i + (d >> 2)) >> 6 of the window registers
				{
					const int srcl = lane + (d >> 2);
					const int addr = (srcl & 63) << 2;
					const bool up = srcl >= 64;
					float2 rot[4];
					auto pull = [&](const float2 &v) {
						return make_float2(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v.x))),
						                   __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v.y))));
					};
					switch (d & 3) {                                  // (wave-uniform)
					case 0: rot[0] = pull(wv[0]); rot[1] = pull(wv[4]); rot[2] = pull(wv[8]); rot[3] = pull(wv[12]); break;
					case 1: rot[0] = pull(wv[1]); rot[1] = pull(wv[5]); rot[2] = pull(wv[9]); rot[3] = pull(wv[13]); break;
					case 2: rot[0] = pull(wv[2]); rot[1] = pull(wv[6]); rot[2] = pull(wv[10]); rot[3] = pull(wv[14]); break;
					default: rot[0] = pull(wv[3]); rot[1] = pull(wv[7]); rot[2] = pull(wv[11]); rot[3] = pull(wv[15]); break;
					}
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int i = lane + 64 * r;
						float2 x = up ? (r < 3 ? rot[r < 3 ? r + 1 : 3] : make_float2(0.f, 0.f)) : rot[r];
						// (kept samples 0 .. 191 lie inside any window this kernel is started for: 4 * 191 + 80 < 960 <= in_len)
						const bool ok = r < 3 || (i < 234 && 4 * i + d < in_len);
						// (as pass 2 does it: the mean is subtracted whether or not the lane has a sample; the zero test knows)
						x.x -= avr;
						x.y -= avi;
						th_q[q][r] = atan2_turns(x.y, x.x);
						if (!ok || (x.x == 0.0f && x.y == 0.0f))
							zm_q |= 1u << (4 * q + r);
					}
				}
			} else if (kind == 0)
				corr_fixed<SPS, 11, 3, 3>(xst, sps, w, lane, ctab, cfl, corr);
			else
				corr_fixed<SPS, 7, 3, 3>(xst, sps, w, lane, ctab, cfl, corr);
		} else if constexpr (FAC) {
			const int len = bt.sync[0][0].len;
			for (int j = lane; j < w; j += 64) {
				const float2 *xp = L.x + j;
				v2f acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
				for (int n = 0; n < len; n++) {
					const float2 x = xp[n * sps];
					const float2 c0 = L.coef[n], c1 = L.coef[tl + n];
					acc0 = __builtin_elementwise_fma((v2f){-c0.y, c0.y}, (v2f){x.y, x.x}, acc0);
					acc0 = __builtin_elementwise_fma((v2f){c0.x, c0.x}, (v2f){x.x, x.y}, acc0);
					acc1 = __builtin_elementwise_fma((v2f){-c1.y, c1.y}, (v2f){x.y, x.x}, acc1);
					acc1 = __builtin_elementwise_fma((v2f){c1.x, c1.x}, (v2f){x.x, x.y}, acc1);
				}
				float cj = 0.f;
				cj += sqrtf(fmaf(acc0.x, acc0.x, acc0.y * acc0.y));
				corr[j] = cj;                               // sequence 0
				cj += sqrtf(fmaf(acc1.x, acc1.x, acc1.y * acc1.y));
				corr[cwh + j] = cj;                         // what sequence 1 is ranked and timed on
			}
		} else
		for (int j = lane; j < w; j += 64) {
			float cj = 0.f;
			int base = 0, wb = 0;
			for (int ch = 0; ch < nch; ch++) {
				const int len = F::clen(bt, kind, ch);
				const float2 *xp = L.x + wb + j;          // staged window of this chunk
				const float2 *cp = L.coef + base;
				// (ar, ai) += c x as two packed FMAs: (-c.im x.im, c.im x.re) first, then c.re (x.re, x.im) -- the order the
				// scalar chains ar = fma(c.re, x.re, fma(-c.im, x.im, ar)), ai = fma(c.re, x.im, fma(c.im, x.re, ai)) had
				v2f acc = {0.f, 0.f};
				for (int n = 0; n < len; n++) {
					const float2 x = xp[n * sps];
					const float2 cf = cp[n];
					acc = __builtin_elementwise_fma((v2f){-cf.y, cf.y}, (v2f){x.y, x.x}, acc);
					acc = __builtin_elementwise_fma((v2f){cf.x, cf.x}, (v2f){x.x, x.y}, acc);
				}
				const float ar = acc.x, ai = acc.y;
				base += len;
				wb += len * sps + w - 1;
				cj += sqrtf(fmaf(ar, ar, ai * ai));
			}
			corr[j] = cj;
		}
	}
	}
	WSYNC();
	GMR1_STAMP(2);
	if (a.dbg_stop == 2) return;
	}

	// per-row (lane-resident) burst parameters: looked up here, not before pass 1, which has no register to spare for them
	const int kind_r = (!GEN && row_live) ? op_kind(g_row) : 0;
	const int type_r = GEN ? a.fixed_type : (kind_r ? GMR1_HIP_DC6 : GMR1_HIP_BCCH);
	typedef Fmt<GEN> F;
	const int in_len_r = kind_r ? a.in_len[1] : a.in_len[0];
	const float fsh_r = row_live ? op_fsh(g_row) : 0.0f;
	const DevBurst &bt_r = c_types[type_r];
	const float fs_r = (fsh_r - F::rotation(bt_r)) / (float)sps;     // pi4cxpsk.c:539
	const int w_r = in_len_r - F::len(bt_r) * sps + 1;
	const float2 *__restrict__ in_r = a.iq + (row_live ? op_off(g_row) : 0);

	// =========================== rows: peak + early/late timing ===========================
	// osmo_cxvec_peak_energy_find(corr, 3, PEAK_EARLY_LATE, &peak), pi4cxpsk.c:240
	// LAT: the one burst is row 0's; the other rows work on ITS correlation (speculative bisection below)
	const float *cr = L.corr + (LAT ? 0 : row) * cw;          // FAC: moved to the second array for the second sequence
	const int w_p = LAT ? __builtin_amdgcn_readlane(w_r, 0) : w_r;
	const int tl_p = LAT ? F::tl(c_types[__builtin_amdgcn_readlane(type_r, 0)], __builtin_amdgcn_readlane(kind_r, 0)) : F::tl(bt_r, kind_r);
	// (LAT: the loop's windows have 10 * sps + 1 or 20 * sps + 1 lags -- a constant lets the two little loops below unroll)
	const int win = LAT ? 3 : (w_p < 3 ? w_p : 3);
	const bool w3 = !LAT && __ballot(win != 3) == 0;   // (wave-uniform) every row's energy window is three lags
	float toa_r = 0.f, p_pwr = 0.f;                    // pi4cxpsk.c:227-237: the best sequence so far
	int sid_r = -1;
	if constexpr (PART == 2) {
		toa_r = pre->cut->toa;
		sid_r = pre->cut->sid;
		avr_r = pre->cut->avr;
		avi_r = pre->cut->avi;
	}
	for (int sq = 0; sq < (PART == 2 ? 0 : (FAC ? 2 : 1)); sq++) {
		if (FAC)
			cr = L.corr + row * cw + sq * cwh;
		GMR1_STAMP(10);
		int mi;
		if constexpr (LAT) {
			// the one burst's <= 128 lags, two per lane: the largest window energy as a 32-bit maximum (a non-negative float's
			// bits order like its value) by four DPP steps and four v_readlane, then the LOWEST lag that has it from two ballots
			// -- what the 64-bit (energy, ~index) keys of the batch form decide, in a quarter of the instructions
			constexpr int NH = SPS == 4 ? 2 : 3;              // 81 lags at sps 4, <= 161 at sps 8
			uint32_t eb2[NH];
			bool ok2[NH];
			uint32_t mx = 0;
#pragma unroll
			for (int h = 0; h < NH; h++) {
				const int m = lane + 64 * h;
				ok2[h] = m + win <= w_p;
				float e = 0.f;
				for (int k = 0; k < win; k++) {
					const float c = cr[ok2[h] ? m + k : 0];
					e += c * c;
				}
				eb2[h] = ok2[h] ? __builtin_bit_cast(uint32_t, e) : 0u;
				mx = eb2[h] > mx ? eb2[h] : mx;
			}
			uint32_t o;
			o = dpp<0xB1>(mx); mx = o > mx ? o : mx;
			o = dpp<0x4E>(mx); mx = o > mx ? o : mx;
			o = dpp<0x141>(mx); mx = o > mx ? o : mx;
			o = dpp<0x140>(mx); mx = o > mx ? o : mx;
			const uint32_t m01 = max((uint32_t)__builtin_amdgcn_readlane((int)mx, 0), (uint32_t)__builtin_amdgcn_readlane((int)mx, 16));
			const uint32_t m23 = max((uint32_t)__builtin_amdgcn_readlane((int)mx, 32), (uint32_t)__builtin_amdgcn_readlane((int)mx, 48));
			const uint32_t M = max(m01, m23);
			mi = 0;
#pragma unroll
			for (int h = NH - 1; h >= 0; h--) {
				const unsigned long long bh = __ballot(ok2[h] && eb2[h] == M);
				if (bh)
					mi = 64 * h + __builtin_ctzll(bh);
			}
		} else {
		if (w3) {
			// every row's window energy is over three lags (any search window of >= 3 lags): per lane the best of its lags
			// m = col, col + 16, ... in ascending order (a later one must be strictly larger), then over the row the largest
			// energy as a 32-bit maximum (a non-negative float's bits order like its value) and the LOWEST lag that has it --
			// what the 64-bit (energy, ~lag) keys below decide, in a third of the instructions
			uint32_t be = 0;
			int bl = 0x7fffffff;
			for (int m = col; m + 3 <= w_p; m += 16) {
				const float c0 = cr[m], c1 = cr[m + 1], c2 = cr[m + 2];
				float e = c0 * c0;                           // (0 + c0^2 is c0^2)
				e += c1 * c1;
				e += c2 * c2;
				const uint32_t eb = __builtin_bit_cast(uint32_t, e);
				const bool better = bl == 0x7fffffff || eb > be;
				be = better ? eb : be;
				bl = better ? m : bl;
			}
			uint32_t mx = be, o;
			o = dpp<0xB1>(mx); mx = o > mx ? o : mx;
			o = dpp<0x4E>(mx); mx = o > mx ? o : mx;
			o = dpp<0x141>(mx); mx = o > mx ? o : mx;
			o = dpp<0x140>(mx); mx = o > mx ? o : mx;
			uint32_t lo = (bl != 0x7fffffff && be == mx) ? (uint32_t)bl : 0x7fffffffu;
			o = dpp<0xB1>(lo); lo = o < lo ? o : lo;
			o = dpp<0x4E>(lo); lo = o < lo ? o : lo;
			o = dpp<0x141>(lo); lo = o < lo ? o : lo;
			o = dpp<0x140>(lo); lo = o < lo ? o : lo;
			mi = lo == 0x7fffffffu ? -1 : (int)lo;
		} else {
		unsigned long long key = 0;
		for (int m = col; m + win <= w_p; m += 16) {
			float e = 0.f;
			for (int k = 0; k < win; k++) {
				const float c = cr[m + k];
				e += c * c;
			}
			const unsigned long long kk = ((unsigned long long)__builtin_bit_cast(uint32_t, e) << 32) | (uint32_t)(~m);
			key = kk > key ? kk : key;
		}
		key = row_max_u64<1>(key);
		key = row_max_u64<2>(key);
		key = row_max_u64<4>(key);
		key = row_max_u64<8>(key);
		mi = (int)(~(uint32_t)key);
		}
		}
		if (mi < 0 || mi + win > w_p)
			mi = 0;
		GMR1_STAMP(11);
		int p = mi;
		if (w3 || LAT) {
			float pe = -1.f;
#pragma unroll
			for (int k = 0; k < 3; k++) {
				const float c = cr[mi + k];
				const float e = c * c;
				const bool up = e > pe;
				pe = up ? e : pe;
				p = up ? mi + k : p;
			}
		} else {
			float pe = -1.f;
			for (int k = 0; k < win; k++) {
				const float c = cr[mi + k];
				const float e = c * c;
				if (e > pe) { pe = e; p = mi + k; }
			}
		}
		GMR1_STAMP(8);
		if constexpr (LAT && PART == 1) {
			if (co_on) {
				// the helper wave starts on the sync-symbol terms of round(toa) = p - 1, p, p + 1
				co_p = __builtin_amdgcn_readfirstlane(p);
				if (lane == 0) {
					pre->co->p = co_p;
					lds_post(&pre->co->p_id, pre->co_id);
				}
			}
		}
		// interpolated correlation at `pos` (lanes 0-7 of the row) and at `pos + 2` (lanes 8-15): same
		// fractional part, so the same 21 weights; lane sub = col & 7 holds taps k = 3 sub - 10 + {0,1,2}
		const int ipt = col >> 3, isub = col & 7;
		// what the evaluations below share: the lane's first tap, as an index offset and as the float the weight needs, the
		// sign bit sin(pi (k - f)) = -(-1)^k sin(pi f) sets for an even first tap, whether the lane has taps at all (sub 7
		// would hold k = 11, 12, 13) and the one bound an array index has to be checked against: i = ib + k with |k| <= 10
		// lies in [max(ib - 10, 0), min(ib + 11, w - 1)) exactly if 0 <= i < w - 1
		const int k0 = 3 * isub - 10;
		const float kf0 = (float)k0, kf1 = (float)(k0 + 1), kf2 = (float)(k0 + 2);
		const uint32_t sgn0 = (isub & 1) ? 0u : 0x80000000u;
		const bool has_taps = isub != 7;
		const uint32_t wlim = (uint32_t)(w_p - 1);
		// -> acc: the lane's own half-row sum (at `pos` in lanes 0-7, at `pos + 2` in lanes 8-15), oth: the other half's
		// (guard: the weight 1 within 0.01 of a whole tap, osmo_sinc's small-argument case.  The walk's levels stand on odd
		// multiples of 1/2 ... 1/256, at least pi / 256 = 0.0123 from every tap: only the evaluation at the point reached --
		// a multiple of 1/512 -- can come that close, so the eight evaluations of the walk go without the two compares and the
		// select per tap)
		auto interp2 = [&](float pos, float &acc_o, float &oth_o, auto guard_tag) {
			constexpr bool GUARD = decltype(guard_tag)::value;
			const float fl = floorf(pos);
			const int ib = (int)fl + 2 * ipt;
			const float f = pos - fl;
			const float S = __builtin_amdgcn_sinf(0.5f * f);       // sin(pi f)
			float acc = 0.f;
			if (SMALL && w_p <= 8) {
				// A search window of at most eight lags (NT3: seven) has at most seven correlation values to interpolate
				// over: lane sub takes the ONE value cr[sub] with the weight of tap k = sub - ib, instead of three of the
				// 21 tap places of which most lie outside the array (a third of the instructions of the timing rows).
				int b = ib - 10, e = ib + 11;
				if (b < 0) b = 0;
				if (e >= w_p) e = w_p - 1;
				const int k = isub - ib;
				const float sg = (k & 1) ? S : -S;
				const float xx = kPif * ((float)k - f);
				const float wgt = (!GUARD || xx >= 0.01f || xx <= -0.01f) ? sg * __builtin_amdgcn_rcpf(xx) : 1.0f;
				const bool valid = isub >= b && isub < e;          // (|k| <= 10 follows from b and e)
				const float c = cr[isub];
				acc = valid ? c * wgt : 0.0f;
			} else {
				// (the three values are read whether or not they lie in the array -- an LDS address outside it is harmless --
				// and dropped by the select)
				const int i0 = ib + k0;
				const float *cp = cr + i0;
				const float c0 = cp[0], c1 = cp[1], c2 = cp[2];
				const float S0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, S) ^ sgn0);
				const float x0 = kPif * (kf0 - f), x1 = kPif * (kf1 - f), x2 = kPif * (kf2 - f);
				const float w0 = (!GUARD || x0 >= 0.01f || x0 <= -0.01f) ? S0 * __builtin_amdgcn_rcpf(x0) : 1.0f;
				const float w1 = (!GUARD || x1 >= 0.01f || x1 <= -0.01f) ? -S0 * __builtin_amdgcn_rcpf(x1) : 1.0f;
				const float w2 = (!GUARD || x2 >= 0.01f || x2 <= -0.01f) ? S0 * __builtin_amdgcn_rcpf(x2) : 1.0f;
				const bool v0 = has_taps && (uint32_t)i0 < wlim;
				const bool v1 = has_taps && (uint32_t)(i0 + 1) < wlim;
				const bool v2 = has_taps && (uint32_t)(i0 + 2) < wlim;
				acc += v0 ? c0 * w0 : 0.0f;
				acc += v1 ? c1 * w1 : 0.0f;
				acc += v2 ? c2 * w2 : 0.0f;
			}
			acc += row_xorf<1>(acc);
			acc += row_xorf<2>(acc);
			acc += row_xorf<4>(acc);
			acc_o = acc;
			oth_o = row_xorf<8>(acc);
		};
		float early = (float)p - 1.0f, incr = 0.5f;
		bool active = true;
		float toa_s, pk_s;
		if constexpr (LAT) {
			// THREE levels of the bisection per evaluation (the eight groups of 8 lanes each interpolate the correlation at one
			// candidate position and two samples later -- same fractional part, same 21 weights, lane sub holding taps
			// k = 3 sub - 10 + {0,1,2}): group 0 at the current point, groups 1 / 2 where the search goes if the early / the
			// late side wins, groups 3..6 one level further down.  The candidates are formed by the float operations the
			// level-by-level walk performs and summed in its order, so the decisions are its decisions; the walk itself is
			// scalar work on two ballots.  Nine levels = three evaluations, then the peak value at the point reached.
			const int grp = lane >> 3, isub8 = lane & 7;
			// (index checks and tap signs as in interp2 above: 0 <= i < w - 1 is the whole bound, one XOR sets the sign)
			const int k08 = 3 * isub8 - 10;
			const float kg0 = (float)k08, kg1 = (float)(k08 + 1), kg2 = (float)(k08 + 2);
			const uint32_t sgn8 = (isub8 & 1) ? 0u : 0x80000000u;
			const bool taps8 = isub8 != 7;
			auto interp_pair = [&](float pos, float &se, float &sl) {
				const float fl = floorf(pos);
				const int ib = (int)fl;
				const float f = pos - fl;
				const float S = __builtin_amdgcn_sinf(0.5f * f);       // sin(pi f); sin(pi (k - f)) = -(-1)^k sin(pi f)
				const int i0 = ib + k08;
				const float *cp = cr + i0;
				// early sum over i0 + {0, 1, 2}, late sum two lags further up: five values, whether or not they lie in the array
				const float c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3], c4 = cp[4];
				const float S0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, S) ^ sgn8);
				const float x0 = kPif * (kg0 - f), x1 = kPif * (kg1 - f), x2 = kPif * (kg2 - f);
				const float w0 = (x0 >= 0.01f || x0 <= -0.01f) ? S0 * __builtin_amdgcn_rcpf(x0) : 1.0f;
				const float w1 = (x1 >= 0.01f || x1 <= -0.01f) ? -S0 * __builtin_amdgcn_rcpf(x1) : 1.0f;
				const float w2 = (x2 >= 0.01f || x2 <= -0.01f) ? S0 * __builtin_amdgcn_rcpf(x2) : 1.0f;
				float ae = 0.f, al = 0.f;
				ae += (taps8 && (uint32_t)i0 < wlim) ? c0 * w0 : 0.0f;
				al += (taps8 && (uint32_t)(i0 + 2) < wlim) ? c2 * w0 : 0.0f;
				ae += (taps8 && (uint32_t)(i0 + 1) < wlim) ? c1 * w1 : 0.0f;
				al += (taps8 && (uint32_t)(i0 + 3) < wlim) ? c3 * w1 : 0.0f;
				ae += (taps8 && (uint32_t)(i0 + 2) < wlim) ? c2 * w2 : 0.0f;
				al += (taps8 && (uint32_t)(i0 + 4) < wlim) ? c4 * w2 : 0.0f;
				ae += row_xorf<1>(ae);
				ae += row_xorf<2>(ae);
				ae += row_xorf<4>(ae);
				al += row_xorf<1>(al);
				al += row_xorf<2>(al);
				al += row_xorf<4>(al);
				se = ae;
				sl = al;
			};
			// The loop's front (PART 1) stops after the FIRST evaluation.  Everything it hands on depends on the timing through
			// the pick round(toa) alone (pi4cxpsk.c:292-295; the feedback is align += round(toa) - e_toa, gmr1_rx.c:782; no
			// record carries toa), and round(toa) is settled by the first two levels: the walk starts on the whole lag p - 1
			// and steps by 1/2, 1/4, ... 1/512, so after two levels toa = early + 1 is p -+ 1/4 or p -+ 3/4 and the seven levels
			// left move it by less than 1/4 in all -- never across p -+ 1/2; a level that ties ends the walk in both forms.
			constexpr int kEvals = PART == 1 ? 1 : 3;
	#pragma unroll 1
			for (int it = 0; it < kEvals; it++) {
				const float half = incr * 0.5f, quarter = incr * 0.25f;
				float pos = early;
				if (grp == 1) {
					pos = early - incr;
				} else if (grp == 2) {
					pos = early + incr;
				} else if (grp >= 3 && grp <= 6) {
					const float a1 = grp < 5 ? early - incr : early + incr;
					pos = (grp & 1) ? a1 - half : a1 + half;           // 3: - -, 4: - +, 5: + -, 6: + +
				}
				float se, sl;
				interp_pair(pos, se, sl);
				const float ee = se * se, le = sl * sl;
				const unsigned long long m_neg = __ballot(ee > le), m_pos = __ballot(ee < le);
				auto dec = [&](int gq) -> int { return ((m_neg >> (8 * gq)) & 1ull) ? -1 : (((m_pos >> (8 * gq)) & 1ull) ? 1 : 0); };
				const int d0 = dec(0);
				if (d0 == 0) break;
				early = d0 < 0 ? early - incr : early + incr;
				const int d1 = dec(d0 < 0 ? 1 : 2);
				if (d1 == 0) break;
				early = d1 < 0 ? early - half : early + half;
				const int d2 = dec(3 + (d0 > 0 ? 2 : 0) + (d1 > 0 ? 1 : 0));
				if (d2 == 0) break;
				early = d2 < 0 ? early - quarter : early + quarter;
				incr *= 0.125f;
			}
			toa_s = early + 1.0f;
			GMR1_STAMP(9);
			float pa, po;
			interp2(toa_s, pa, po, std::true_type{});
			pk_s = ipt ? po : pa;
		} else {
			// early energy ee > late energy le: early -= incr, ee < le: early += incr.  Lanes 8-15 hold the late sum as their
			// own, so they see the two energies swapped and step by -incr instead: the same decisions without the selects
			float sincr = ipt ? -incr : incr;
			{
				// the first level sits on a whole lag (early = p - 1): sin(pi f) = 0, so tap 0 has the weight 1 and the twenty
				// others (+-)0 -- the two sums ARE cr[p - 1] and cr[p + 1] (0 outside the array), no interpolation to run
				const int io = p - 1 + 2 * ipt;
				const float c = cr[io];
				const float sa = (uint32_t)io < wlim ? c : 0.0f;
				const float so = row_xorf<8>(sa);
				const float ea = sa * sa, eo = so * so;
				if (ea > eo) early -= sincr;
				else if (ea < eo) early += sincr;
				else active = false;
				sincr *= 0.5f;
			}
	#pragma unroll 1
			for (int it = 1; it < 9; it++) {              // incr = 0.25 ... 1/512 (> 1/1024)
				float sa, so;
				interp2(early, sa, so, std::false_type{});
				const float ea = sa * sa, eo = so * so;
				if (active) {
					if (ea > eo) early -= sincr;
					else if (ea < eo) early += sincr;
					else active = false;
				}
				sincr *= 0.5f;
			}
			toa_s = early + 1.0f;
			float pa, po;
			interp2(toa_s, pa, po, std::true_type{});
			pk_s = ipt ? po : pa;
		}
		pk_s = pk_s * __builtin_amdgcn_rcpf((float)tl_p);
		if (pk_s * pk_s > p_pwr) {                      // needs strictly more than what is there (0 at first)
			p_pwr = pk_s * pk_s;
			toa_r = toa_s;
			sid_r = sq;
		}
	}
	const bool found_r = sid_r >= 0;
	if constexpr (PART != 2) {
		GMR1_STAMP(3);
		if (a.dbg_stop == 3) return;
	}
	const int d_r = (int)roundf(toa_r);

	// QX: whose speculated pick held (per row; a burst that was not found has no pass 2 at all)
	const bool hit_r = QX && found_r && d_r == (int)(zm_q >> 24) && a.dbg_stop != 101;      // (101: experiment, every burst takes the old route)
#ifdef GMR1_HIP_PROFILE
	if constexpr (QX) {
		// (counted only when asked for, GMR1_HIP_DBG_STOP=105: fifteen thousand atomic adds to one word per launch cost the profiling
		// build a tenth of the kernel's time -- and made an experiment that switched them off with the mis-speculation look like
		// a 10 % gain)
		if (a.dbg_stop == 105 && col == 0 && row_live && found_r && !hit_r)
			atomicAdd(&g_prof_miss, 1);
	}
#endif

	// pass-2 operands of a burst (its 234 symbols at stride sps from sample d, re-read from
	// L2 / Infinity Cache): fetched two bursts ahead of their use
	struct Sym4 { float2 x[NSYM]; int ok; };
	auto fetch = [&](int q, Sym4 &o) {
		const int g = g0 + q;
		o.ok = 0;
#pragma unroll
		for (int r = 0; r < NSYM; r++) o.x[r] = make_float2(0.f, 0.f);
		if (q < 0 || q >= 4 || g >= n_end || a.dbg_stop == 100)     // 100: timing experiment, pass 2 without its re-read
			return;
		const int src = 16 * q;
		const int kind = __builtin_amdgcn_readlane(kind_r, src);
		const DevBurst &bt = c_types[GEN ? a.fixed_type : (kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH)];
		const int in_len = __builtin_amdgcn_readlane(in_len_r, src);
		const int d = __builtin_amdgcn_readlane(d_r, src);
		const float2 *__restrict__ in = a.iq + op_off(g);
		const int blen = F::len(bt);
		if constexpr (PL) {
			// the kept samples d, d + 4, ... are consecutive in plane (offset + d) & 3
			const long long sr = (long long)io.offset[g] + d;
			in = a.iq + (sr & 3) * a.plane_stride + (sr >> 2);
		}
#pragma unroll
		for (int r = 0; r < NSYM; r++) {
			const int i = lane + 64 * r;
			const int j = i * sps + d;
			if (i < blen && j >= 0 && j < in_len) {
				if (LAT && PART == 2)
					o.x[r] = pre->win_r[j];
				else
					o.x[r] = PL ? in[i] : in[j];
				o.ok |= 1 << r;
			}
		}
	};
	// pass 2 takes the bursts last-read first: burst 3's window was streamed in a few microseconds ago and may still be
	// in this XCD's L2, burst 0's is long gone from it
	// (two bursts ahead of their use: asking for all four at once was measured and is no faster)
	Sym4 first, second;
	int first_q = -1;                                  // QX: the mis-speculated burst whose samples `first` holds

	// =========================== rows: sync symbols, frequency, phase ===========================
	const int nbits_r = F::nbits(bt_r);
	const int nch_r = F::nch(bt_r);
	const int tl_r = F::tl(bt_r, kind_r);
	float ffe_r = 0.f, psi_r = 0.f;
	if constexpr (PART == 2) {
		ffe_r = pre->cut->ffe;
		psi_r = pre->cut->psi;
		fetch(0, first);
	} else if constexpr (LAT) {
		if constexpr (PART == 0)
			fetch(0, first);            // the one burst's kept samples travel during the whole sync-term phase
		bool have = false;
		if constexpr (PART == 1) {
			if (co_on) {
				// what the helper wave made of the candidate that round(toa) turned out to be
				const int k = __builtin_amdgcn_readfirstlane(d_r) - (co_p - 1);
				if ((unsigned)k < 3u && lds_wait_eq(&pre->co->s_id, pre->co_id)) {
					ffe_r = pre->co->ffe[k];
					psi_r = pre->co->psi[k];
					have = true;
				}
			}
		}
		if (!have)
			lat_sync_terms(xst_lat, kind_r, d_r, row_live, fs_r, sps, w_r, in_len_r, col, ffe_r, psi_r);
	} else {
		// lane col holds sync symbols n = col and n = col + 16 (< tl <= 32); their samples are asked for first (loads
		// come back in order), then pass 2's
		float2 t0[NSH], xr[NSH];
		int chn[NSH], spos[NSH], idxv[NSH], nnv[NSH];
#pragma unroll
		for (int h = 0; h < NSH; h++) {
			const int n = col + 16 * h;
			t0[h] = xr[h] = make_float2(0.f, 0.f);
			chn[h] = -1;
			spos[h] = idxv[h] = nnv[h] = 0;
			if (n < tl_r && row_live) {
				int ch = 0, base = 0, cum = 0, wb = 0;
				for (int c = 0; c < nch_r - 1; c++) {
					cum += F::clen(bt_r, kind_r, c);
					if (n >= cum) { base = cum; ch = c + 1; wb += F::clen(bt_r, kind_r, c) * sps + w_r - 1; }
				}
				const int nn = n - base;
				const int sp = F::cpos(bt_r, ch) + nn;
				const int idx = sp * sps + d_r;
				if (idx >= 0 && idx < in_len_r) {
					// LAT, the wave's one burst: its sync-chunk windows are still staged (normalised, which no angle
					// below notices) -- no second trip to L2
					if constexpr (PL) {
						// sample sp sps + d of the window = place sp of the plane the kept samples lie in
						const long long sr = (long long)(row_live ? io.offset[g_row] : 0) + d_r;
						xr[h] = (a.iq + (sr & 3) * a.plane_stride + (sr >> 2))[sp];
					} else if (QX && hit_r) {
						xr[h] = xs_q[h];            // taken from the staged windows when the pick was speculated (normalised)
					} else
					xr[h] = LAT ? xst_lat[wb + nn * sps + d_r] : in_r[idx];
				}
				chn[h] = ch;
				spos[h] = sp;
				idxv[h] = idx;
				nnv[h] = nn;
			}
		}
		if constexpr (LAT) {
			if constexpr (PART == 0)
				fetch(0, first);        // the one burst's kept samples travel during the whole sync-term phase
		} else if constexpr (QX) {
			// the (rare) burst whose pick is not the speculated one: its kept samples travel during the rest of this phase (the last
			// such burst of the wave; a second one in the same wave waits for its samples in pass 2)
			const unsigned long long mb = __ballot(found_r && !hit_r && row_live);       // (row-uniform: bit 16 q speaks for burst q)
			const unsigned missm = (unsigned)((mb & 1ull) | ((mb >> 15) & 2ull) | ((mb >> 30) & 4ull) | ((mb >> 45) & 8ull));
			first_q = missm ? 31 - __builtin_clz(missm) : -1;
			if (first_q >= 0)
				fetch(first_q, first);
		} else {
			fetch(3, first);
			fetch(2, second);
		}
#pragma unroll
		for (int h = 0; h < NSH; h++) {
			if (chn[h] >= 0) {
				const int idx = idxv[h], ch = chn[h], nn = nnv[h], sp = spos[h];
				float2 x = xr[h];
				if (!LAT && idx >= 0 && idx < in_len_r && !(QX && hit_r)) {
					x.x -= avr_r;
					x.y -= avi_r;
					if constexpr (QL) {
						// as the staged windows are normalised (a scale no angle below notices; the same numbers whichever
						// way the sample came)
						x.x *= inv_r;
						x.y *= inv_r;
					}
				}
				float s, c;
				sincos_fast(fs_r * (float)idx, s, c);
				x = cmul(x, make_float2(c, s));
				t0[h] = conj_ref_mul(nbits_r, F::sym(bt_r, kind_r, FAC ? (sid_r > 0 ? 1 : 0) : 0, ch, nn, col + 16 * h), x);
				chn[h] = ch;
				spos[h] = sp;
			}
		}
		// chunk sums (pi4cxpsk.c:381-389); the window scale 1/sigma is irrelevant to every angle
		constexpr int NCS = GEN ? 4 : 3;                // chunks a format can have (the fused formats: three)
		if constexpr (!SMALL) if (nch_r > 1) {
			float sumr[NCS], sumi[NCS];
#pragma unroll
			for (int c = 0; c < NCS; c++) {
				const float pr = (chn[0] == c ? t0[0].x : 0.f) + (chn[1] == c ? t0[1].x : 0.f);
				const float pi = (chn[0] == c ? t0[0].y : 0.f) + (chn[1] == c ? t0[1].y : 0.f);
				sumr[c] = row_sum(pr);
				sumi[c] = row_sum(pi);
			}
			float f = 0.f;
#pragma unroll
			for (int i = 1; i < NCS; i++) {
				if (i < nch_r) {
					const float ppos = (float)F::cpos(bt_r, i - 1) + (float)F::clen(bt_r, kind_r, i - 1) / 2.0f;
					const float cpos = (float)F::cpos(bt_r, i) + (float)F::clen(bt_r, kind_r, i) / 2.0f;
					const float re = sumr[i] * sumr[i - 1] - sumi[i] * (-sumi[i - 1]);
					const float im = sumr[i] * (-sumi[i - 1]) + sumi[i] * sumr[i - 1];
					f += atan2_fast(im, re) / (cpos - ppos);
				}
			}
			ffe_r = f / (float)(nch_r - 1);
		}
		// carrier phase of the frequency-corrected sync symbols (pi4cxpsk.c:415-433,574-575)
		float tr = 0.f, ti = 0.f;
#pragma unroll
		for (int h = 0; h < NSH; h++) {
			float2 tt = t0[h];
			if (ffe_r != 0.0f) {
				float s, c;
				sincos_fast(-ffe_r * (float)spos[h], s, c);
				tt = cmul(tt, make_float2(c, s));
			}
			tr += tt.x;
			ti += tt.y;
		}
		psi_r = atan2_fast(row_sum(ti), row_sum(tr));
	}
	if constexpr (PART != 2) {
	GMR1_STAMP(4);
	if (a.dbg_stop == 5) return;

	// per-burst results
	if (col == 0 && row_live && (PART != 1 || io.rv)) {
		const int rv = found_r ? 0 : -1;
		io.rv[g_row] = rv;
		if (io.sync_id) io.sync_id[g_row] = found_r ? sid_r : -1;
		if (io.toa) io.toa[g_row] = found_r ? toa_r : 0.f;
		if (io.freq_err) io.freq_err[g_row] = found_r ? ffe_r : 0.f;
	}
	}
	if constexpr (PART == 1) {
		// (row 0 holds the burst: lane 0's values, for every lane)
		auto first_f = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
		pre->out = {__builtin_amdgcn_readfirstlane(found_r ? 1 : 0), __builtin_amdgcn_readfirstlane(d_r), __builtin_amdgcn_readfirstlane(sid_r),
		            first_f(found_r ? toa_r : 0.f), first_f(found_r ? ffe_r : 0.f), first_f(psi_r), first_f(avr_r), first_f(avi_r)};
		if (lane == 0)
			*pre->cut = {found_r ? 1 : 0, d_r, sid_r, toa_r, ffe_r, psi_r, avr_r, avi_r};
		return;
	}

	// =========================== pass 2: soft symbols / soft bits ===========================
	// the soft-bit table overlays the pass-1 data (fused: where the branch metrics go after pass 2; demodulation
	// only: behind the soft-bit rows, which themselves overlay the correlation the rows above were reading)
	const unsigned char *lut = lds_raw + (GEN ? 4 * EBROW : 0);
	if constexpr (LAT) {
		lut = pre->lut;                             // the work-group's resident copy
		WSYNC();
	} else {
		// the soft-bit table (2 KB, L2-resident) goes to LDS now that the rows are done with the pass-1 data it overlays
		unsigned char *const lutw = lds_raw + (GEN ? 4 * EBROW : 0);
		const uint4 *__restrict__ lut_src = reinterpret_cast<const uint4 *>(FAC ? g_sb_lut1.v : g_sb_lut.v);
		const uint4 lut_a = lut_src[lane], lut_b = lut_src[lane + 64];
		WSYNC();
		reinterpret_cast<uint4 *>(lutw)[lane] = lut_a;
		reinterpret_cast<uint4 *>(lutw)[lane + 64] = lut_b;
		WSYNC();
	}
	int row_ok = 0, row_chain = 0;
	Sym4 nxt1, nxt2;
	if constexpr (!QX) { nxt1 = first; nxt2 = second; }
#pragma unroll
	for (int q = 3; q >= 0; q--) {
		const int g = g0 + q;
		Sym4 cur;
		if constexpr (LAT) {
			if (q != 0)
				continue;
			cur = first;
		} else if constexpr (QX) {
			cur.ok = 0;
#pragma unroll
			for (int r = 0; r < NSYM; r++) cur.x[r] = make_float2(0.f, 0.f);
		} else {
			cur = nxt1;
			nxt1 = nxt2;
			fetch(q - 2, nxt2);         // the samples of burst q - 2 travel while q and q - 1 are worked on
		}
		if (g >= n_end)
			continue;
		const int src = 16 * q;
		const bool hit = QX && __builtin_amdgcn_readlane((int)hit_r, src) != 0;
		if constexpr (QX) {
			if (!hit && __builtin_amdgcn_readlane((int)found_r, src) != 0) {
				// the pick is not the speculated one: this burst's kept samples come from memory
				if (q == first_q)
					cur = first;
				else
					fetch(q, cur);
			}
		}
		const bool found = __builtin_amdgcn_readlane((int)found_r, src) != 0;
		const int kind = __builtin_amdgcn_readlane(kind_r, src);
		const int type = GEN ? a.fixed_type : (kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH);
		const DevBurst &bt = c_types[type];
		const int d = __builtin_amdgcn_readlane(d_r, src);
		const float fs = lane_val(fs_r, src);
		const float rps = -lane_val(ffe_r, src);
		const float psi = lane_val(psi_r, src);
		const float avr = lane_val(avr_r, src), avi = lane_val(avi_r, src);
		const int blen = F::len(bt), nbits = F::nbits(bt);
		float *gss = io.ssyms ? io.ssyms + (size_t)g * a.ssyms_stride : nullptr;
		int8_t *eb = L.eb + q * (GEN ? EBROW : 432);
		row_chain |= kind << q;
		if (!found) {
			if (io.ebits)
				for (int i = lane; i < a.ebits_stride; i += 64)
					io.ebits[(size_t)g * a.ebits_stride + i] = 0;
			if constexpr (GEN)                                  // (a fused decoder reads the LDS row)
				for (int i = lane; i < EBROW / 4; i += 64)
					reinterpret_cast<uint32_t *>(eb)[i] = 0;
			if (gss)
				for (int i = lane; i < blen; i += 64)
					gss[i] = 0.f;
			continue;
		}
		row_ok |= 1 << q;
		// where the soft bits of symbol i go (position of the symbol's bits among the e-bits, -1: sync / guard)
		int ordv[NSYM];
#pragma unroll
		for (int r = 0; r < NSYM; r++) {
			const int i = lane + 64 * r;
			ordv[r] = i < blen ? bt.ord_of_sym[i] : -1;
		}
		// phase of symbol i in TURNS: arg(x_i) + fs (i sps + d) + rps i - psi  =  arg(x_i) + A i + B
		// (pi4cxpsk.c:351-371 derotation, :574-588 frequency / phase correction, folded into one fma), carried
		// scaled by 2048 (exact): floor(2048 th) & 2046 is the byte offset of the phase's cell in the soft-bit table
		const float kInv2Pi = 0.159154943091895336f;
		const float At = (fs * (float)sps + rps) * kInv2Pi;
		float Bt = (fs * (float)d - psi) * kInv2Pi;
		Bt -= rintf(Bt);
		const float A2 = At * 2048.0f, B2 = Bt * 2048.0f;
		const float scale = (float)(1 << nbits);
#pragma unroll
		for (int r = 0; r < NSYM; r++) {
			const int i = lane + 64 * r;
			if (i >= blen)
				continue;
			// (a sample outside the window counts as 0 + 0j, not as minus the mean: the subtraction runs for every lane
			// and the zero test knows which lanes had a sample)
			float2 x = cur.x[r];
			x.x -= avr;
			x.y -= avi;
			float at;
			bool zero;
			if (QX && hit) {
				at = th_q[QX ? q : 0][r];
				zero = ((zm_q >> (4 * q + r)) & 1u) != 0;
			} else {
				at = atan2_turns(x.y, x.x);
				zero = !(cur.ok & (1 << r)) || (x.x == 0.0f && x.y == 0.0f);   // cargf(0) = 0
			}
			const float th2 = fmaf(A2, (float)i, fmaf(at, 2048.0f, B2));
			if (gss) {
				float th = th2 * (1.0f / 2048.0f);
				th -= rintf(th);
				gss[i] = zero ? 0.0f : th * scale;
			}
			const int ord = ordv[r];
			if (ord >= 0) {
				int cell;
				asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(cell) : "v"(th2));
				cell = zero ? 0 : (cell & 2046);
				if constexpr (FAC)
					eb[ord] = (int8_t)lut[cell];                   // one soft bit per symbol
				else
					*reinterpret_cast<uint16_t *>(eb + 2 * ord) = *reinterpret_cast<const uint16_t *>(lut + cell);
			}
		}
		if (io.ebits) {
			WSYNC();
			const int neb = bt.ebits;
			int8_t *ge = io.ebits + (size_t)g * a.ebits_stride;
			for (int i = lane; i < a.ebits_stride; i += 64)
				ge[i] = i < neb ? eb[i] : (int8_t)0;
		}
	}
	if (GEN || (a.dbg_stop && a.dbg_stop < 7))
		return;
	GMR1_STAMP(5);

	// =========================== rows: layer 1 ===========================
	WSYNC();     // the window is dead: bm / surv / ubits overlay it
	DecPre dpre;
	if constexpr (LAT) {
		// one burst (row 0; what the other rows decode is never looked at), tables from the work-group's LDS copies
		const int chain = row_chain & 1;
		const bool ok = (row_ok & 1) != 0;
#pragma unroll
		for (int it = 0; it < 4; it++) {
			const int k = lane + 64 * it;
			if (k < kSteps12) {
				const uint32_t stw = pre->steps[chain * kSteps12 + k];
				const uint32_t ia = (uint32_t)(uint8_t)L.eb[stw & 0x3ffu] | ((stw >> 2) & 0x100u);
				const uint32_t ib = (uint32_t)(uint8_t)L.eb[(stw >> 16) & 0x3ffu] | ((stw >> 18) & 0x100u);
				lat_expand_step(pre->vtab, k, ok ? pre->cost_a[ia] + pre->cost_b[ib] : 0u);
			}
		}
		dpre.dc = pre->dc; dpre.sy0 = pre->sy0; dpre.sy1 = pre->sy1;
#ifdef GMR1_HIP_PROFILE
		dpre.stamp = io.stamp;
#endif
	} else {
		branch_metrics4_k5_12<ACC>(L.eb, 432, row_ok, row_chain, L.bm, lane);
	}
	WSYNC();
	GMR1_STAMP(6);
	if (a.dbg_stop == 7)
		return;
	if constexpr (PART == 2)
		return;                                     // (the decoder runs on another wave, a round later: k_rx_chain_pipe)
	uint32_t syn, fae;
	if constexpr (LAT)
		decode1_k5_12_lat<ACC>(pre->vtab, L.surv, L.ubits, lane, syn, fae, &dpre);
	else
		decode4_k5_12<ACC>(L.bm, L.surv, L.ubits, lane, syn, fae, nullptr);
	GMR1_STAMP(7);
	if (col == 0 && row_live) {
		if ((row_ok >> row) & 1) {
			store_l2(io.l2 + (size_t)g_row * 24, L.ubits + row * 8);
			io.crc[g_row] = syn ? 1 : 0;
			io.conv[g_row] = (int32_t)fae;
		} else {

			uint32_t *l2w = reinterpret_cast<uint32_t *>(io.l2 + (size_t)g_row * 24);
#pragma unroll
			for (int i = 0; i < 6; i++)
				l2w[i] = 0;
			io.crc[g_row] = -1;
			io.conv[g_row] = 0;
		}
	}
}

// resident waves per SIMD each instantiation is compiled for: the headline one (windows <= 1024 samples at sps 4) fits six
// without spilling, the run-time-sps one five; the long windows (sps 8: 32 samples per lane) need the registers of three
// (the receive loop's instantiation, EN, keeps the window registers for the burst energy: five)
#ifndef GMR1_EXP_RX4_WAVES
#define GMR1_EXP_RX4_WAVES 6
#endif
template <int NPL, int SPS, bool EN = false>
constexpr int kRx4Waves = NPL > 16 ? 3 : ((SPS == 4 && !EN) ? GMR1_EXP_RX4_WAVES : 5);   // (seven: 72 VGPRs; measured again in round 4 with the LDS of seven -- no gain interleaved, spills and 4 % slower planar)

template <int NPL, int SPS, bool ACC = false, bool EN = false, bool PL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kRx4Waves<NPL, SPS, EN>, kRx4Waves<NPL, SPS, EN>)))
void k_rx4(RxArgs a, int stage_samples, int cw, int bpw)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	// bpw bursts per wavefront: 4 for throughput; 1 when the batch is too small to fill the machine anyway,
	// which shortens the critical path of a wave to a quarter
	int g0 = blockIdx.x * bpw;
	int n_end = min(a.n, g0 + bpw);
	if (EN && a.seg_count) {
		// bursts listed in segments with unused slots at each segment's end (the receive loop's CCCH lists); with seg_first a
		// launch takes one time slice of every list: block -> (segment, group of four behind the slice's first slot)
		int sg = g0 / a.seg_stride;
		int lo = 0;
		if (a.seg_first) {
			if (a.seg_groups > 0) {
				sg = (int)blockIdx.x / a.seg_groups;
				lo = (a.seg_first[sg] + 3) & ~3;
				g0 = sg * a.seg_stride + lo + ((int)blockIdx.x % a.seg_groups) * bpw;
			} else {
				lo = (a.seg_first[sg] + 3) & ~3;
			}
		}
		const int base = sg * a.seg_stride;
		n_end = min(min(a.n, g0 + bpw), base + min(a.seg_count[sg], a.seg_stride));
		if (g0 >= n_end || g0 < base + lo)
			return;
	}
	const RxIo io = {
#ifdef GMR1_HIP_PROFILE
	                 nullptr,
#endif
	                 a.offset, a.kind, a.freq_shift, a.l2, a.crc, a.conv, a.rv, a.sync_id, a.toa, a.freq_err, a.energy,
	                 a.ebits, a.ssyms};
	rx4_body<NPL, SPS, false, false, false, ACC, EN, PL>(a, io, stage_samples, cw, g0, n_end, lds_raw, (int)threadIdx.x);
}

// demodulation only, one burst format per launch, four bursts per wavefront (rx4_body<..., GEN>)
template <int NPL, int SPS, bool FAC = false>
__global__ __launch_bounds__(64) void k_rx4g(RxArgs a, int stage_samples, int cw)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int g0 = blockIdx.x * 4;
	const RxIo io = {
#ifdef GMR1_HIP_PROFILE
	                 nullptr,
#endif
	                 a.offset, nullptr, a.freq_shift, nullptr, nullptr, nullptr, a.rv, a.sync_id, a.toa, a.freq_err, a.energy,
	                 a.ebits, a.ssyms};
	rx4_body<NPL, SPS, false, true, FAC>(a, io, stage_samples, cw, g0, min(a.n, g0 + 4), lds_raw, (int)threadIdx.x);
}

// NT3 speech bursts from samples to speech frames in one launch (what rx_tch3 does with a burst, gmr1_rx.c:551-587:
// gmr1_pi4cxpsk_demod, then gmr1_tch3_decode): the small-format demodulator of four bursts per wavefront, then the TCH3
// decoder (tch3_body.h) on each of the four while their soft bits are still in the wave's LDS rows -- no 212-byte trip to
// HBM and back per burst, one launch, and a compute unit always holds wavefronts in the traffic-bound half next to
// wavefronts in the issue-bound half.
// (resident waves per SIMD: the accelerated decoder's instantiation fits seven -- 72 VGPRs, 5 024 B of LDS -- and gains 2.4 %
// over six; the generic one needs the registers of six)
template <bool ACC>
constexpr int kRx4gTch3Waves = 7;
template <bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kRx4gTch3Waves<ACC>, kRx4gTch3Waves<ACC>))) void k_rx4g_tch3(RxArgs a, Tch3Args t, int stage_samples, int cw)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int g0 = blockIdx.x * 4;
	const int lane = (int)threadIdx.x;
	const int n_end = min(a.n, g0 + 4);
	const RxIo io = {
#ifdef GMR1_HIP_PROFILE
	                 nullptr,
#endif
	                 a.offset, nullptr, a.freq_shift, nullptr, nullptr, nullptr, a.rv, a.sync_id, a.toa, a.freq_err, a.energy,
	                 a.ebits, a.ssyms};
	constexpr int kRow = 216;                    // 212 soft bits + the zero a punctured position reads
	rx4_body<8, 4, false, true, false, false, true, false, kRow>(a, io, stage_samples, cw, g0, n_end, lds_raw, lane);
	// soft-bit rows of the four bursts at the front of the wave's LDS (kRow bytes apart), the decoder's tables behind them
	t3::Tch3Lds *S = reinterpret_cast<t3::Tch3Lds *>(lds_raw + 4 * kRow);
	WSYNC();
	t3::tch3_fill_locof(S, lane);
	if (lane < 4)
		*reinterpret_cast<uint32_t *>(lds_raw + lane * kRow + 212) = 0;      // byte 212: what a punctured position reads
	t3::Tch3Lane lc;
	t3::tch3_lane(lc, S, lane);
	WSYNC();
	for (int q = 0; q < 4 && g0 + q < n_end; q++) {
		t3::tch3_burst<ACC>(t, g0 + q, lane, reinterpret_cast<const int8_t *>(lds_raw + q * kRow), S, lc);
		WSYNC();
	}
}

// ---------------------------------------------------------------------------
// burst type detection (reference src/sdr/pi4cxpsk.c:617-682 gmr1_pi4cxpsk_detect):
// normalise once with the rotation of the first candidate type, run the sync search of
// every candidate, weight the power by 1/|e_toa - toa|, keep the strongest.
// ---------------------------------------------------------------------------
template <int NPL, int SPS>
__global__ __launch_bounds__(64) void k_detect(DetectArgs a, int max_in_len)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const Lds L = lds_carve(lds_raw, max_in_len, a.max_lags, false);
	const int g = blockIdx.x;
	const int sps = SPS ? SPS : a.sps;
	load_normalise<NPL>(a.iq + a.offset[g], a.in_len, L, lane);
	const float fsh = a.freq_shift ? a.freq_shift[g] : 0.0f;
	const float fs = (fsh - a.rot0) / (float)sps;
	const float e_toa = a.e_toa ? a.e_toa[g] : -1.0f;
	int p_id = -1, p_sid = -1, rv = 0;
	float p_toa = 0.f, p_pwr = 0.f;
	if (a.carry) {
		// a list of more than four candidates runs as several launches: pick up where the last one stopped
		rv = a.rv[g];
		p_id = a.bt_id[g]; p_sid = a.sync_id[g]; p_toa = a.toa[g]; p_pwr = a.best_pwr[g];
	}
	for (int id = 0; id < a.n_types && rv == 0; id++) {
		float toa, pwr;
		const int sid = sync_search<SPS>(a.types[id], a.in_len, a.sps, fs, L, lane, 0, toa, pwr);
		if (sid < 0) {
			rv = sid;
			break;
		}
		if (e_toa >= 0.0f)
			pwr = (float)((double)pwr / fabs((double)(e_toa - toa)));
		if (pwr > p_pwr) {
			p_id = a.first + id; p_sid = sid; p_pwr = pwr; p_toa = toa;
		}
	}
	if (lane == 0) {
		a.rv[g] = rv;
		if (a.bt_id) a.bt_id[g] = rv ? -1 : p_id;
		if (a.sync_id) a.sync_id[g] = rv ? -1 : p_sid;
		if (a.toa) a.toa[g] = rv ? 0.f : p_toa;
		if (a.best_pwr) a.best_pwr[g] = p_pwr;
	}
}

// ---------------------------------------------------------------------------
// modulation order estimate (reference src/sdr/pi4cxpsk.c:693-729 gmr1_pi4cxpsk_mod_order):
// w = v^2 / |v|^2 on the pi/4-derotated window; BPSK if |sum w|^2 >= |sum w^2|^2 / 2, else QPSK
// ---------------------------------------------------------------------------
template <int NPL>
__global__ __launch_bounds__(64) void k_mod_order(ModOrderArgs a, int max_in_len)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const Lds L = lds_carve(lds_raw, max_in_len, 0, false);
	const int g = blockIdx.x;
	load_normalise<NPL>(a.iq + a.offset[g], a.in_len, L, lane);
	WSYNC();
	const float fsh = a.freq_shift ? a.freq_shift[g] : 0.0f;
	const float fs = (fsh - (kPif / 4)) / (float)a.sps;
	float sbr = 0.f, sbi = 0.f, sqr = 0.f, sqi = 0.f;
	for (int i = lane; i < a.in_len; i += 64) {
		float2 v = L.x[i];
		if (fs != 0.0f) {
			float s, c;
			sincos_fast(fs * (float)i, s, c);
			v = cmul(v, make_float2(c, s));
		}
		const float nn = v.x * v.x + v.y * v.y;
		const float2 vv = cmul(v, v);
		const float2 w = make_float2(vv.x / nn, vv.y / nn);
		const float2 ww = cmul(w, w);
		sbr += w.x; sbi += w.y;
		sqr += ww.x; sqi += ww.y;
	}
	sbr = wave_sum(sbr); sbi = wave_sum(sbi);
	sqr = wave_sum(sqr); sqi = wave_sum(sqi);
	if (lane == 0) {
		const float pb = sbr * sbr + sbi * sbi;
		const float pq = sqr * sqr + sqi * sqi;
		a.order[g] = pb < (pq / 2.0f) ? 4 : 2;
	}
}

// ---------------------------------------------------------------------------
// The layer-1 chain under libosmocore's accelerated decoder on soft bits from OUTSIDE (they may hold -128, and two of
// those in one trellis step cost 256: one more than a byte lane of the branch-metric word takes).  Same packed-word
// butterfly, same windows, same survivor walk as decode4_k5_12<true>; the four costs of a step are 16-bit lanes of two
// words, formed here from the soft bits themselves.
// ---------------------------------------------------------------------------
template <int PH>
__device__ __forceinline__ uint32_t k5w_step(uint32_t w, const uint16_t *__restrict__ c4, uint32_t oo, uint32_t op)
{
	uint32_t p;
	if constexpr (PH == 0) p = dpp<0x128>(w);                // row_ror:8
	else if constexpr (PH == 1) p = dpp<0x141>(w);           // row_half_mirror: xor 7
	else if constexpr (PH == 2) p = dpp<0x4E>(w);            // quad_perm [2,3,0,1]
	else p = dpp<0xB1>(w);                                   // quad_perm [1,0,3,2]
	const uint32_t t1 = ((uint32_t)c4[oo] << 16) + w;
	const uint32_t t2 = ((uint32_t)c4[op] << 16) + p;
	return t1 < t2 ? t1 : t2;
}

__global__ __launch_bounds__(64) void k_l1_acc(L1Args a)
{
	__shared__ __align__(16) int8_t s_eb[4 * kEbRow];
	__shared__ __align__(16) uint2 s_bmw[4 * kSteps12];        // per step: costs of the coded words 00, 01 | 10, 11
	__shared__ __align__(16) uint64_t s_surv[kSteps12];
	__shared__ __align__(16) uint32_t s_ub[4 * 8];
	const int lane = threadIdx.x;
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	const int g0 = blockIdx.x * 4;
	const int neb = a.chain == kChainCcch ? 432 : 424;
	const int chain = a.chain == kChainCcch ? 1 : 0;

	for (int q = 0; q < 4; q++) {
		const int g = g0 + q;
		if (g < a.n) {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)g * neb);
			uint32_t *dst = reinterpret_cast<uint32_t *>(s_eb + q * kEbRow);
			for (int i = lane; i < neb / 4; i += 64)
				dst[i] = src[i];
		}
	}
	WSYNC();
	for (int it = lane; it < 4 * kSteps12; it += 64) {
		const int q = it / kSteps12, k = it % kSteps12;
		uint2 v = make_uint2(0u, 0u);
		if (g0 + q < a.n) {
			const uint32_t st = c_steps.w[chain][k];
			int va = s_eb[q * kEbRow + (st & 0x3ffu)], vb = s_eb[q * kEbRow + ((st >> 16) & 0x3ffu)];
			if (st & 0x400u) va = (int8_t)(-va);                  // gmr1_scramble_sbit: -128 stays -128
			if (st & 0x4000000u) vb = (int8_t)(-vb);
			const uint32_t a0 = va < 0 ? (uint32_t)(-va) : 0u, a1 = va > 0 ? (uint32_t)va : 0u;
			const uint32_t b0 = vb < 0 ? (uint32_t)(-vb) : 0u, b1 = vb > 0 ? (uint32_t)vb : 0u;
			v = make_uint2((a0 + b0) | ((a0 + b1) << 16), (a1 + b0) | ((a1 + b1) << 16));
		}
		s_bmw[it] = v;
	}
	WSYNC();

	const uint32_t dc = c_dec.v[loc];
	uint32_t oo[4], op[4], T[16];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		oo[ph] = (dc >> (2 * ph)) & 3u;
		op[ph] = (dc >> (8 + 2 * ph)) & 3u;
	}
#pragma unroll
	for (int j = 0; j < 16; j++)
		T[j] = (dc >> 16) & (1u << j);
	const uint16_t *c = reinterpret_cast<const uint16_t *>(s_bmw + row * kSteps12);
	uint16_t *dump = reinterpret_cast<uint16_t *>(s_surv) + lane;
	uint32_t w = (loc ? kAccLeadK5r2 << 16 : 0u) | T[0];
	w = k5w_step<0>(w, c + 0, oo[0], op[0]) + T[1];
	w = k5w_step<1>(w, c + 4, oo[1], op[1]) + T[2];
	w = k5w_step<2>(w, c + 8, oo[2], op[2]) + T[3];
	w = k5w_step<3>(w, c + 12, oo[3], op[3]);
	w = (w & 0xffff0000u) | T[0];
#pragma unroll 1
	for (int m = 0; m < 13; m++) {
		const uint16_t *cm = c + 4 * (4 + 16 * m);
#pragma unroll
		for (int j = 0; j < 16; j += 4) {
			w = k5w_step<0>(w, cm + 4 * (j + 0), oo[0], op[0]) + T[(j + 1) & 15];
			w = k5w_step<1>(w, cm + 4 * (j + 1), oo[1], op[1]) + T[(j + 2) & 15];
			w = k5w_step<2>(w, cm + 4 * (j + 2), oo[2], op[2]) + T[(j + 3) & 15];
			w = k5w_step<3>(w, cm + 4 * (j + 3), oo[3], op[3]) + (j + 4 < 16 ? T[(j + 4) & 15] : 0u);
		}
		dump[m * 64] = (uint16_t)w;
		w = (w & 0xffff0000u) | T[0];
	}
	uint32_t syn;
	k5_12_survivors_crc(s_surv, s_ub, lane, syn);
	const int g = g0 + row;
	if (loc == 0 && g < a.n) {
		store_l2(a.l2 + (size_t)g * 24, s_ub + row * 8);
		a.crc[g] = syn ? 1 : 0;
		a.conv[g] = 0;
	}
}

__global__ __launch_bounds__(64) void k_l1(L1Args a)
{
	__shared__ __align__(16) int8_t s_eb[4 * kEbRow];
	__shared__ __align__(16) uint32_t s_bm[4 * kSteps12];
	__shared__ __align__(16) uint64_t s_surv[kSteps12];
	__shared__ __align__(16) uint32_t s_ub[4 * 8];
	const int lane = threadIdx.x;
	const int g0 = blockIdx.x * 4;
	const int neb = a.chain == kChainCcch ? 432 : 424;
	const int chain = a.chain == kChainCcch ? 1 : 0;

	// soft bits HBM -> LDS, 4 bytes per lane
	for (int q = 0; q < 4; q++) {
		const int g = g0 + q;
		if (g < a.n) {
			const uint32_t *src = reinterpret_cast<const uint32_t *>(a.ebits + (size_t)g * neb);
			uint32_t *dst = reinterpret_cast<uint32_t *>(s_eb + q * kEbRow);
			for (int i = lane; i < neb / 4; i += 64)
				dst[i] = src[i];
		}
	}
	WSYNC();
	for (int q = 0; q < 4; q++) {
		if (g0 + q < a.n) {
			branch_metrics_k5_12(s_eb + q * kEbRow, chain, s_bm + q * kSteps12, lane);
		} else {
			for (int k = lane; k < kSteps12; k += 64)
				s_bm[q * kSteps12 + k] = 0;
		}
	}
	WSYNC();
	uint32_t syn, fae;
	decode4_k5_12(s_bm, s_surv, s_ub, lane, syn, fae);
	const int row = lane >> 4;
	const int g = g0 + row;
	if ((lane & 15) == 0 && g < a.n) {
		store_l2(a.l2 + (size_t)g * 24, s_ub + row * 8);
		a.crc[g] = syn ? 1 : 0;
		a.conv[g] = (int32_t)fae;
	}
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
template <int NPL, int SPS>
static hipError_t launch_rx_t(const RxArgs &a, bool decode, int max_in_len, int max_len, hipStream_t stream)
{
	size_t off[3];
	const size_t lds = lds_layout(max_in_len, max_len, decode, off);
	if (decode) {
		const bool sliced = a.seg_first && a.seg_groups > 0 && a.seg_stride > 0;       // the receive loop's lists, one time slice
		const int grid = sliced ? (a.n / a.seg_stride) * a.seg_groups : (a.n + 3) / 4;
		// (windows beyond 2048 samples -- more than 8 samples per symbol -- exist only in the one-burst-at-a-time body)
		if (a.impl == 1 || NPL > 32) {
			if (a.conv_acc)
				hipLaunchKernelGGL((k_rx<NPL, SPS, true, true>), dim3(grid), dim3(64), lds, stream, a, max_in_len, max_len);
			else
				hipLaunchKernelGGL((k_rx<NPL, SPS, true>), dim3(grid), dim3(64), lds, stream, a, max_in_len, max_len);
		} else if constexpr (NPL <= 32) {
			const int cw = (max_len + 15) & ~15;
			size_t off4[4];
			const size_t lds4 = lds4_layout(a.stage_samples, cw, off4, false, true);
			static size_t pad = (size_t)-1;     // profiling only: extra LDS per wave to cap the occupancy
			if (pad == (size_t)-1) {
				const char *e = profile_env("GMR1_HIP_LDS_PAD");
				pad = e ? (size_t)atoi(e) : 0;
			}
			// small batches (the receive loop's rounds): one burst per wave, four times the waves
			static int bpw_force = -1;          // profiling only: GMR1_HIP_RX_BPW = 1 | 4
			if (bpw_force < 0) {
				const char *e = profile_env("GMR1_HIP_RX_BPW");
				bpw_force = e ? atoi(e) : 0;
			}
			const int bpw = sliced ? 4 : (bpw_force == 1 || bpw_force == 4 ? bpw_force : (a.n <= 4096 ? 1 : 4));
			const int grid4 = sliced ? (a.n / a.seg_stride) * a.seg_groups : (a.n + bpw - 1) / bpw;
			// (the instantiation with the burst energy and the segment bound is the receive loop's: see rx4_body's EN)
			const bool en = a.energy != nullptr || a.seg_count != nullptr;
			if (a.plane_stride) {
				// polyphase-planar sample array: the fused batch kernel at 4 samples per symbol (the host refuses anything else)
				if constexpr (NPL == 16 && SPS == 4) {
					if (en)
						return hipErrorInvalidValue;
					if (a.conv_acc)
						hipLaunchKernelGGL((k_rx4<16, 4, true, false, true>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
					else
						hipLaunchKernelGGL((k_rx4<16, 4, false, false, true>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
				} else
					return hipErrorInvalidValue;
			} else
			if (a.conv_acc) {
				if (en)
					hipLaunchKernelGGL((k_rx4<NPL, SPS, true, true>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
				else
					hipLaunchKernelGGL((k_rx4<NPL, SPS, true, false>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
			} else {
				if (en)
					hipLaunchKernelGGL((k_rx4<NPL, SPS, false, true>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
				else
					hipLaunchKernelGGL((k_rx4<NPL, SPS, false, false>), dim3(grid4), dim3(64), lds4 + pad, stream, a, a.stage_samples, cw, bpw);
			}
		}
	} else {
		hipLaunchKernelGGL((k_rx<NPL, SPS, false>), dim3(a.n), dim3(64), lds, stream, a, max_in_len, max_len);
	}
	return hipGetLastError();
}

hipError_t launch_rx(const RxArgs &a, bool decode, int max_in_len, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (max_in_len > kMaxInLen)
		return hipErrorInvalidValue;
	// symbols per burst: fused path is BCCH/DC6 (234); generic path sizes for the longest format
	// third kernel argument: lags the correlation accumulator must hold (fused path: 20*sps + 1)
	// (demodulation only: the lags of this launch's one format, in_len - symbols * sps + 1; ssyms_stride = symbols per burst)
	const int w_demod = a.in_len[0] - a.ssyms_stride * a.sps + 1;
	const int max_len = decode ? (20 * a.sps + 1) : (w_demod > kMaxWindow ? w_demod : kMaxWindow);
	if (!decode && (a.impl == 2 || a.impl == 3 || a.impl == 4) && a.sps == 4 && max_in_len <= 1024) {
		// large batch of one simple burst format (the host checked what rx4_body<GEN> assumes; impl 3: also what its
		// small variant assumes; impl 4: the small variant for two training sequences and BPSK): four bursts per wave
		const int cw1 = (a.in_len[0] - a.ssyms_stride * 4 + 1 + 15) & ~15;      // lags (ssyms_stride = symbols per burst)
		const int cw = a.impl == 4 ? 2 * cw1 : cw1;                            // impl 4: two correlation arrays per burst
		size_t off4[4];
		const size_t lds4 = lds4_layout(a.stage_samples, cw, off4, true);
		const int grid4 = (a.n + 3) / 4;
		if (a.impl == 4)
			hipLaunchKernelGGL((k_rx4g<8, 4, true>), dim3(grid4), dim3(64), lds4, stream, a, a.stage_samples, cw);
		else if (a.impl == 3)
			hipLaunchKernelGGL((k_rx4g<8, 4>), dim3(grid4), dim3(64), lds4, stream, a, a.stage_samples, cw);
		else
			hipLaunchKernelGGL((k_rx4g<16, 4>), dim3(grid4), dim3(64), lds4, stream, a, a.stage_samples, cw);
		return hipGetLastError();
	}
	if (max_in_len <= 1024) {
		if (a.sps == 4)
			return launch_rx_t<16, 4>(a, decode, max_in_len, max_len, stream);
		return launch_rx_t<16, 0>(a, decode, max_in_len, max_len, stream);
	}
	if (max_in_len <= 2048) {
		if (a.sps == 4)
			return launch_rx_t<32, 4>(a, decode, max_in_len, max_len, stream);
		return launch_rx_t<32, 0>(a, decode, max_in_len, max_len, stream);
	}
	return launch_rx_t<64, 0>(a, decode, max_in_len, max_len, stream);
}

// Interleaved sample array -> polyphase-planar (what gmr1_hip_rx_bcch_ccch_batch_planar_dev reads): a work-group takes 256 sps
// consecutive samples; thread t of it reads samples t, t + 256, ... (coalesced) and, through LDS, writes place t of each of
// the sps planes (coalesced again).  HBM-bound by construction: every sample read once, written once.
constexpr int kPlanarTile = 256;
__global__ __launch_bounds__(256) void k_to_planar(const float2 *__restrict__ in, float2 *__restrict__ out, unsigned long long n,
                                                   int sps, long long plane_stride)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float2 *t = reinterpret_cast<float2 *>(lds_raw);
	const unsigned long long p0 = (unsigned long long)blockIdx.x * kPlanarTile;     // first place of the tile in every plane
	const unsigned long long s0 = p0 * (unsigned long long)sps;
	for (int k = 0; k < sps; k++) {
		const unsigned long long s = s0 + (unsigned long long)(k * kPlanarTile + (int)threadIdx.x);
		t[k * kPlanarTile + threadIdx.x] = s < n ? in[s] : make_float2(0.f, 0.f);
	}
	__syncthreads();
	for (int ph = 0; ph < sps; ph++) {
		const unsigned long long s = s0 + (unsigned long long)((int)threadIdx.x * sps + ph);
		if (s < n)
			out[(long long)ph * plane_stride + (long long)(p0 + threadIdx.x)] = t[(int)threadIdx.x * sps + ph];
	}
}

hipError_t launch_to_planar(const float2 *in, float2 *out, unsigned long long n, int sps, long long plane_stride, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	const unsigned long long places = (n + (unsigned long long)sps - 1) / (unsigned long long)sps;
	const unsigned long long grid = (places + kPlanarTile - 1) / kPlanarTile;
	if (grid > 0x7fffffffull)
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_to_planar, dim3((unsigned)grid), dim3(256), (size_t)sps * kPlanarTile * 8, stream, in, out, n, sps, plane_stride);
	return hipGetLastError();
}

#ifdef GMR1_HIP_PROFILE
extern "C" int gmr1_hip_prof_stamps(unsigned long long *out16)
{
	return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) == hipSuccess ? 0 : -5;
}
// reads (and clears) the count of mis-speculated picks
extern "C" int gmr1_hip_prof_miss(void)
{
	int v = 0, z = 0;
	if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_prof_miss), sizeof(v)) != hipSuccess)
		return -5;
	hipMemcpyToSymbol(HIP_SYMBOL(g_prof_miss), &z, sizeof(z));
	return v;
}

extern "C" int gmr1_hip_prof_flag(int v)
{
	return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_flag), &v, sizeof(v)) == hipSuccess ? 0 : -5;
}
#endif

hipError_t launch_rx_tch3(const RxArgs &a, const Tch3Args &t, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.impl != 3 || a.sps != 4 || a.in_len[0] > 512 || a.ebits_stride != 212 || a.n != t.n)
		return hipErrorInvalidValue;
	const int cw = (a.in_len[0] - a.ssyms_stride * 4 + 1 + 15) & ~15;          // lags (ssyms_stride = symbols per burst)
	size_t off4[4];
	// the demodulator's phases as rx4_body<8, 4, GEN, ..., EBROW = 216> carves them (pass 1: staged windows, correlation,
	// coefficients; pass 2: four 216-byte soft-bit rows and the soft-bit table) -- the same function sizes them here --
	// and, after them, the rows + the decoder's tables
	size_t lds = lds4_layout(a.stage_samples, cw, off4, true, false, 216);
	const size_t need = 4 * 216 + sizeof(t3::Tch3Lds);
	if (lds < need)
		lds = need;
	{
		static size_t pad = (size_t)-1;     // profiling only: extra LDS per wave to cap the occupancy
		if (pad == (size_t)-1) {
			const char *e = profile_env("GMR1_HIP_LDS_PAD");
			pad = e ? (size_t)atoi(e) : 0;
		}
		lds += pad;
	}
	const int grid4 = (a.n + 3) / 4;
	if (t.conv_acc)
		hipLaunchKernelGGL(k_rx4g_tch3<true>, dim3(grid4), dim3(64), lds, stream, a, t, a.stage_samples, cw);
	else
		hipLaunchKernelGGL(k_rx4g_tch3<false>, dim3(grid4), dim3(64), lds, stream, a, t, a.stage_samples, cw);
	return hipGetLastError();
}

#include "rx_loop_kernels.inc"

hipError_t launch_detect(const DetectArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.in_len > kMaxInLen)
		return hipErrorInvalidValue;
	size_t off[3];
	const size_t lds = lds_layout(a.in_len, a.max_lags, false, off);
	if (a.in_len <= 1024) {
		if (a.sps == 4)
			hipLaunchKernelGGL((k_detect<16, 4>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
		else
			hipLaunchKernelGGL((k_detect<16, 0>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	} else if (a.in_len <= 2048) {
		if (a.sps == 4)
			hipLaunchKernelGGL((k_detect<32, 4>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
		else
			hipLaunchKernelGGL((k_detect<32, 0>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	} else {
		hipLaunchKernelGGL((k_detect<64, 0>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	}
	return hipGetLastError();
}

hipError_t launch_mod_order(const ModOrderArgs &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.in_len > kMaxInLen)
		return hipErrorInvalidValue;
	size_t off[3];
	const size_t lds = lds_layout(a.in_len, 0, false, off);
	if (a.in_len <= 1024)
		hipLaunchKernelGGL((k_mod_order<16>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	else if (a.in_len <= 2048)
		hipLaunchKernelGGL((k_mod_order<32>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	else
		hipLaunchKernelGGL((k_mod_order<64>), dim3(a.n), dim3(64), lds, stream, a, a.in_len);
	return hipGetLastError();
}

hipError_t launch_l1(const L1Args &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.conv_acc)
		hipLaunchKernelGGL(k_l1_acc, dim3((a.n + 3) / 4), dim3(64), 0, stream, a);
	else
		hipLaunchKernelGGL(k_l1, dim3((a.n + 3) / 4), dim3(64), 0, stream, a);
	return hipGetLastError();
}

}  // namespace gmr1
