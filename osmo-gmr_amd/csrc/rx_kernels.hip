// rx_kernels.hip -- normal-burst receive kernels for gfx950 (MI355X).
//
//   k_rx<DECODE=false> : pi/4-CxPSK burst demodulation, one burst per wavefront
//                        (reference src/sdr/pi4cxpsk.c:520-602 gmr1_pi4cxpsk_demod)
//   k_rx<DECODE=true>  : the same, four bursts per wavefront back to back, followed
//                        by the BCCH/CCCH layer-1 chain for the four bursts at once:
//                        descramble + de-interleave folded into the branch-metric
//                        gather, 16-state K=5 rate-1/2 Viterbi with one burst per
//                        16-lane DPP row, traceback, CRC16, LSB-first packing
//                        (reference src/l1/bcch.c:83-103, src/l1/ccch.c:87-107 and
//                        libosmocore's generic osmo_conv_decode).
//   k_l1               : the layer-1 chain alone on soft bits read from HBM.
//
// Design notes (DESIGN.md has the long form):
//   * Work-groups are single 64-lane wavefronts: every hand-off goes through the
//     wave's own LDS slice and needs no s_barrier.
//   * Samples are loaded once from HBM with coalesced 8-byte-per-lane loads, DC /
//     power normalised in registers and parked in LDS; everything else reads LDS.
//   * The sync search never derotates the window: |sum conj(ref_n) x[.] e^{j th n}|
//     is evaluated with per-burst rotated coefficients, which drops ~1000
//     sincos per burst.  Only the 234 decimated symbols are derotated.
//   * Viterbi state s lives in lane rotr^k(s) of its row at trellis step k, so the
//     add-compare-select butterfly is in place: the partner metric is one DPP
//     lane-xor away and no metric ever moves.  Decisions are collected in VGPRs: lane (k mod 64)
//     keeps the 64-bit ballot of step k (4 bursts x 16 states).
#include "gmr1_dev.h"

namespace gmr1 {

#define WSYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
		__builtin_amdgcn_wave_barrier();                          \
	} while (0)

static constexpr float kPif = 3.14159265358979323846f;
static constexpr uint32_t kMaxAe = 0x00ffffffu;   // libosmocore MAX_AE
static constexpr int kSteps12 = 212;              // 208 data + 4 flush steps (BCCH/CCCH)
static constexpr int kEbitsLds = 704;             // >= 662 (NT9), multiple of 16

// ---------------------------------------------------------------------------
// compile-time tables
// ---------------------------------------------------------------------------
struct ScrTable { uint32_t w[24]; };
static constexpr ScrTable make_scr()
{
	// GMR-1 scrambler (reference src/l1/scramb.c:39-52): 15-bit LFSR, seed 0x4d4b
	ScrTable t{};
	uint16_t r = 0x4d4b;
	for (int i = 0; i < 24 * 32; i++) {
		uint32_t b = ((r >> 14) ^ r) & 1u;
		r = (uint16_t)((r << 1) | b);
		t.w[i >> 5] |= b << (i & 31);
	}
	return t;
}
__constant__ ScrTable c_scr = make_scr();

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

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o);
	return v;
}

__device__ __forceinline__ float half_sum(float v)   // within each 32-lane half
{
#pragma unroll
	for (int o = 16; o > 0; o >>= 1)
		v += __shfl_xor(v, o);
	return v;
}

__device__ __forceinline__ float cabs_d(float re, float im)
{
	// glibc hypotf evaluates in double; do the same so |.| agrees to the last bit
	return (float)sqrt((double)re * (double)re + (double)im * (double)im);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// conj(ref) * v for ref = modulating value of sync symbol `sym` (exact: ref is +-1 / +-j)
__device__ __forceinline__ float2 conj_ref_mul(int nbits, int sym, float2 v)
{
	if (nbits == 2) {
		switch (sym & 3) {
		case 0: return v;
		case 1: return make_float2(v.y, -v.x);
		case 2: return make_float2(-v.x, -v.y);
		default: return make_float2(-v.y, v.x);
		}
	}
	return (sym & 1) ? make_float2(-v.x, -v.y) : v;
}

__device__ __forceinline__ uint32_t rotl4(uint32_t x, int r) { return ((x << r) | (x >> (4 - r))) & 15u; }
__device__ __forceinline__ uint32_t rotr4(uint32_t x, int r) { return ((x >> r) | (x << (4 - r))) & 15u; }

// K=5 rate-1/2 code (g0 = 1+D^3+D^4, g1 = 1+D+D^2+D^4; reference src/l1/conv.c:123-145)
__device__ __forceinline__ uint32_t out_k5_12(uint32_t s, uint32_t b)
{
	uint32_t reg = (s << 1) | b;
	return ((uint32_t)(__popc(reg & 0x19u) & 1) << 1) | (uint32_t)(__popc(reg & 0x17u) & 1);
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// value of lane (l ^ X) within a 16-lane row, X in {8,4,2,1}
template <int X>
__device__ __forceinline__ uint32_t row_xor(uint32_t v)
{
	if constexpr (X == 8) return dpp<0x128>(v);                    // row_ror:8
	else if constexpr (X == 4) return dpp<0x1B>(dpp<0x141>(v));    // half_mirror, then quad [3,2,1,0]
	else if constexpr (X == 2) return dpp<0x4E>(v);                // quad_perm [2,3,0,1]
	else return dpp<0xB1>(v);                                      // quad_perm [1,0,3,2]
}

// ---------------------------------------------------------------------------
// LDS carve-up of one wavefront
// ---------------------------------------------------------------------------
struct Lds {
	float2 *x;        // normalised input window           [max_in_len]
	float *corr;      // accumulated sync correlation      [kMaxWindow]
	float2 *coef;     // rotated sync reference            [kMaxCoef]
	float2 *y;        // decimated symbols                 [max_len]
	int8_t *eb;       // soft bits of the current burst    [kEbitsLds]
	uint32_t *bm;     // branch metrics, 4 rows x 212      (DECODE only)
	uint64_t *surv;   // survivor ballots, aliases x       [212]
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline size_t lds_layout(int max_in_len, int max_len, bool decode, size_t *off)
{
	size_t o = 0;
	off[0] = o; o += align16((size_t)max_in_len * 8);
	off[1] = o; o += kMaxWindow * 4;
	off[2] = o; o += kMaxCoef * 8;
	off[3] = o; o += align16((size_t)max_len * 8);
	off[4] = o; o += kEbitsLds;
	off[5] = o; if (decode) o += 4 * kSteps12 * 4;
	return align16(o);
}

// ---------------------------------------------------------------------------
// demodulation of one burst by one wavefront
// returns the reference's rv (0, or -1 when no sync sequence has power)
// ---------------------------------------------------------------------------
template <int NPL>
__device__ int demod_one(const DevBurst *__restrict__ bt, const float2 *__restrict__ in, int in_len,
                         int sps, float freq_shift, const Lds &L, int lane,
                         int &sync_id_o, float &toa_o, float &ferr_o, float *__restrict__ g_ssyms)
{
	const int nbits = bt->nbits;
	const int blen = bt->len;
	const int w = in_len - blen * sps + 1;

	// ---- load + normalise (osmo_cxvec_sig_normalize, decim 1) ------------------
	float2 v[NPL];
	float sr = 0.f, si = 0.f;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		int idx = lane + 64 * k;
		v[k] = (idx < in_len) ? in[idx] : make_float2(0.f, 0.f);
		sr += v[k].x;
		si += v[k].y;
	}
	sr = wave_sum(sr);
	si = wave_sum(si);
	const float avr = sr / (float)in_len, avi = si / (float)in_len;
	float acc = 0.f;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		int idx = lane + 64 * k;
		v[k].x -= avr;
		v[k].y -= avi;
		if (idx < in_len)
			acc += v[k].x * v[k].x + v[k].y * v[k].y;
	}
	float sigma = wave_sum(acc) / (float)in_len;
	float stddev = sqrtf(sigma);
	if (stddev == 0.0f)
		stddev = 1.0f;
	const float inv = 1.0f / stddev;
#pragma unroll
	for (int k = 0; k < NPL; k++) {
		int idx = lane + 64 * k;
		if (idx < in_len)
			L.x[idx] = make_float2(v[k].x * inv, v[k].y * inv);
	}
	for (int j = lane; j < w; j += 64)
		L.corr[j] = 0.f;

	// per-sample derotation step (pi4cxpsk.c:539)
	const float fs = (freq_shift - bt->rotation) / (float)sps;

	// ---- sync search (pi4cxpsk.c:184-268) --------------------------------------
	float p_toa = 0.f, p_pwr = 0.f;
	int p_idx = -1;
	const int win = w < 3 ? w : 3;

	for (int sq = 0; sq < bt->n_sync; sq++) {
		const int tl = bt->sync_tl[sq];
		const int nch = bt->n_chunks[sq];

		// rotated reference: conj(ref_n) * e^{j fs sps n}; the common phase of a lag
		// drops out under |.|, so the window itself is never derotated here
		WSYNC();
		for (int n = lane; n < tl; n += 64) {
			int ch = 0, base = 0;
			while (n >= base + bt->sync[sq][ch].len) {
				base += bt->sync[sq][ch].len;
				ch++;
			}
			const int nn = n - base;
			const int sym = bt->sync[sq][ch].syms[nn];
			float s, c;
			sincosf(fs * (float)(nn * sps), &s, &c);
			L.coef[n] = conj_ref_mul(nbits, sym, make_float2(c, s));
		}
		WSYNC();

		for (int j = lane; j < w; j += 64) {
			float cj = L.corr[j];
			int base = 0;
			for (int ch = 0; ch < nch; ch++) {
				const int pos = bt->sync[sq][ch].pos, len = bt->sync[sq][ch].len;
				const float2 *xp = L.x + pos * sps + j;
				float ar = 0.f, ai = 0.f;
				for (int n = 0; n < len; n++) {
					const float2 x = xp[n * sps];
					const float2 cf = L.coef[base + n];
					ar = fmaf(cf.x, x.x, fmaf(-cf.y, x.y, ar));
					ai = fmaf(cf.x, x.y, fmaf(cf.y, x.x, ai));
				}
				base += len;
				cj += cabs_d(ar, ai);
			}
			L.corr[j] = cj;
		}
		WSYNC();

		// osmo_cxvec_peak_energy_find(corr, 3, PEAK_EARLY_LATE, &peak)
		float bv = -1.f;
		int bi = 0x7fffffff;
		for (int m = lane; m + win <= w; m += 64) {
			float e = 0.f;
			for (int k = 0; k < win; k++) {
				float c = L.corr[m + k];
				e += c * c;
			}
			if (e > bv) { bv = e; bi = m; }
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			float ov = __shfl_xor(bv, o);
			int oi = __shfl_xor(bi, o);
			if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
		}
		const int mi = (bi == 0x7fffffff) ? 0 : bi;
		int p = mi;
		{
			float pe = -1.f;
			for (int k = 0; k < win; k++) {
				float c = L.corr[mi + k];
				float e = c * c;
				if (e > pe) { pe = e; p = mi + k; }
			}
		}

		// sinc-interpolated value of corr at `pos`: lanes t<21 of each 32-lane half
		auto interp_term = [&](float pos, int t) -> float {
			const int i0 = (int)floorf(pos);
			int b = i0 - 10, e = i0 + 11;
			if (b < 0) b = 0;
			if (e >= w) e = w - 1;
			const int i = i0 - 10 + t;
			if (t >= 21 || i < b || i >= e)
				return 0.f;
			const float xx = kPif * ((float)i - pos);
			const float sc = (xx >= 0.01f || xx <= -0.01f) ? (sinf(xx) / xx) : 1.0f;
			return L.corr[i] * sc;
		};

		float early = (float)p - 1.0f, late = (float)p + 1.0f, incr = 0.5f;
		while (incr > (1.0f / 1024.0f)) {
			float t = interp_term(lane < 32 ? early : late, lane & 31);
			t = half_sum(t);
			const float ev = __shfl(t, 0), lv = __shfl(t, 32);
			const float ee = ev * ev, le = lv * lv;
			if (ee > le)      { early -= incr; late -= incr; }
			else if (ee < le) { early += incr; late += incr; }
			else break;
			incr *= 0.5f;
		}
		const float s_toa = early + 1.0f;
		float pk = interp_term(s_toa, lane & 31);
		pk = __shfl(half_sum(pk), 0);
		pk = pk / (float)tl;
		const float s_pwr = pk * pk;
		if (s_pwr > p_pwr) {
			p_pwr = s_pwr;
			p_toa = s_toa;
			p_idx = sq;
		}
	}

	sync_id_o = p_idx;
	toa_o = p_toa;
	if (p_idx < 0) {
		ferr_o = 0.f;
		return -1;
	}
	const int sq = p_idx;
	const int nch = bt->n_chunks[sq];

	// ---- align + decimate (pi4cxpsk.c:286-297), derotating only what is kept ----
	const int d = (int)roundf(p_toa);
	for (int i = lane; i < blen; i += 64) {
		const int j = i * sps + d;
		float2 x = (j >= 0 && j < in_len) ? L.x[j] : make_float2(0.f, 0.f);
		if (fs != 0.0f) {
			float s, c;
			sincosf(fs * (float)j, &s, &c);
			x = cmul(x, make_float2(c, s));
		}
		L.y[i] = x;
	}
	WSYNC();

	// ---- fine frequency error from the sync chunks (pi4cxpsk.c:360-406) ---------
	float ffe = 0.f;
	if (nch > 1) {
		float cr = 0.f, ci = 0.f;
		if (lane < nch) {
			const int pos = bt->sync[sq][lane].pos, len = bt->sync[sq][lane].len;
			for (int j = 0; j < len; j++) {
				float2 t = conj_ref_mul(nbits, bt->sync[sq][lane].syms[j], L.y[pos + j]);
				cr += t.x;
				ci += t.y;
			}
		}
		float f = 0.f;
		float ppos = (float)bt->sync[sq][0].pos + (float)bt->sync[sq][0].len / 2.0f;
		float pr = __shfl(cr, 0), pi = __shfl(ci, 0);
		for (int i = 1; i < nch; i++) {
			const float cpos = (float)bt->sync[sq][i].pos + (float)bt->sync[sq][i].len / 2.0f;
			const float r = __shfl(cr, i), q = __shfl(ci, i);
			// corr[i] * conj(corr[i-1])
			const float re = r * pr - q * (-pi);
			const float im = r * (-pi) + q * pr;
			f += atan2f(im, re) / (cpos - ppos);
			ppos = cpos; pr = r; pi = q;
		}
		f /= (float)(nch - 1);
		ffe = f;
	}
	ferr_o = ffe;

	// ---- rotate by -ffe (pi4cxpsk.c:574-575), in registers -------------------------
	if (ffe != 0.0f) {
		const float rps = -ffe;
		for (int i = lane; i < blen; i += 64) {
			float s, c;
			sincosf(rps * (float)i, &s, &c);
			L.y[i] = cmul(L.y[i], make_float2(c, s));
		}
		WSYNC();
	}

	// ---- carrier phase from the sync symbols (pi4cxpsk.c:415-433) ------------------
	float phr = 0.f, phi = 0.f;
	for (int ch = 0; ch < nch; ch++) {
		const int pos = bt->sync[sq][ch].pos, len = bt->sync[sq][ch].len;
		for (int j = 0; j < len; j++) {
			float2 t = conj_ref_mul(nbits, bt->sync[sq][ch].syms[j], L.y[pos + j]);
			phr += t.x;
			phi += t.y;
		}
	}
	const float pm = cabs_d(phr, phi);
	const float2 cph = make_float2(phr / pm, -(phi / pm));   // conj(phasor)

	// ---- soft symbols + soft bits (pi4cxpsk.c:442-503) ------------------------------
	const float dd = (2.0f * kPif) / (float)(1 << nbits);
	const int mask = (1 << nbits) - 1;
	const int nd = bt->n_data;
	for (int i = lane; i < blen; i += 64) {
		const float2 yy = cmul(L.y[i], cph);
		const float sv = atan2f(yy.y, yy.x) / dd;
		if (g_ssyms)
			g_ssyms[i] = sv;
		int ord = -1;
		for (int c = 0; c < nd; c++) {
			const int dp = bt->dpos[c], dl = bt->dlen[c];
			if (i >= dp && i < dp + dl)
				ord = bt->dcum[c] + (i - dp);
		}
		if (ord >= 0) {
			const float svr = roundf(sv);
			const int sp = (int)svr & mask;
			const int ss = (svr > sv ? (sp - 1) : (sp + 1)) & mask;
			const int dq = (int)roundf((2.0f * fabsf(svr - sv)) * 64.0f);
			if (nbits == 2) {
				// symbol -> bits 0:00 1:01 2:11 3:10 (pi4cxpsk.c:95-100)
				const int p0 = sp >> 1, p1 = (sp ^ (sp >> 1)) & 1;
				const int s0 = ss >> 1, s1 = (ss ^ (ss >> 1)) & 1;
				const int v0 = 127 - ((p0 ^ s0) ? dq : (dq >> 1));
				const int v1 = 127 - ((p1 ^ s1) ? dq : (dq >> 1));
				L.eb[2 * ord]     = (int8_t)(p0 ? -v0 : v0);
				L.eb[2 * ord + 1] = (int8_t)(p1 ? -v1 : v1);
			} else {
				const int p0 = sp & 1, s0 = ss & 1;
				const int v0 = 127 - ((p0 ^ s0) ? dq : (dq >> 1));
				L.eb[ord] = (int8_t)(p0 ? -v0 : v0);
			}
		}
	}
	WSYNC();
	return 0;
}

// ---------------------------------------------------------------------------
// branch metrics of one burst into bm[0..212): byte ov = cost of coded word ov
// (descramble + de-interleave folded into the gather)
//   bcch.c:91-92 / ccch.c:95-96, interleave.c:73-87, scramb.c:63-73
// ---------------------------------------------------------------------------
__device__ __forceinline__ void branch_metrics_k5_12(const int8_t *__restrict__ eb, int off,
                                                     uint32_t *__restrict__ bm, int lane)
{
	for (int k = lane; k < kSteps12; k += 64) {
		int c0[2], c1[2];
#pragma unroll
		for (int j = 0; j < 2; j++) {
			const int kc = 2 * k + j;
			const int ei = 53 * ((5 * kc) & 7) + (kc >> 3) + off;
			int v = eb[ei];
			if ((c_scr.w[ei >> 5] >> (ei & 31)) & 1u)
				v = (int8_t)(-v);
			const int e0 = v - 127, e1 = v + 127;
			c0[j] = v ? ((e0 * e0) >> 9) : 0;
			c1[j] = v ? ((e1 * e1) >> 9) : 0;
		}
		bm[k] = (uint32_t)(c0[0] + c0[1]) | ((uint32_t)(c0[0] + c1[1]) << 8) |
		        ((uint32_t)(c1[0] + c0[1]) << 16) | ((uint32_t)(c1[0] + c1[1]) << 24);
	}
}

// ---------------------------------------------------------------------------
// 4 x (K=5, rate 1/2, 208 bits + flush) Viterbi, one burst per 16-lane row
// ---------------------------------------------------------------------------
template <int PH>
__device__ __forceinline__ void acs_step(uint32_t &ae, uint32_t bmw, uint32_t sh_own, uint32_t sh_par,
                                         bool b_is_one, bool flush, unsigned long long &ballot)
{
	const uint32_t par = row_xor<(8 >> PH)>(ae);
	const uint32_t n_own = ae + ((bmw >> sh_own) & 0xffu);
	const uint32_t n_par = par + ((bmw >> sh_par) & 0xffu);
	uint32_t nw = n_own < n_par ? n_own : n_par;
	nw = nw < kMaxAe ? nw : kMaxAe;
	// hi predecessor ((t>>1)+8) wins only when strictly better: ties keep the lower state
	const bool dec = b_is_one ? (n_own < n_par) : (n_par < n_own);
	if (flush && b_is_one)
		nw = kMaxAe;            // flush steps only take the b=0 transitions
	ae = nw;
	ballot = __ballot(dec);
}

__device__ void decode4_k5_12(const uint32_t *__restrict__ bm /* 4 x 212 */, uint64_t *__restrict__ surv,
                              int lane, uint32_t words[7], uint32_t &syn_o, uint32_t &final_ae)
{
	const int row = lane >> 4;
	const uint32_t loc = (uint32_t)lane & 15u;
	uint32_t sh_own[4], sh_par[4];
	bool b1[4];
#pragma unroll
	for (int ph = 0; ph < 4; ph++) {
		const uint32_t s = rotl4(loc, ph);
		const uint32_t b = s >> 3;
		b1[ph] = b != 0;
		sh_own[ph] = 8u * out_k5_12(s, b);
		sh_par[ph] = 8u * out_k5_12(s ^ 8u, b);
	}
	uint32_t ae = loc ? kMaxAe : 0u;
	const uint32_t *bmr = bm + row * kSteps12;

	int slo[4] = {0, 0, 0, 0}, shi[4] = {0, 0, 0, 0};
#pragma unroll
	for (int blk = 0; blk < 4; blk++) {
		const int kend = (kSteps12 - blk * 64) < 64 ? (kSteps12 - blk * 64) : 64;
		for (int kk = 0; kk < kend; kk += 4) {
			const int k = blk * 64 + kk;
			const bool fl = k >= 208;
			unsigned long long m;
			acs_step<0>(ae, bmr[k + 0], sh_own[0], sh_par[0], b1[0], fl, m);
			slo[blk] = (lane == kk + 0) ? (int)(uint32_t)m : slo[blk];
			shi[blk] = (lane == kk + 0) ? (int)(uint32_t)(m >> 32) : shi[blk];
			acs_step<1>(ae, bmr[k + 1], sh_own[1], sh_par[1], b1[1], fl, m);
			slo[blk] = (lane == kk + 1) ? (int)(uint32_t)m : slo[blk];
			shi[blk] = (lane == kk + 1) ? (int)(uint32_t)(m >> 32) : shi[blk];
			acs_step<2>(ae, bmr[k + 2], sh_own[2], sh_par[2], b1[2], fl, m);
			slo[blk] = (lane == kk + 2) ? (int)(uint32_t)m : slo[blk];
			shi[blk] = (lane == kk + 2) ? (int)(uint32_t)(m >> 32) : shi[blk];
			acs_step<3>(ae, bmr[k + 3], sh_own[3], sh_par[3], b1[3], fl, m);
			slo[blk] = (lane == kk + 3) ? (int)(uint32_t)m : slo[blk];
			shi[blk] = (lane == kk + 3) ? (int)(uint32_t)(m >> 32) : shi[blk];
		}
	}
	// 212 = 53 * 4 steps: the layout is back to identity, state 0 sits in lane 0 of the row
	final_ae = ae;

	WSYNC();
#pragma unroll
	for (int blk = 0; blk < 4; blk++) {
		const int k = blk * 64 + lane;
		if (k < kSteps12)
			surv[k] = (uint64_t)(uint32_t)slo[blk] | ((uint64_t)(uint32_t)shi[blk] << 32);
	}
	WSYNC();

	// traceback: one lane per row (osmo_conv_decode_get_output, end state 0 after flush)
	uint32_t syn = 0;
#pragma unroll
	for (int i = 0; i < 7; i++)
		words[i] = 0;
	if (loc == 0) {
		const uint16_t *s16 = reinterpret_cast<const uint16_t *>(surv) + row;
		uint32_t cur = 0;
		for (int k = kSteps12 - 1; k >= 208; k--) {
			const uint32_t l = rotr4(cur, (k + 1) & 3);
			const uint32_t dbit = ((uint32_t)s16[4 * k] >> l) & 1u;
			cur = (cur >> 1) | (dbit << 3);
		}
#pragma unroll
		for (int wi = 6; wi >= 0; wi--) {
			uint32_t wv = 0;
			for (int bit = (wi == 6 ? 15 : 31); bit >= 0; bit--) {
				const int k = wi * 32 + bit;
				const uint32_t l = rotr4(cur, (k + 1) & 3);
				const uint32_t dbit = ((uint32_t)s16[4 * k] >> l) & 1u;
				const uint32_t ob = cur & 1u;
				wv |= ob << bit;
				syn ^= ob ? (uint32_t)c_syn.s[k] : 0u;
				cur = (cur >> 1) | (dbit << 3);
			}
			words[wi] = wv;
		}
	}
	syn_o = syn;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int NPL, bool DECODE>
__global__ __launch_bounds__(64) void k_rx(RxArgs a, int max_in_len, int max_len)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	size_t off[6];
	lds_layout(max_in_len, max_len, DECODE, off);
	Lds L;
	L.x = reinterpret_cast<float2 *>(lds_raw + off[0]);
	L.corr = reinterpret_cast<float *>(lds_raw + off[1]);
	L.coef = reinterpret_cast<float2 *>(lds_raw + off[2]);
	L.y = reinterpret_cast<float2 *>(lds_raw + off[3]);
	L.eb = reinterpret_cast<int8_t *>(lds_raw + off[4]);
	L.bm = reinterpret_cast<uint32_t *>(lds_raw + off[5]);
	L.surv = reinterpret_cast<uint64_t *>(lds_raw + off[0]);

	constexpr int PER = DECODE ? 4 : 1;
	const int g0 = blockIdx.x * PER;
	int row_ok = 0;   // bit q set: burst q of this wave demodulated fine

	for (int q = 0; q < PER; q++) {
		const int g = g0 + q;
		if (g >= a.n) {
			if (DECODE)
				for (int k = lane; k < kSteps12; k += 64)
					L.bm[q * kSteps12 + k] = 0;
			continue;
		}
		int type, in_len, chain_off = 0;
		if (DECODE) {
			const int kind = a.kind[g] ? 1 : 0;
			type = kind ? GMR1_HIP_DC6 : GMR1_HIP_BCCH;
			in_len = a.in_len[kind];
			chain_off = kind ? 4 : 0;
		} else {
			type = a.fixed_type;
			in_len = a.in_len[0];
		}
		type = __builtin_amdgcn_readfirstlane(type);
		in_len = __builtin_amdgcn_readfirstlane(in_len);
		const DevBurst *bt = a.types + type;
		const float fsh = a.freq_shift ? a.freq_shift[g] : 0.0f;
		int sid = -1;
		float toa = 0.f, fe = 0.f;
		float *gss = a.ssyms ? a.ssyms + (size_t)g * a.ssyms_stride : nullptr;

		WSYNC();
		const int rv = demod_one<NPL>(bt, a.iq + a.offset[g], in_len, a.sps, fsh, L, lane, sid, toa, fe, gss);

		if (lane == 0) {
			a.rv[g] = rv;
			if (a.sync_id) a.sync_id[g] = sid;
			if (a.toa) a.toa[g] = rv ? 0.f : toa;
			if (a.freq_err) a.freq_err[g] = rv ? 0.f : fe;
		}
		const int neb = bt->ebits;
		if (a.ebits) {
			int8_t *ge = a.ebits + (size_t)g * a.ebits_stride;
			for (int i = lane; i < a.ebits_stride; i += 64)
				ge[i] = (rv == 0 && i < neb) ? L.eb[i] : (int8_t)0;
		}
		if (rv && gss)
			for (int i = lane; i < bt->len; i += 64)
				gss[i] = 0.f;
		if (DECODE) {
			if (rv == 0) {
				row_ok |= 1 << q;
				branch_metrics_k5_12(L.eb, chain_off, L.bm + q * kSteps12, lane);
			} else {
				for (int k = lane; k < kSteps12; k += 64)
					L.bm[q * kSteps12 + k] = 0;
			}
		}
	}

	if (DECODE) {
		WSYNC();
		uint32_t words[7], syn, fae;
		decode4_k5_12(L.bm, L.surv, lane, words, syn, fae);
		const int row = lane >> 4;
		const int g = g0 + row;
		if ((lane & 15) == 0 && g < a.n) {
			uint32_t *l2w = reinterpret_cast<uint32_t *>(a.l2 + (size_t)g * 24);
			if ((row_ok >> row) & 1) {
#pragma unroll
				for (int i = 0; i < 6; i++)
					l2w[i] = words[i];
				a.crc[g] = syn ? 1 : 0;
				a.conv[g] = (int32_t)fae;
			} else {
#pragma unroll
				for (int i = 0; i < 6; i++)
					l2w[i] = 0;
				a.crc[g] = -1;
				a.conv[g] = 0;
			}
		}
	}
}

__global__ __launch_bounds__(64) void k_l1(L1Args a)
{
	__shared__ __align__(16) int8_t s_eb[kEbitsLds];
	__shared__ __align__(16) uint32_t s_bm[4 * kSteps12];
	__shared__ __align__(16) uint64_t s_surv[kSteps12];
	const int lane = threadIdx.x;
	const int g0 = blockIdx.x * 4;
	const int neb = a.chain == kChainCcch ? 432 : 424;
	const int off = a.chain == kChainCcch ? 4 : 0;

	for (int q = 0; q < 4; q++) {
		const int g = g0 + q;
		WSYNC();
		if (g < a.n) {
			const int8_t *src = a.ebits + (size_t)g * neb;
			for (int i = lane; i < neb; i += 64)
				s_eb[i] = src[i];
			WSYNC();
			branch_metrics_k5_12(s_eb, off, s_bm + q * kSteps12, lane);
		} else {
			for (int k = lane; k < kSteps12; k += 64)
				s_bm[q * kSteps12 + k] = 0;
		}
	}
	WSYNC();
	uint32_t words[7], syn, fae;
	decode4_k5_12(s_bm, s_surv, lane, words, syn, fae);
	const int g = g0 + (lane >> 4);
	if ((lane & 15) == 0 && g < a.n) {
		uint32_t *l2w = reinterpret_cast<uint32_t *>(a.l2 + (size_t)g * 24);
#pragma unroll
		for (int i = 0; i < 6; i++)
			l2w[i] = words[i];
		a.crc[g] = syn ? 1 : 0;
		a.conv[g] = (int32_t)fae;
	}
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static int g_max_len_cache = 468;

size_t rx_lds_bytes(int max_in_len)
{
	size_t off[6];
	return lds_layout(max_in_len, g_max_len_cache, true, off);
}

template <int NPL>
static hipError_t launch_rx_npl(const RxArgs &a, bool decode, int max_in_len, int max_len, hipStream_t stream)
{
	size_t off[6];
	const size_t lds = lds_layout(max_in_len, max_len, decode, off);
	if (decode) {
		const int grid = (a.n + 3) / 4;
		hipLaunchKernelGGL((k_rx<NPL, true>), dim3(grid), dim3(64), lds, stream, a, max_in_len, max_len);
	} else {
		hipLaunchKernelGGL((k_rx<NPL, false>), dim3(a.n), dim3(64), lds, stream, a, max_in_len, max_len);
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
	const int max_len = decode ? 234 : 468;
	if (max_in_len <= 1024)
		return launch_rx_npl<16>(a, decode, max_in_len, max_len, stream);
	return launch_rx_npl<32>(a, decode, max_in_len, max_len, stream);
}

hipError_t launch_l1(const L1Args &a, hipStream_t stream)
{
	if (a.n <= 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_l1, dim3((a.n + 3) / 4), dim3(64), 0, stream, a);
	return hipGetLastError();
}

}  // namespace gmr1
