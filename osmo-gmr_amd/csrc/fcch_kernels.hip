// fcch_kernels.hip -- FCCH acquisition kernels for gfx950 (MI355X).
//
//   k_fcch_sweep  : the rough sweep in ONE pass over the raw search window: per lag tile, statistics partials of its own
//                   span, every sps-th sample to LDS, the 117 / 468-tap real dual-chirp correlation of the RAW samples
//                   as a banded-Toeplitz product on the matrix cores (v_mfma_f32_16x16x4_f32), written out per lag
//                   (osmo_cxvec_sig_normalize is linear: the normalisation is applied afterwards; fcch.c:230-238)
//   k_fcch_energy : mean / deviation from all tiles' partials, the correction mu * sum_n r[n] e^{j fs n}, |.|^2 per lag,
//                   the tile's best 5-sample window (osmo_cxvec_peak_energy_find); in small launches the stream's last
//                   work-group also picks (below)
//   k_fcch_pick   : per stream: best window over tiles, energy centroid, toa (fcch.c:241)
//   k_fcch_multi  : gmr1_fcch_rough_multi's period fold, threshold and ranked de-duplicated peak list (fcch.c:341-496)
//   k_fcch_fine   : two wavefronts per burst: normalise, mix with up / down chirp (or the dual chirp for the SNR
//                   estimate), direct N-point DFT, 5-bin centroid / 6 largest bins (fcch.c:512-628 gmr1_fcch_fine,
//                   fcch.c:643-708 gmr1_fcch_snr)
//   k_fcch_stats, k_fcch_corr : the two-pass form of the rough sweep (round 4; profiling build, GMR1_HIP_FCCH_TWO_PASS)
//   acq_step1..4, k_acq_glue  : what gmr1_rx does between two sweeps of its acquisition (gmr1_rx.c:605-702), run by the
//                   producing sweep's last thread (AcqTail, fcch_acq.h) or as a launch of its own
//
// HBM traffic per 1-s stream (93 600 samples): 748.8 kB read once (x 1.065: the tiles' overlap), 187 kB of raw correlation
// written and read back, a few kB of partials.
#include <cstring>
#include "gmr1_dev.h"
#include "fcch_acq.h"
#include "profile_env.h"

namespace gmr1 {

static constexpr float kPif = 3.14159265358979323846f;

__constant__ FcchTables c_fcch;

hipError_t upload_fcch_tables(const FcchTables *host, hipStream_t stream)
{
	return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_fcch), host, sizeof(FcchTables), 0, hipMemcpyHostToDevice, stream);
}

// ---------------------------------------------------------------------------
// helpers (same forms as rx_kernels.hip)
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dppf(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum(float v)
{
	v += dppf<0xB1>(v);                 // xor 1
	v += dppf<0x4E>(v);                 // xor 2
	v += dppf<0x1B>(dppf<0x141>(v));    // xor 4
	v += dppf<0x128>(v);                // xor 8
	return v;
}
__device__ __forceinline__ float lane_val(float v, int l)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v)
{
	v = row_sum(v);
	return (lane_val(v, 0) + lane_val(v, 16)) + (lane_val(v, 32) + lane_val(v, 48));
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o);
	return v;
}
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
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---------------------------------------------------------------------------
// pass 1: statistics + decimation
//   grid (tiles, streams), 256 threads; tile = kStatSpan raw samples
// ---------------------------------------------------------------------------
static constexpr int kStatSpan = 8192;

__global__ __launch_bounds__(256) void k_fcch_stats(FcchRoughArgs a)
{
	const int s = blockIdx.y, tile = blockIdx.x;
	const float2 *__restrict__ in = a.iq + a.offset[s];
	float2 *__restrict__ dec = a.dec + (size_t)s * a.dec_stride;
	const int begin = tile * kStatSpan;
	const int end = min(begin + kStatSpan, a.len);
	const int ndec = a.len / a.sps;
	float sr = 0.f, si = 0.f, sq = 0.f;
	const bool vec_ok = ((a.offset[s] & 1ull) == 0) && a.sps == 4 && (end - begin) == kStatSpan;
	if (vec_ok) {
		// 16-byte loads: two samples per lane, 4 loads in flight; sample 2t is the decimated one
		// when (begin + 2t) % 4 == 0, i.e. on even t (begin is a multiple of 8192)
		const float4 *in4 = reinterpret_cast<const float4 *>(in + begin);
		const bool keep = (threadIdx.x & 1) == 0;
		// all 16 loads of the tile are issued before the first is consumed (the stream is read once:
		// nothing but memory-level parallelism hides the HBM latency)
		typedef float v4f __attribute__((ext_vector_type(4)));
		const v4f *in4v = reinterpret_cast<const v4f *>(in4);
		float4 p[kStatSpan / 512];
#pragma unroll
		for (int it = 0; it < kStatSpan / 512; it++) {
			const v4f t = __builtin_nontemporal_load(&in4v[it * 256 + (int)threadIdx.x]);
			p[it] = make_float4(t.x, t.y, t.z, t.w);
		}
#pragma unroll
		for (int it = 0; it < kStatSpan / 512; it++) {
			sr += p[it].x + p[it].z;
			si += p[it].y + p[it].w;
			sq = fmaf(p[it].x, p[it].x, fmaf(p[it].y, p[it].y, fmaf(p[it].z, p[it].z, fmaf(p[it].w, p[it].w, sq))));
			const int di = (begin + it * 512 + 2 * (int)threadIdx.x) >> 2;
			if (keep && di < ndec)
				dec[di] = make_float2(p[it].x, p[it].y);
		}
	} else {
		for (int i = begin + (int)threadIdx.x; i < end; i += 256) {
			const float2 v = in[i];
			sr += v.x;
			si += v.y;
			sq = fmaf(v.x, v.x, fmaf(v.y, v.y, sq));
			if (i % a.sps == 0 && i / a.sps < ndec)
				dec[i / a.sps] = v;
		}
	}
	__shared__ float red[3][4];
	sr = wave_sum(sr); si = wave_sum(si); sq = wave_sum(sq);
	const int wv = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) { red[0][wv] = sr; red[1][wv] = si; red[2][wv] = sq; }
	__syncthreads();
	if (threadIdx.x == 0) {
		float *p = a.partial + ((size_t)s * a.n_stat_tiles + tile) * 4;
		p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
		p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
		p[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
		p[3] = 0.f;
	}
}

// ---------------------------------------------------------------------------
// pass 2: correlation, energies, best 5-window per tile
//   grid (lag tiles, streams), 256 threads, 8 lags per thread
// ---------------------------------------------------------------------------
static constexpr int kLagsPerThread = 8;
static constexpr int kTileLags = 256 * kLagsPerThread;          // 2048
static constexpr int kTileStep = kTileLags - 4;                 // windows of 5 overlap the next tile by 4

// padded LDS index: one spare slot per 8 samples makes the 8-sample lane stride conflict-free
__host__ __device__ __forceinline__ constexpr int pad8(int i) { return i + (i >> 3); }

template <int NT>
__global__ __launch_bounds__(256) void k_fcch_corr(FcchRoughArgs a)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float2 *xs = reinterpret_cast<float2 *>(lds_raw);                    // pad8(kTileLags + NT + 8) samples
	float *en = reinterpret_cast<float *>(xs + pad8(kTileLags + NT + 8) + 1);   // kTileLags energies
	__shared__ float s_stat[4];
	__shared__ float s_best[4];
	__shared__ int s_bidx[4];

	const int s = blockIdx.y, tile = blockIdx.x;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int ndec = a.len / a.sps;
	const int nlags = ndec - NT + 1;
	const int m0 = tile * kTileStep;
	const float2 *__restrict__ dec = a.dec + (size_t)s * a.dec_stride;

	// ---- stream statistics from the pass-1 partials (every block redoes this tiny reduction)
	if (wv == 0) {
		double dr = 0.0, di = 0.0, dq = 0.0;
		for (int t = lane; t < a.n_stat_tiles; t += 64) {
			const float *p = a.partial + ((size_t)s * a.n_stat_tiles + t) * 4;
			dr += p[0]; di += p[1]; dq += p[2];
		}
		dr = wave_sum_d(dr); di = wave_sum_d(di); dq = wave_sum_d(dq);
		if (lane == 0) {
			const double n = (double)a.len;
			const double ar = dr / n, ai = di / n;
			// sigma^2 = mean |x - avg|^2 = mean |x|^2 - |avg|^2
			double var = dq / n - (ar * ar + ai * ai);
			if (var < 0.0) var = 0.0;
			float sd = sqrtf((float)var);
			if (sd == 0.0f) sd = 1.0f;
			s_stat[0] = (float)ar; s_stat[1] = (float)ai; s_stat[2] = 1.0f / sd;
		}
	}
	__syncthreads();
	const float avr = s_stat[0], avi = s_stat[1], inv = s_stat[2];
	const float fs = a.freq_shift ? a.freq_shift[s] : 0.0f;

	// ---- stage normalised (and frequency shifted) samples m0 .. m0 + kTileLags + NT - 2
	const int nstage = kTileLags + NT - 1;
	for (int i = tid; i < nstage; i += 256) {
		const int gi = m0 + i;
		float2 v = make_float2(0.f, 0.f);
		if (gi < ndec) {
			v = dec[gi];
			v.x = (v.x - avr) * inv;
			v.y = (v.y - avi) * inv;
			if (fs != 0.0f) {
				float sn, cs;
				sincos_fast(fs * (float)gi, sn, cs);
				v = cmul(v, make_float2(cs, sn));
			}
		}
		xs[pad8(i)] = v;
	}
	__syncthreads();

	// ---- 8 consecutive lags per thread, sliding 8-sample register window
	const float *__restrict__ ref = c_fcch.dual[a.tab];
	float2 acc[kLagsPerThread];
	float2 win[kLagsPerThread];
	const int base = tid * kLagsPerThread;
	static_assert(kLagsPerThread == 8, "the refill addresses below rely on a thread's lags starting on a multiple of 8");
	// pad8(i) = i + i / 8 is additive over multiples of 8: sample base + n0 + u + 8 (base, n0 multiples of 8, u < 8) sits at
	// pad8(base) + 9 (n0 / 8) + u + 9 -- a pointer that advances by 9 per round of eight taps and immediate offsets, instead of
	// an add, a shift and an add per tap (a quarter of the kernel's vector instructions)
	const float2 *__restrict__ wp = xs + pad8(base);
#pragma unroll
	for (int q = 0; q < kLagsPerThread; q++) {
		acc[q] = make_float2(0.f, 0.f);
		win[q] = wp[q];
	}
	wp += kLagsPerThread + 1;
	// tap n multiplies sample (lag + n): window slot q holds sample base + n + q
	// (kept rolled: fully unrolled the 117 taps need 256 VGPRs and one wave per SIMD)
#pragma unroll 1
	for (int n0 = 0; n0 < NT; n0 += kLagsPerThread) {
#pragma unroll
		for (int u = 0; u < kLagsPerThread; u++) {
			const int n = n0 + u;
			if (n < NT) {
				const float r = ref[n];
#pragma unroll
				for (int q = 0; q < kLagsPerThread; q++) {
					const float2 x = win[(u + q) % kLagsPerThread];
					acc[q].x = fmaf(r, x.x, acc[q].x);
					acc[q].y = fmaf(r, x.y, acc[q].y);
				}
				// slot u is now free: refill with sample base + n + 8
				win[u] = wp[u];
			}
		}
		wp += kLagsPerThread + 1;
	}
#pragma unroll
	for (int q = 0; q < kLagsPerThread; q++) {
		const int m = m0 + base + q;
		const float e = (m < nlags) ? fmaf(acc[q].x, acc[q].x, acc[q].y * acc[q].y) : -1.0f;
		en[base + q] = e;
		if (a.energy && m < nlags && (tile == 0 || base + q >= 4))
			a.energy[(size_t)s * a.energy_stride + m] = e;
	}
	__syncthreads();

	// ---- best 5-sample window starting inside this tile (first maximum wins)
	float bv = -1.0f;
	int bi = 0x7fffffff;
	for (int i = tid; i < kTileStep; i += 256) {
		const int m = m0 + i;
		if (m + 5 <= nlags) {
			float e = 0.f;
#pragma unroll
			for (int k = 0; k < 5; k++)
				e += en[i + k];
			if (e > bv) { bv = e; bi = m; }
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const float ov = __shfl_xor(bv, o);
		const int oi = __shfl_xor(bi, o);
		if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
	}
	if (lane == 0) { s_best[wv] = bv; s_bidx[wv] = bi; }
	__syncthreads();
	if (tid == 0) {
		for (int q = 1; q < 4; q++)
			if (s_best[q] > bv || (s_best[q] == bv && s_bidx[q] < bi)) { bv = s_best[q]; bi = s_bidx[q]; }
		float *o = a.tile_best + ((size_t)s * a.n_lag_tiles + tile) * 8;
		o[0] = bv;
		o[1] = __builtin_bit_cast(float, bi);
		if (bi != 0x7fffffff) {
			for (int k = 0; k < 5; k++)
				o[2 + k] = en[bi - m0 + k];
		}
	}
}

// ---------------------------------------------------------------------------
// The rough sweep in ONE pass over the raw samples (round 5; replaces k_fcch_stats + k_fcch_corr, which stay for A/B in the
// profiling build).  osmo_cxvec_sig_normalize is linear: with mean mu and deviation sigma of the whole window,
//     sum_n r[n] (x[m + n] - mu) / sigma  =  (sum_n r[n] x[m + n]  -  mu sum_n r[n]) / sigma
// (and with a frequency shift: x[i] e^{j fs i} in the first sum, mu e^{j fs m} sum_n r[n] e^{j fs n} for the second), so the
// correlation of the RAW decimated samples does not have to wait for the statistics of the whole window:
//
//   k_fcch_sweep   one work-group per tile of 2044 lags: the tile's raw samples once from HBM (16-byte non-temporal loads)
//                  -> sum x, sum |x|^2 partials of the tile's own span AND the decimated samples into LDS -> raw correlation
//                  of 2048 lags, 8 bytes per lag back to HBM (the place the decimated copy used to take).  The streaming
//                  read and the arithmetic now overlap inside one kernel instead of following each other in two.
//   k_fcch_energy  per tile: mean / deviation from the partials, the correction above, |.|^2 per lag (optionally stored:
//                  gmr1_fcch_rough_multi), best 5-lag energy window of the tile -- what the second half of k_fcch_corr did.
//
// The correlation itself runs on the matrix cores.  **This is the one place in the library that does, and it is an
// experiment the round-4 review asked for, outside north_star's "no MFMA"**: 16 consecutive lags of a real 117-tap FIR are a
// 16 x 132 banded Toeplitz matrix T[i][k] = r[k - i] times the 132 samples under them, and 16 such blocks side by side
// (lags m, m + 16, ...) make the other operand a 132 x 16 matrix W[k][j] = x[m + 16 j + k]: D = T W is
// v_mfma_f32_16x16x4_f32, 33 of them per 256 lags and component (re, im).  The FP32 matrix peak equals the FP32 vector peak
// on this part, so nothing is gained in arithmetic rate -- but one MFMA issues 1 024 multiply-adds from ONE issue slot,
// where the vector form spent 30 % of its issue slots on loads, staging and address arithmetic beside its FMAs (88 % VALU
// busy, profiles/r04v).  11 % of the multiply-adds hit the zeros of the band.  The sums are formed in the MFMA's order (k
// in fours), not the reference's n = 0 ... 116: energies differ in the last bits, `toa` does not (asserted on every stream
// of tests/test_gpu_fcch.py and of the bench).
//
// A non-finite sample now reaches the lags of its whole 16-lag block (0 x Inf), not only the lags whose taps cover it.
// ---------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ constexpr int pad16(int i) { return i + (i >> 4); }
template <int NT>
struct SweepDims {
	static constexpr int KP = ((NT + 15 + 3) / 4) * 4;              // Toeplitz width, a whole number of MFMA k-steps (132 / 484)
	static constexpr int NS = kTileLags + KP + 16;                  // decimated samples a tile stages
	static constexpr int RT = KP + 20;                              // chirp with 15 zeros in front, zeros behind
	static constexpr size_t lds = (size_t)pad16(NS) * 8 + (size_t)RT * 4;
};

// FOLD (large launches of streams of at most 64 tiles; the receive loop's acquisition -- a few hundred work-groups whose LATENCY
// counts -- keeps the two-kernel form: a folded tile spends two coherent store round trips before its matrix work and two
// coherent load round trips behind it, 14 us more on a 10 us kernel, measured): the tile also does what k_fcch_energy did with its lags -- the normalisation applied,
// |.|^2 per lag, the tile's best 5-lag window -- so that the raw correlation (8 bytes a lag written, and read back by the next
// kernel) never leaves the chip.  The normalisation needs the mean and deviation of the WHOLE window, i.e. every tile's
// partial sums: each tile publishes its own with coherent stores -- sum re, sum im, sum |x|^2, each word with the launch's
// epoch beside it -- as soon as it has read its samples, before the matrix work (half the kernel's time), and behind the
// matrix work reads the stream's records with coherent loads until every word of them carries this launch's epoch.  The tiles of a stream are consecutive
// work-groups and start together, so nothing waits in practice; the wait is BOUNDED all the same, and a tile that gives up
// writes its raw correlation and a mark in its result slot, for k_fcch_energy (started behind every folded sweep; a tile
// without the mark costs it one load) to finish the old way.
// The sums behind the statistics are formed exactly as k_fcch_energy forms them (same partials, same order, in double); the
// four lags a tile's windows share with the next tile are this tile's own here (the matrix product groups their taps
// differently than the neighbour's: last bits).
// (every word carries its epoch: the three sums go out as three 8-byte coherent stores {value, epoch} -- an aligned 8-byte store
// of one lane is one transaction --, fire and forget: the publishing wave does not wait for them, and a reader needs ONE round
// trip to see whether all three words of a record are this launch's)
struct FoldRec { float sr, si, sq; bool ok; };
__device__ __forceinline__ void fold_publish(float *rec, float sr, float si, float sq, uint32_t epoch)
{
	const unsigned long long a = (unsigned long long)__builtin_bit_cast(uint32_t, sr) | ((unsigned long long)epoch << 32);
	const unsigned long long b = (unsigned long long)__builtin_bit_cast(uint32_t, si) | ((unsigned long long)epoch << 32);
	const unsigned long long c = (unsigned long long)__builtin_bit_cast(uint32_t, sq) | ((unsigned long long)epoch << 32);
	asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\tglobal_store_dwordx2 %0, %2, off offset:8 sc0 sc1\n\t"
	             "global_store_dwordx2 %0, %3, off offset:16 sc0 sc1"
	             :: "v"(rec), "v"(a), "v"(b), "v"(c) : "memory");
}
__device__ __forceinline__ FoldRec fold_read(const float *rec, uint32_t epoch)
{
	unsigned long long a, b, c;
	asm volatile("global_load_dwordx2 %0, %3, off sc0 sc1\n\tglobal_load_dwordx2 %1, %3, off offset:8 sc0 sc1\n\t"
	             "global_load_dwordx2 %2, %3, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
	             : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(rec) : "memory");
	const bool ok = (uint32_t)(a >> 32) == epoch && (uint32_t)(b >> 32) == epoch && (uint32_t)(c >> 32) == epoch;
	return {__builtin_bit_cast(float, (uint32_t)a), __builtin_bit_cast(float, (uint32_t)b), __builtin_bit_cast(float, (uint32_t)c), ok};
}
#ifdef GMR1_HIP_PROFILE
__device__ float g_fold_dbg[2][64][4];         // stream 1's statistics as the folded sweep's tiles [0] / k_fcch_energy's groups [1] formed them
#endif
constexpr float kFoldGaveUp = -2.0f;            // tile_best[0] of a tile whose wait ran out (a window's energy is never negative)
constexpr int kFoldMaxTiles = 64;

__device__ inline void fcch_pick_body(const FcchRoughArgs &a, const AcqTail &tl, int s, int lane);      // (below)
constexpr int kPickStreams = 1 << 16;
__device__ unsigned int g_pick_count[kPickStreams];      // work-groups of the stream that are through (back to 0 by the last one)

template <int NT, bool FOLD = false>
__global__ __launch_bounds__(256) void k_fcch_sweep(FcchRoughArgs a)
{
	typedef SweepDims<NT> D;
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float2 *xs = reinterpret_cast<float2 *>(lds_raw);                   // pad16(NS) samples
	float *rt = reinterpret_cast<float *>(xs + pad16(D::NS));           // RT floats
	const int s = blockIdx.y, tile = blockIdx.x;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int sps = a.sps;
	const int ndec = a.len / sps;
	const int nlags = ndec - NT + 1;
	const int m0 = tile * kTileStep;
	const float2 *__restrict__ in = a.iq + a.offset[s];
	const float fs = a.freq_shift ? a.freq_shift[s] : 0.0f;
	const bool last = tile == (int)gridDim.x - 1;

	for (int i = tid; i < D::RT; i += 256) {
		const int n = i - 15;
		rt[i] = (n >= 0 && n < NT) ? c_fcch.dual[a.tab][n] : 0.0f;
	}
	// (slots past the window's last decimated sample are read by the MFMAs -- against zero taps, or for lags that are not
	// kept -- and must hold zeros, not whatever the LDS held: only the window's last tiles have any)
	for (int i = max(ndec - m0, 0) + tid; i < D::NS; i += 256)
		xs[pad16(i)] = make_float2(0.f, 0.f);

	// ---- the tile's raw samples: statistics over its own span, every sps-th sample to LDS
	const long long raw0 = (long long)m0 * sps;
	const long long own_end = last ? (long long)a.len : (long long)(m0 + kTileStep) * sps;
	const long long raw_end = min((long long)(m0 + D::NS) * sps, (long long)a.len);
	float sr = 0.f, si = 0.f, sq = 0.f;
	auto take = [&](long long j, float2 v) {
		if (j < own_end) {
			sr += v.x;
			si += v.y;
			sq = fmaf(v.x, v.x, fmaf(v.y, v.y, sq));
		}
	};
	auto keep = [&](int di, float2 v) {          // decimated sample di of the window
		if (di < ndec) {
			if (fs != 0.0f) {
				float sn, cs;
				sincos_fast(fs * (float)di, sn, cs);
				v = cmul(v, make_float2(cs, sn));
			}
			xs[pad16(di - m0)] = v;
		}
	};
	if (sps == 4 && ((a.offset[s] & 1ull) == 0)) {
		// 16-byte loads, two samples each; of a pair (2 t, 2 t + 1) counted from the tile's first sample (a multiple of 4
		// from the window's) sample 2 t is a kept one on even t
		const v4f *in4 = reinterpret_cast<const v4f *>(in + raw0);
		const long long npair = (raw_end - raw0) >> 1;
		constexpr int NB = 9;
		for (long long b0 = 0; b0 < npair; b0 += (long long)NB * 256) {
			v4f p[NB];
#pragma unroll
			for (int it = 0; it < NB; it++) {
				const long long t = b0 + (long long)it * 256 + tid;
				p[it] = t < npair ? __builtin_nontemporal_load(&in4[t]) : (v4f){0.f, 0.f, 0.f, 0.f};
			}
#pragma unroll
			for (int it = 0; it < NB; it++) {
				const long long t = b0 + (long long)it * 256 + tid;
				if (t < npair) {
					const long long j = raw0 + 2 * t;
					take(j, make_float2(p[it].x, p[it].y));
					take(j + 1, make_float2(p[it].z, p[it].w));
					if ((t & 1) == 0)
						keep((int)(j >> 2), make_float2(p[it].x, p[it].y));
				}
			}
		}
		if (((raw_end - raw0) & 1) && tid == 0) {
			const long long j = raw_end - 1;
			const float2 v = in[j];
			take(j, v);
			if ((j & 3) == 0)
				keep((int)(j >> 2), v);
		}
	} else {
		for (long long j = raw0 + tid; j < raw_end; j += 256) {
			const float2 v = in[j];
			take(j, v);
			if (j % sps == 0)
				keep((int)(j / sps), v);
		}
	}
	__shared__ float red[3][4];
	sr = wave_sum(sr); si = wave_sum(si); sq = wave_sum(sq);
	if (lane == 0) { red[0][wv] = sr; red[1][wv] = si; red[2][wv] = sq; }
	__syncthreads();
	if (tid == 0) {
		const float p0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
		const float p1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
		const float p2 = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
		float *p = a.partial + ((size_t)s * a.n_stat_tiles + tile) * 4;
		p[0] = p0; p[1] = p1; p[2] = p2; p[3] = 0.f;
		if constexpr (FOLD)
			fold_publish(a.fold_partial + ((size_t)s * a.n_stat_tiles + tile) * 8, p0, p1, p2, a.epoch);
	}

	// ---- raw correlation of lags m0 ... m0 + 2047: wave wv takes the sets wv and wv + 4 of 256 lags each
	// A (16 x 4 per step): lane (li, lk) holds T[li][k0 + lk] = r[k0 + lk - li]; B (4 x 16): lane (li, lk) holds
	// x[set + 16 li + k0 + lk]; D: lane (li, lk), register v = lag set + 16 li + 4 lk + v
	const int li = lane & 15, lk = lane >> 4;
	const float *rp = rt + 15 + lk - li;
	const float2 *xp0 = xs + pad16(256 * wv) + 17 * li + lk;             // (k0 + lk) >> 4 == k0 >> 4: lk < 4, k0 in fours
	const float2 *xp1 = xs + pad16(256 * (wv + 4)) + 17 * li + lk;
	v4f dr0 = {0.f, 0.f, 0.f, 0.f}, di0 = dr0, dr1 = dr0, di1 = dr0;
	constexpr int NG = D::KP / 16, REM = (D::KP % 16) / 4;               // groups of four k-steps (16 samples = 17 padded slots)
#pragma unroll 2
	for (int g = 0; g < NG; g++) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const float r = rp[16 * g + 4 * u];
			const float2 x0 = xp0[17 * g + 4 * u], x1 = xp1[17 * g + 4 * u];
			dr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x0.x, dr0, 0, 0, 0);
			di0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x0.y, di0, 0, 0, 0);
			dr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x1.x, dr1, 0, 0, 0);
			di1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x1.y, di1, 0, 0, 0);
		}
	}
#pragma unroll
	for (int u = 0; u < REM; u++) {
		const float r = rp[16 * NG + 4 * u];
		const float2 x0 = xp0[17 * NG + 4 * u], x1 = xp1[17 * NG + 4 * u];
		dr0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x0.x, dr0, 0, 0, 0);
		di0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x0.y, di0, 0, 0, 0);
		dr1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x1.x, dr1, 0, 0, 0);
		di1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, x1.y, di1, 0, 0, 0);
	}
	float2 *__restrict__ acc = a.dec + (size_t)s * a.dec_stride;
	__shared__ float s_stat[4];
	__shared__ int s_have;
	if constexpr (FOLD) {
		// ---- the window's statistics, once every tile's record carries this launch's epoch (wave 0; bounded)
		if (wv == 0) {
			const size_t slot = (size_t)s * a.n_stat_tiles + (lane < a.n_stat_tiles ? lane : 0);
			FoldRec r = {0.f, 0.f, 0.f, false};
			bool have = false;
			for (int poll = 0; poll < a.fold_polls; poll++) {
				r = fold_read(a.fold_partial + slot * 8, a.epoch);
				have = __ballot(lane < a.n_stat_tiles && !r.ok) == 0;
				if (have)
					break;
				__builtin_amdgcn_s_sleep(4);
			}
			// (k_fcch_energy's sums: lane t holds tile t's partial, the rest zero; the same butterfly in double)
			double dr = lane < a.n_stat_tiles ? (double)r.sr : 0.0, di = lane < a.n_stat_tiles ? (double)r.si : 0.0,
			       dq = lane < a.n_stat_tiles ? (double)r.sq : 0.0;
			dr = wave_sum_d(dr); di = wave_sum_d(di); dq = wave_sum_d(dq);
			float rr = 0.f, ri = 0.f;
			for (int n = lane; n < NT; n += 64) {
				const float r1 = c_fcch.dual[a.tab][n];
				float sn = 0.f, cs = 1.f;
				if (fs != 0.0f)
					sincos_fast(fs * (float)n, sn, cs);
				rr = fmaf(r1, cs, rr);
				ri = fmaf(r1, sn, ri);
			}
			rr = wave_sum(rr); ri = fs != 0.0f ? wave_sum(ri) : 0.0f;
			if (lane == 0) {
				const double n = (double)a.len;
				const double ar = dr / n, ai = di / n;
				double var = dq / n - (ar * ar + ai * ai);
				if (var < 0.0) var = 0.0;
				float sd = sqrtf((float)var);
				if (sd == 0.0f) sd = 1.0f;
				s_stat[2] = 1.0f / sd;
				s_stat[0] = (float)ar * rr - (float)ai * ri;
				s_stat[1] = (float)ar * ri + (float)ai * rr;
				s_have = have ? 1 : 0;
#ifdef GMR1_HIP_PROFILE
				if (s == 1 && tile < 64) {
					g_fold_dbg[0][tile][0] = s_stat[0]; g_fold_dbg[0][tile][1] = s_stat[1]; g_fold_dbg[0][tile][2] = s_stat[2];
					g_fold_dbg[0][tile][3] = (float)dr;
				}
#endif
			}
		}
		__syncthreads();                                  // (also: every wave is through with the staged samples)
	}
	const bool folded = FOLD && s_have != 0;
	if (!folded) {
		// each lane holds four consecutive lags: 32 bytes; the tile's own lags only (the next tile writes the four beyond)
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const v4f dr = h ? dr1 : dr0, di = h ? di1 : di0;
			const int i0 = 256 * (wv + 4 * h) + 16 * li + 4 * lk;
#pragma unroll
			for (int v = 0; v < 4; v++) {
				const int i = i0 + v, m = m0 + i;
				if (i < kTileStep && m < nlags)
					acc[m] = make_float2(dr[v], di[v]);
			}
		}
		if constexpr (FOLD) {
			if (tid == 0)
				a.tile_best[((size_t)s * a.n_lag_tiles + tile) * 8] = kFoldGaveUp;
		}
	}
	if constexpr (FOLD) {
		if (folded) {
			// ---- energies of the tile's 2048 lags over the staged samples (dead), then its best 5-lag window: k_fcch_energy's steps
			float *en = reinterpret_cast<float *>(lds_raw);
			const float mr = s_stat[0], mi = s_stat[1], inv = s_stat[2];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const v4f dr = h ? dr1 : dr0, di = h ? di1 : di0;
				const int i0 = 256 * (wv + 4 * h) + 16 * li + 4 * lk;
#pragma unroll
				for (int v = 0; v < 4; v++) {
					const int i = i0 + v, m = m0 + i;
					float e = -1.0f;
					if (m < nlags) {
						float2 mu = make_float2(mr, mi);
						if (fs != 0.0f) {
							float sn, cs;
							sincos_fast(fs * (float)m, sn, cs);
							mu = cmul(mu, make_float2(cs, sn));
						}
						const float cx = (dr[v] - mu.x) * inv, cy = (di[v] - mu.y) * inv;
						e = fmaf(cx, cx, cy * cy);
						if (a.energy && i < kTileStep)
							a.energy[(size_t)s * a.energy_stride + m] = e;
					}
					en[i] = e;
				}
			}
			__syncthreads();
			__shared__ float s_best[4];
			__shared__ int s_bidx[4];
			float bv = -1.0f;
			int bi = 0x7fffffff;
			for (int i = tid; i < kTileStep; i += 256) {
				const int m = m0 + i;
				if (m + 5 <= nlags) {
					float e = 0.f;
#pragma unroll
					for (int k = 0; k < 5; k++)
						e += en[i + k];
					if (e > bv) { bv = e; bi = m; }
				}
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float ov = __shfl_xor(bv, o);
				const int oi = __shfl_xor(bi, o);
				if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
			}
			if (lane == 0) { s_best[wv] = bv; s_bidx[wv] = bi; }
			__syncthreads();
			if (tid == 0) {
				for (int q = 1; q < 4; q++)
					if (s_best[q] > bv || (s_best[q] == bv && s_bidx[q] < bi)) { bv = s_best[q]; bi = s_bidx[q]; }
				float *o = a.tile_best + ((size_t)s * a.n_lag_tiles + tile) * 8;
				o[0] = bv;
				o[1] = __builtin_bit_cast(float, bi);
				if (bi != 0x7fffffff) {
					for (int k = 0; k < 5; k++)
						o[2 + k] = en[bi - m0 + k];
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------
// The arithmetic gmr1_rx's acquisition does between two sweeps (fcch_single_init gmr1_rx.c:605-639, fcch_multi_process
// :643-702), per carrier `k` / candidate slot `t`, from the sweep's result the calling thread holds:
//   step 1  after the 330 ms rough sweep : align += toa, bounds, window of the fine stage
//   step 2  after the fine stage         : align += toa, freq_err, base_align, window + shift of the 650 ms sweep
//   step 3  after rough_multi            : one window per candidate slot (bounds: a candidate outside drops the carrier)
//   step 4  after fine over the slots    : refined window + shift of the SNR stage
// Called by the producing sweep's last thread (AcqTail, fcch_acq.h) or, one launch each, by k_acq_glue.
// ---------------------------------------------------------------------------
__device__ inline void acq_step1(const AcqArgs &a, int k, int toa1, int rv1)
{
	const int64_t len = (int64_t)a.len[k];
	int stat = a.stat[k];
	if (!stat) {
		if (rv1) {
			stat = rv1;
		} else {
			a.align[k] += toa1;
			if ((int64_t)a.align[k] + a.flen > len)
				stat = -1;
		}
	}
	a.stat[k] = stat;
	a.off[k] = a.base[k] + (uint64_t)(stat ? 0 : a.align[k]);
}

__device__ inline void acq_step2(const AcqArgs &a, int k, int ftoa, float fe)
{
	const int64_t len = (int64_t)a.len[k];
	int stat = a.stat[k];
	float fs = 0.f;
	if (!stat) {
		a.align[k] += ftoa;
		a.ferr[k] = fe;
		int ba = a.align[k] - a.flen;
		if (ba < 0) ba = 0;
		a.base_align[k] = ba;
		if (!a.can3[k] || (int64_t)ba + a.wl3 > len)
			stat = -1;
		fs = -fe;
	}
	a.stat[k] = stat;
	a.off[k] = a.base[k] + (uint64_t)((stat || !a.can3[k]) ? 0 : a.base_align[k]);
	a.fs[k] = fs;
}

// (peaks: the carrier's candidate list, kAcqPeaks wide -- global memory or LDS; cnt: rough_multi's count)
__device__ inline void acq_step3(const AcqArgs &a, int k, int j, const int32_t *peaks, int cnt)
{
	const int t = k * kAcqPeaks + j;
	const int64_t len = (int64_t)a.len[k];
	int stat = a.stat[k];
	bool live = false;
	int64_t p = 0;
	if (!stat) {
		if (cnt < 0) {
			stat = cnt;
		} else {
			// a carrier with any candidate outside its samples is dropped as a whole
			bool ok = true;
			for (int q = 0; q < cnt; q++) {
				const int64_t pq = (int64_t)a.base_align[k] + peaks[q];
				if (pq < 0 || pq + a.flen > len) ok = false;
			}
			if (!ok)
				stat = -1;
			else if (j < cnt) {
				live = true;
				p = (int64_t)a.base_align[k] + peaks[j];
			}
		}
	}
	a.live[t] = live ? 1 : 0;
	a.off[t] = a.base[k] + (uint64_t)(live ? p : 0);
	a.fs[t] = live ? -a.ferr[k] : 0.f;
	// (every slot of the carrier computes the same verdict; slot 0 records it -- behind the others' reads of it only when
	// they are lanes of one wave or threads of one launch that read before any writes: callers see to that)
	if (j == 0 && stat)
		a.stat[k] = stat;
}

__device__ inline void acq_step4(const AcqArgs &a, int t, int ctoa, float cfe)
{
	const int k = t / kAcqPeaks, j = t % kAcqPeaks;
	const int64_t len = (int64_t)a.len[k];
	const bool live = a.live[t] != 0;
	int64_t p = 0;
	if (live)
		p = (int64_t)a.base_align[k] + a.peaks[(size_t)k * kAcqPeaks + j] + ctoa;
	const bool inside = live && p >= 0 && p + a.flen <= len;
	a.off[t] = a.base[k] + (uint64_t)(inside ? p : 0);
	a.fs[t] = inside ? -(a.ferr[k] + cfe) : 0.f;
}

// second half: the normalisation applied to the raw correlation, energies, the tile's best 5-lag window
// (tiles: lag tiles per work-group -- several where there are thousands of them, so that the statistics are formed once for
// all; one where the launch is small and its latency is what counts: the receive loop's acquisition)
// (fold: behind a folded sweep -- only the tiles that gave up there are left to do)
template <int NT>
__global__ __launch_bounds__(256) void k_fcch_energy(FcchRoughArgs a, int kEnergyTiles, int pick, AcqTail tl, int fold)
{
	__shared__ float en[kTileLags];
	__shared__ int s_last;
	__shared__ float s_stat[8];
	__shared__ float s_best[4];
	__shared__ int s_bidx[4];
	const int s = blockIdx.y;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int ndec = a.len / a.sps;
	const int nlags = ndec - NT + 1;
	const float fs = a.freq_shift ? a.freq_shift[s] : 0.0f;
	// behind a folded sweep: is any of this work-group's tiles left?  (the marks were written by the launch before: plain loads)
	bool todo = !fold;
	if (fold) {
		for (int tile = blockIdx.x * kEnergyTiles; tile < min(((int)blockIdx.x + 1) * kEnergyTiles, a.n_lag_tiles); tile++)
			todo = todo || a.tile_best[((size_t)s * a.n_lag_tiles + tile) * 8] == kFoldGaveUp;
	}
	if (todo) {
	if (wv == 0) {
		double dr = 0.0, di = 0.0, dq = 0.0;
		for (int t = lane; t < a.n_stat_tiles; t += 64) {
			const float *p = a.partial + ((size_t)s * a.n_stat_tiles + t) * 4;
			dr += p[0]; di += p[1]; dq += p[2];
		}
		dr = wave_sum_d(dr); di = wave_sum_d(di); dq = wave_sum_d(dq);
		// sum_n r[n] e^{j fs n}: what the mean contributes to every lag (times e^{j fs m})
		float rr = 0.f, ri = 0.f;
		for (int n = lane; n < NT; n += 64) {
			const float r = c_fcch.dual[a.tab][n];
			float sn = 0.f, cs = 1.f;
			if (fs != 0.0f)
				sincos_fast(fs * (float)n, sn, cs);
			rr = fmaf(r, cs, rr);
			ri = fmaf(r, sn, ri);
		}
		rr = wave_sum(rr); ri = fs != 0.0f ? wave_sum(ri) : 0.0f;
		if (lane == 0) {
			const double n = (double)a.len;
			const double ar = dr / n, ai = di / n;
			double var = dq / n - (ar * ar + ai * ai);
			if (var < 0.0) var = 0.0;
			float sd = sqrtf((float)var);
			if (sd == 0.0f) sd = 1.0f;
			s_stat[2] = 1.0f / sd;
			// mu * sum_n r[n] e^{j fs n}
			s_stat[0] = (float)ar * rr - (float)ai * ri;
			s_stat[1] = (float)ar * ri + (float)ai * rr;
#ifdef GMR1_HIP_PROFILE
			if (s == 1 && blockIdx.x < 64) {
				g_fold_dbg[1][blockIdx.x][0] = s_stat[0]; g_fold_dbg[1][blockIdx.x][1] = s_stat[1]; g_fold_dbg[1][blockIdx.x][2] = s_stat[2];
				g_fold_dbg[1][blockIdx.x][3] = (float)dr;
			}
#endif
		}
	}
	__syncthreads();
	const float mr = s_stat[0], mi = s_stat[1], inv = s_stat[2];
	const float2 *__restrict__ acc = a.dec + (size_t)s * a.dec_stride;
	// (kEnergyTiles lag tiles per work-group: the statistics above are formed once for them)
	for (int tile = blockIdx.x * kEnergyTiles; tile < min(((int)blockIdx.x + 1) * kEnergyTiles, a.n_lag_tiles); tile++) {
	if (fold && a.tile_best[((size_t)s * a.n_lag_tiles + tile) * 8] != kFoldGaveUp)
		continue;                                   // (uniform: the sweep finished this tile itself)
	const int m0 = tile * kTileStep;
	__syncthreads();
	// two lags per lane and load: 16 bytes (the stream's array, the tile's first lag and 2 tid are all even)
	for (int i = 2 * tid; i < kTileLags; i += 512) {
		const int m = m0 + i;
		float2 c2[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
		if (m + 1 < nlags) {
			const float4 q = *reinterpret_cast<const float4 *>(acc + m);
			c2[0] = make_float2(q.x, q.y);
			c2[1] = make_float2(q.z, q.w);
		} else if (m < nlags) {
			c2[0] = acc[m];
		}
#pragma unroll
		for (int h = 0; h < 2; h++) {
			float e = -1.0f;
			if (m + h < nlags) {
				float2 mu = make_float2(mr, mi);
				if (fs != 0.0f) {
					float sn, cs;
					sincos_fast(fs * (float)(m + h), sn, cs);
					mu = cmul(mu, make_float2(cs, sn));
				}
				const float cx = (c2[h].x - mu.x) * inv, cy = (c2[h].y - mu.y) * inv;
				e = fmaf(cx, cx, cy * cy);
				if (a.energy && i + h < kTileStep)
					a.energy[(size_t)s * a.energy_stride + m + h] = e;
			}
			en[i + h] = e;
		}
	}
	__syncthreads();
	// ---- best 5-sample window starting inside this tile (first maximum wins)
	float bv = -1.0f;
	int bi = 0x7fffffff;
	for (int i = tid; i < kTileStep; i += 256) {
		const int m = m0 + i;
		if (m + 5 <= nlags) {
			float e = 0.f;
#pragma unroll
			for (int k = 0; k < 5; k++)
				e += en[i + k];
			if (e > bv) { bv = e; bi = m; }
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const float ov = __shfl_xor(bv, o);
		const int oi = __shfl_xor(bi, o);
		if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
	}
	if (lane == 0) { s_best[wv] = bv; s_bidx[wv] = bi; }
	__syncthreads();
	if (tid == 0) {
		for (int q = 1; q < 4; q++)
			if (s_best[q] > bv || (s_best[q] == bv && s_bidx[q] < bi)) { bv = s_best[q]; bi = s_bidx[q]; }
		float *o = a.tile_best + ((size_t)s * a.n_lag_tiles + tile) * 8;
		o[0] = bv;
		o[1] = __builtin_bit_cast(float, bi);
		if (bi != 0x7fffffff) {
			for (int k = 0; k < 5; k++)
				o[2 + k] = en[bi - m0 + k];
		}
	}
	}
	}
	if (!pick)
		return;
	// the stream's last work-group picks (thread 0 wrote this group's tiles: its fence, then its count); small launches only,
	// which are never folded
	if (tid == 0) {
		__threadfence();
		s_last = atomicAdd(&g_pick_count[s], 1u) == gridDim.x - 1 ? 1 : 0;
	}
	__syncthreads();
	if (!s_last || wv != 0)
		return;
	if (lane == 0)
		g_pick_count[s] = 0;
	__threadfence();
	fcch_pick_body(a, tl, s, lane);
}

// ---------------------------------------------------------------------------
// pass 3: per stream, best tile -> centroid -> toa    (one wavefront per stream)
// Run by the LAST work-group of k_fcch_energy to finish the stream (a counter per stream tells which: one launch fewer in
// a chain of small dependent ones), or as the kernel k_fcch_pick.  The tiles' results were written by other work-groups of
// the same launch: they are read with device-scope loads.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float ld_dev(const float *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void fcch_pick_body(const FcchRoughArgs &a, const AcqTail &tl, int s, int lane)
{
	float bv = -1.0f;
	int bi = 0x7fffffff, bt = 0;
	for (int t = lane; t < a.n_lag_tiles; t += 64) {
		const float *o = a.tile_best + ((size_t)s * a.n_lag_tiles + t) * 8;
		const float v = ld_dev(o);
		const int i = __builtin_bit_cast(int, ld_dev(o + 1));
		if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; bt = t; }
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const float ov = __shfl_xor(bv, o);
		const int oi = __shfl_xor(bi, o);
		const int ot = __shfl_xor(bt, o);
		if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bt = ot; }
	}
	// PEAK_WEIGH_WIN: energy-weighted centroid of the 5 samples (lanes 0..4 fetch them at once, lane 0 sums in order)
	float e5 = 0.f;
	if (bi != 0x7fffffff && lane < 5)
		e5 = ld_dev(a.tile_best + ((size_t)s * a.n_lag_tiles + bt) * 8 + 2 + lane);
	float num = 0.f, den = 0.f;
	for (int k = 0; k < 5; k++) {
		const float e = __shfl(e5, k);
		num += e * (float)(bi + k);
		den += e;
	}
	if (lane == 0) {
		int toa = 0, rv = 0;
		if (bi == 0x7fffffff) {
			rv = -22;
		} else {
			const float pos = num / den;
			toa = (int)round((double)(pos * (float)a.sps));     // fcch.c:241
		}
		a.toa[s] = toa;
		if (a.rv) a.rv[s] = rv;
		if (tl.step == 1)
			acq_step1(tl.g, s, toa, rv);
	}
}

__global__ __launch_bounds__(64) void k_fcch_pick(FcchRoughArgs a, AcqTail tl)
{
	fcch_pick_body(a, tl, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------
// fine acquisition / SNR: one wavefront per burst of exactly len*sps samples
// ---------------------------------------------------------------------------
// mode 0: gmr1_fcch_fine (toa, freq_error) ; mode 1: gmr1_fcch_snr
// kFineWaves wavefronts per burst share the DFT's bins (a 117-point burst: one bin per lane of two waves instead of two
// passes of one -- the acquisition chain of the receive loop is a handful of these small dependent launches, so their
// latency counts); the statistics are formed by every wave for itself, lane for lane as before, and the ordered tail
// (peak search, centroid) is wave 0's: every sum keeps its order, the results are bit for bit those of one wave.
constexpr int kFineWaves = 2;
template <int N>
__global__ __launch_bounds__(64 * kFineWaves) void k_fcch_fine(FcchFineArgs a, AcqTail tl)
{
	if (tl.skip_dead && !tl.skip_dead[blockIdx.x])
		return;                              // a candidate slot without a candidate (the acquisition chain's lists)
	constexpr int NT = 64 * kFineWaves;
	constexpr int PER = (N + NT - 1) / NT;   // bins per thread
	__shared__ float2 s_up[N], s_dn[N];
	__shared__ float2 s_tw[N];               // the DFT's twiddles: read N times per bin, so from LDS, not from memory
	__shared__ float s_e[2][N];
	const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int sps = a.sps, nraw = N * sps;
	const float2 *__restrict__ in = a.iq + a.offset[b];
	const float fs = a.freq_shift ? a.freq_shift[b] : 0.0f;
	const int tab = a.tab;

	// Everything this work-group reads from memory is asked for up front, in one round: the kernel sits in the receive
	// loop's acquisition chain three times and its dependent memory round trips were most of its time.  Windows to 512
	// samples (117 symbols at <= 4 samples per symbol) are held in registers for both statistics passes.
	constexpr int kRawRegs = 8;
	const bool in_regs = nraw <= 64 * kRawRegs;
	float2 raw[kRawRegs];
	if (in_regs) {
#pragma unroll
		for (int j = 0; j < kRawRegs; j++) {
			const int i = lane + 64 * j;
			raw[j] = i < nraw ? in[i] : make_float2(0.f, 0.f);
		}
	}
	float2 mine[PER], t_tw[PER], t_up[PER], t_shf[PER];
	float t_dual[PER];
#pragma unroll
	for (int p = 0; p < PER; p++) {
		const int i = tid + NT * p;
		mine[p] = t_tw[p] = t_up[p] = t_shf[p] = make_float2(0.f, 0.f);
		t_dual[p] = 0.f;
		if (i < N) {
			mine[p] = in[i * sps];
			t_tw[p] = c_fcch.twid[tab][i];
			if (a.mode == 0) {
				t_up[p] = c_fcch.up[tab][i];
				t_shf[p] = c_fcch.shift[tab][i];
			} else {
				t_dual[p] = c_fcch.dual[tab][i];
			}
		}
	}

	// statistics over all raw samples (osmo_cxvec_sig_normalize): lane i takes samples i, i + 64, ... in that order
	float sr = 0.f, si = 0.f;
	if (in_regs) {
#pragma unroll
		for (int j = 0; j < kRawRegs; j++)
			if (lane + 64 * j < nraw) { sr += raw[j].x; si += raw[j].y; }
	} else {
		for (int i = lane; i < nraw; i += 64) {
			const float2 v = in[i];
			sr += v.x; si += v.y;
		}
	}
	sr = wave_sum(sr); si = wave_sum(si);
	const float avr = sr / (float)nraw, avi = si / (float)nraw;
	float sq = 0.f;
	if (in_regs) {
#pragma unroll
		for (int j = 0; j < kRawRegs; j++)
			if (lane + 64 * j < nraw) {
				const float dx = raw[j].x - avr, dy = raw[j].y - avi;
				sq = fmaf(dx, dx, fmaf(dy, dy, sq));
			}
	} else {
		for (int i = lane; i < nraw; i += 64) {
			const float2 v = in[i];
			const float dx = v.x - avr, dy = v.y - avi;
			sq = fmaf(dx, dx, fmaf(dy, dy, sq));
		}
	}
	float sd = sqrtf(wave_sum(sq) / (float)nraw);
	if (sd == 0.0f) sd = 1.0f;
	const float inv = 1.0f / sd;

	// normalise, shift, mix
#pragma unroll
	for (int p = 0; p < PER; p++) {
		const int i = tid + NT * p;
		if (i >= N)
			continue;
		s_tw[i] = t_tw[p];
		float2 v = mine[p];
		v.x = (v.x - avr) * inv;
		v.y = (v.y - avi) * inv;
		if (fs != 0.0f) {
			float sn, cs;
			sincos_fast(fs * (float)i, sn, cs);
			v = cmul(v, make_float2(cs, sn));
		}
		if (a.mode == 0) {
			// burst * ref_up / ref_down, then centre the spectrum on bin N/2 (fcch.c:563-580)
			const float2 shf = t_shf[p];
			s_up[i] = cmul(cmul(v, t_up[p]), shf);
			const float2 dn = make_float2(t_up[p].x, -t_up[p].y);   // down = conj(up)
			s_dn[i] = cmul(cmul(v, dn), shf);
		} else {
			const float r = t_dual[p];
			s_up[i] = make_float2(v.x * r, v.y * r);
		}
	}
	__syncthreads();

	// direct DFT, bins tid, tid + NT, ...   (sums over n ascending, per bin, as before; the two modes as loops of their own)
#pragma unroll
	for (int p = 0; p < PER; p++) {
		const int k = tid + NT * p;
		float2 au = make_float2(0.f, 0.f), ad = make_float2(0.f, 0.f);
		if (k < N) {
			int idx = 0;
			if (a.mode == 0) {
#pragma unroll 9
				for (int n = 0; n < N; n++) {
					const float2 tw = s_tw[idx];
					const float2 u = s_up[n];
					const float2 dd = s_dn[n];
					au.x = fmaf(u.x, tw.x, fmaf(-u.y, tw.y, au.x));
					au.y = fmaf(u.x, tw.y, fmaf(u.y, tw.x, au.y));
					ad.x = fmaf(dd.x, tw.x, fmaf(-dd.y, tw.y, ad.x));
					ad.y = fmaf(dd.x, tw.y, fmaf(dd.y, tw.x, ad.y));
					idx += k;
					idx -= idx >= N ? N : 0;
				}
			} else {
#pragma unroll 9
				for (int n = 0; n < N; n++) {
					const float2 tw = s_tw[idx];
					const float2 u = s_up[n];
					au.x = fmaf(u.x, tw.x, fmaf(-u.y, tw.y, au.x));
					au.y = fmaf(u.x, tw.y, fmaf(u.y, tw.x, au.y));
					idx += k;
					idx -= idx >= N ? N : 0;
				}
			}
			s_e[0][k] = fmaf(au.x, au.x, au.y * au.y);
			s_e[1][k] = fmaf(ad.x, ad.x, ad.y * ad.y);
		}
	}
	__syncthreads();

	if (a.mode == 0) {
		if (wave != 0)
			return;                          // (no barrier below in this branch)
		// 5-bin energy window centroid of both spectra (PEAK_WEIGH_WIN)
		float peak[2];
#pragma unroll
		for (int h = 0; h < 2; h++) {
			float bv = -1.0f;
			int bi = 0x7fffffff;
			for (int m = lane; m + 5 <= N; m += 64) {
				float e = 0.f;
				for (int k = 0; k < 5; k++)
					e += s_e[h][m + k];
				if (e > bv) { bv = e; bi = m; }
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float ov = __shfl_xor(bv, o);
				const int oi = __shfl_xor(bi, o);
				if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
			}
			if (bi == 0x7fffffff) bi = 0;
			float num = 0.f, den = 0.f;
			for (int k = 0; k < 5; k++) {
				num += s_e[h][bi + k] * (float)(bi + k);
				den += s_e[h][bi + k];
			}
			peak[h] = num / den;
		}
		if (lane == 0) {
			// fcch.c:600-615
			const int mid = N >> 1;
			const float bin_hz = 23400.0f / (float)N;
			const float pu = (peak[0] - (float)mid) * bin_hz;
			const float pd = (peak[1] - (float)mid) * bin_hz;
			const float ferr_hz = (pu + pd) / 2.0f;
			a.freq_err[b] = (2.0f * kPif * ferr_hz) / 23400.0f;      // (fe below: the same expression)
			const float chirp_rate = (2.0f * c_fcch.freq[tab] * 23400.0f * 23400.0f) / (float)(N * 1000);
			const float toa_ms = ((pu - pd) / 2.0f) / chirp_rate;
			const float toa_samples = (toa_ms * 23400.0f * (float)sps) / 1000.0f;
			const int toa = (int)round((double)toa_samples);
			const float fe = (2.0f * kPif * ferr_hz) / 23400.0f;
			a.toa[b] = toa;
			if (tl.step == 2)
				acq_step2(tl.g, b, toa, fe);
			else if (tl.step == 4)
				acq_step4(tl.g, b, toa, fe);
		}
	} else {
		// 6 largest bins, descending, first index wins ties (osmo_cxvec_peaks_scan)
		float top[6];
		for (int r = 0; r < 6; r++) {
			float bv = -1.0f;
			int bi = 0x7fffffff;
			// (every wave scans all bins, lane for lane as one wave would: the barriers below are the work-group's)
			for (int k = lane; k < N; k += 64) {
				const float e = s_e[0][k];
				if (e > bv) { bv = e; bi = k; }
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const float ov = __shfl_xor(bv, o);
				const int oi = __shfl_xor(bi, o);
				if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
			}
			top[r] = bv;
			__syncthreads();
			if (tid == 0 && bi != 0x7fffffff)
				s_e[0][bi] = -2.0f;          // taken
			__syncthreads();
		}
		if (tid == 0)
			a.snr[b] = (top[0] + top[1]) / (top[4] + top[5]);      // fcch.c:701-702
	}
}

// ---------------------------------------------------------------------------
// multi-FCCH detection on the correlation power of a >= 650 ms window
// (reference src/sdr/fcch.c:341-496 gmr1_fcch_rough_multi, :264-326 _peak_record)
//   one 256-thread work-group per stream; the O(N) reductions are parallel, the short
//   ordered tail (peak list maintenance) is done by lane 0 exactly as the reference does
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red, int tid)
{
	v = wave_sum(v);
	__syncthreads();
	if ((tid & 63) == 0) red[tid >> 6] = v;
	__syncthreads();
	return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_fcch_multi(FcchMultiArgs a, AcqTail tl)
{
	extern __shared__ __align__(16) unsigned char lds_raw[];
	float *v = reinterpret_cast<float *>(lds_raw);          // Lw mixed-cycle values
	__shared__ float red[4];
	__shared__ float s_bv[4];
	__shared__ int s_bi[4];
	__shared__ int s_scal[4];
	__shared__ unsigned int s_flags[256];                   // Lw <= 8192 threshold flags
	// the ranked list while it is being kept: LDS, not the caller's array and not a private one -- lane 0 goes through it
	// once per rising edge, and every trip to memory would be on the chain's critical path
	__shared__ int s_toa[32];
	__shared__ float s_pwr[32];
	__shared__ int s_mod[32];                               // toa % Lp of the entries (what _peak_record compares)

	const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const float *__restrict__ cp = a.energy + (size_t)s * a.energy_stride;
	const int cl = a.nlags;
	const int Lw = a.Lw;
	int Lp = a.Lp;

	// ---- strongest lag within the first Lw (first maximum, must exceed 0)
	// (the thread's <= 32 energies of the first cycle are asked for in one round and kept: the mix below wants them again --
	// a loop of dependent loads here was most of this kernel's time, and the kernel is on the receive loop's acquisition chain)
	constexpr int kRegs = 32;                // Lw <= 8192 = 256 x 32 (launch_fcch_multi refuses more)
	float e0[kRegs];
#pragma unroll
	for (int j = 0; j < kRegs; j++) {
		const int i = tid + 256 * j;
		e0[j] = (i < Lw && i < cl) ? cp[i] : 0.0f;
	}
	float bv = 0.0f;
	int bi = 0x7fffffff;
#pragma unroll
	for (int j = 0; j < kRegs; j++)
		if (e0[j] > bv) { bv = e0[j]; bi = tid + 256 * j; }
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const float ov = __shfl_xor(bv, o);
		const int oi = __shfl_xor(bi, o);
		if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
	}
	if (lane == 0) { s_bv[wv] = bv; s_bi[wv] = bi; }
	__syncthreads();
	if (wv == 0) {
		for (int q = 1; q < 4; q++)
			if (s_bv[q] > bv || (s_bv[q] == bv && s_bi[q] < bi)) { bv = s_bv[q]; bi = s_bi[q]; }
		const int pwr_max_idx = (bi == 0x7fffffff) ? 0 : bi;
		// the twin peak one BCCH period later (fcch.c:398-430): lanes 0..20 fetch the 2 x 21 values at once (this kernel sits
		// in the receive loop's acquisition chain: its latency counts), the sums are then formed in the reference's order
		float c0 = 0.f, c1 = 0.f;
		int j0 = 0, j1 = 0;
		bool ok0 = false, ok1 = false;
		if (lane <= 20) {
			j0 = pwr_max_idx + lane - 10;
			j1 = j0 + Lp;
			ok0 = j0 > 0 && j0 < cl;
			ok1 = j1 > 0 && j1 < cl;
			c0 = ok0 ? cp[j0] : 0.f;
			c1 = ok1 ? cp[j1] : 0.f;
		}
		const float w0 = c0 * (float)j0, w1 = c1 * (float)j1;
		float pwrs0 = 0.f, pwrs1 = 0.f, pk0 = 0.f, pk1 = 0.f;
		for (int i = 0; i <= 20; i++) {
			// (a lag outside the sweep contributes nothing: the reference skips it, here it adds +0)
			if (__shfl((int)ok0, i)) { pwrs0 += __shfl(c0, i); pk0 += __shfl(w0, i); }
			if (__shfl((int)ok1, i)) { pwrs1 += __shfl(c1, i); pk1 += __shfl(w1, i); }
		}
		if (tid == 0) {
			pk0 /= pwrs0;
			pk1 /= pwrs1;
			const int nLp = (int)round((double)(pk1 - pk0));
			s_scal[0] = nLp;
			s_scal[1] = (abs(nLp - Lp) > 10) ? 1 : 0;       // also true for NaN -> INT_MIN
		}
	}
	__syncthreads();
	if (s_scal[1]) {
		if (tid == 0)
			a.count[s] = -22;
		if (tl.step == 3 && tid < kAcqPeaks)
			acq_step3(tl.g, s, tid, s_toa, -22);
		return;
	}
	Lp = s_scal[0];

	// ---- mix the two cycles, mean, standard deviation, threshold (fcch.c:435-454)
	float sum = 0.f;
	float e1[kRegs];
#pragma unroll
	for (int j = 0; j < kRegs; j++) {
		const int i = tid + 256 * j;
		e1[j] = (i < Lw && i + Lp < cl) ? cp[i + Lp] : 0.0f;
	}
#pragma unroll
	for (int j = 0; j < kRegs; j++) {
		const int i = tid + 256 * j;
		if (i < Lw) {
			// lags past the end of the sweep count as 0 (the reference over-reads there, fcch.c:438)
			// (i + Lp < cl: then i < cl, so e0 holds cp[i] -- Lp is positive here, |Lp - 7488| <= 10)
			const float m = (i + Lp < cl) ? sqrtf(e0[j] * e1[j]) : 0.0f;
			v[i] = m;
			sum += m;
			e0[j] = m;                   // (the mixed value stays in the register: deviation and threshold below)
		}
	}
	const float avg = block_sum(sum, red, tid) / (float)Lw;
	float sq = 0.f;
#pragma unroll
	for (int j = 0; j < kRegs; j++)
		if (tid + 256 * j < Lw) {
			const float d = e0[j] - avg;
			sq = fmaf(d, d, sq);
		}
	const float stddev = sqrtf(block_sum(sq, red, tid) / (float)Lw);
	const float th = avg + 3.0f * stddev;

	// ---- threshold flags, 32 lags per word: a wave's 64 lags of round j are words 8 j + 2 wv and the next
#pragma unroll
	for (int j = 0; j < kRegs; j++) {
		const int i = tid + 256 * j;
		const unsigned long long f = __ballot(i >= 1 && i < Lw - 1 && e0[j] > th);
		if (lane == 0) {
			s_flags[8 * j + 2 * wv] = (unsigned int)f;
			s_flags[8 * j + 2 * wv + 1] = (unsigned int)(f >> 32);
		}
	}
	__syncthreads();

	// ---- ordered tail: rising edges, 3-point interpolation, ranked de-duplicated list
	// (wave 0 finds the words that have a flag at all -- a handful of 234 -- by four ballots; lane 0 then walks only those,
	// in order; a word's first flag is a rising edge unless the word before it ended on a flag)
	unsigned long long nzm[4] = {0, 0, 0, 0};
	if (wv == 0) {
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const int wd = lane + 64 * c;
			nzm[c] = __ballot(wd * 32 < Lw && s_flags[wd] != 0u);
		}
	}
	if (tid == 0) {
		int *toa = s_toa;
		float *pwr = s_pwr;
		const int N = a.N;
		int n = 0;
		const int sps = a.sps;
		const int half = (a.burst_len * sps) >> 1;
		for (int c = 0; c < 4; c++)
		for (unsigned long long left = nzm[c]; left; left &= left - 1ull) {
			const int wd = 64 * c + __builtin_ctzll(left);
			const unsigned int f = s_flags[wd];
			const unsigned int before = wd > 0 ? (s_flags[wd - 1] >> 31) : 0u;
			// rising edges of the word: a flag whose lower neighbour (the word before's last, for bit 0) is not one
			for (unsigned int edges = f & ~((f << 1) | before); edges; edges &= edges - 1u) {
				const int i = wd * 32 + __builtin_ctz(edges);
				const float p_pwr = v[i - 1] + v[i] + v[i + 1];
				const float p_fpos = (-v[i - 1] + v[i + 1]) / p_pwr;
				const int p_pos = (int)round((double)(((float)i + p_fpos) * (float)sps));
				const int p_mod = p_pos % Lp;
				// _peak_record
				int has_dupe = 0;
				for (int q = 0; q < n; q++) {
					const int dd = s_mod[q] - p_mod;
					if (abs(dd) > half)
						continue;
					if (pwr[q] > p_pwr) {
						if (!has_dupe) has_dupe = 1;
						continue;
					}
					for (int j = q; j < n - 1; j++) { toa[j] = toa[j + 1]; pwr[j] = pwr[j + 1]; s_mod[j] = s_mod[j + 1]; }
					n--;
					has_dupe = -1;
				}
				if (has_dupe <= 0) {
					int q = 0;
					for (; q < n; q++)
						if (p_pwr > pwr[q]) break;
					if (q != N) {
						for (int j = N - 1; j > q; j--) { toa[j] = toa[j - 1]; pwr[j] = pwr[j - 1]; s_mod[j] = s_mod[j - 1]; }
						toa[q] = p_pos;
						pwr[q] = p_pwr;
						s_mod[q] = p_mod;
						if (n != N) n++;
					}
				}
			}
		}
		a.count[s] = n;
		s_scal[2] = n;
	}
	__syncthreads();
	if (wv != 0)
		return;
	const int n_found = s_scal[2];
	// entries the reference's list operations touched: [0, n) hold the peaks; the caller's array beyond stays as it was
	if (lane < n_found)
		a.toa[(size_t)s * a.N + lane] = s_toa[lane];
	if (tl.step == 3 && lane < kAcqPeaks) {
		// (every lane reads the carrier's status before lane 0 may write it: one wave, lockstep up to the store)
		acq_step3(tl.g, s, lane, s_toa, n_found);
	}
}

#ifdef GMR1_HIP_PROFILE
}  // namespace gmr1
extern "C" int gmr1_hip_prof_fold_dbg(float *out512)
{
	return hipDeviceSynchronize() == hipSuccess &&
	       hipMemcpyFromSymbol(out512, HIP_SYMBOL(gmr1::g_fold_dbg), sizeof(gmr1::g_fold_dbg)) == hipSuccess ? 0 : -5;
}
namespace gmr1 {
#endif
// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
int fcch_stat_tiles(int len) { return (len + kStatSpan - 1) / kStatSpan; }
int fcch_lag_tiles(int nlags) { return nlags <= 0 ? 0 : (nlags + kTileStep - 1) / kTileStep; }

template <int NT>
static void launch_corr(const FcchRoughArgs &a, hipStream_t st)
{
	const size_t lds = (size_t)(pad8(kTileLags + NT + 8) + 1) * 8 + (size_t)kTileLags * 4 + 64;
	hipLaunchKernelGGL((k_fcch_corr<NT>), dim3(a.n_lag_tiles, a.n), dim3(256), lds, st, a);
}

bool fcch_one_pass() { return profile_env("GMR1_HIP_FCCH_TWO_PASS") == nullptr; }

template <int NT>
static hipError_t launch_sweep(const FcchRoughArgs &a, const AcqTail &tl, hipStream_t st)
{
	static_assert(SweepDims<NT>::lds <= 64 * 1024, "above the default dynamic LDS limit the launch would need hipFuncSetAttribute");
	// the folded form (the tile finishes its lags itself: k_fcch_sweep<NT, true>) for LARGE launches of streams whose records fit
	// one wave's poll (small ones are latency-bound: see the kernel); the profiling build keeps the other for the comparison
	// (GMR1_HIP_FCCH_UNFOLDED)
	static const bool unfolded = profile_env("GMR1_HIP_FCCH_UNFOLDED") != nullptr;
	const int tiles = (long long)a.n_lag_tiles * a.n >= 4096 ? 4 : 1;
	// Small launches (the receive loop's acquisition: their latency is what counts): the stream's last work-group picks its
	// best tile itself, one launch fewer.  Not for the large ones: the device-scope fence in front of the count writes the
	// XCD's whole L2 back, which with thousands of work-groups in flight quadruples the kernel (0.041 -> 0.162 ms at
	// 1024 streams of 93 600 samples, measured; already at 2 048 work-groups -- 512 carriers' 330 ms windows -- it costs
	// 40 us where the launch of its own costs 7).
	const int pick = (a.toa && (long long)a.n_lag_tiles * a.n <= 512 && a.n <= kPickStreams) ? 1 : 0;
	const bool fold = !unfolded && (long long)a.n_lag_tiles * a.n > 512 && a.fold_partial && a.n_stat_tiles <= kFoldMaxTiles &&
	                  a.n_stat_tiles == a.n_lag_tiles;
	if (fold)
		hipLaunchKernelGGL((k_fcch_sweep<NT, true>), dim3(a.n_lag_tiles, a.n), dim3(256), SweepDims<NT>::lds, st, a);
	else
		hipLaunchKernelGGL((k_fcch_sweep<NT, false>), dim3(a.n_lag_tiles, a.n), dim3(256), SweepDims<NT>::lds, st, a);
	hipLaunchKernelGGL((k_fcch_energy<NT>), dim3((a.n_lag_tiles + tiles - 1) / tiles, a.n), dim3(256), 0, st, a, tiles, pick, tl,
	                   fold ? 1 : 0);
	if (a.toa && !pick)
		hipLaunchKernelGGL(k_fcch_pick, dim3(a.n), dim3(64), 0, st, a, tl);
	return hipGetLastError();
}

hipError_t launch_fcch_rough_tail(const FcchRoughArgs &a, int ntaps, const AcqTail &tl, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	if (ntaps != 117 && ntaps != 468)
		return hipErrorInvalidValue;
	if (fcch_one_pass()) {
		// (the partials are per lag tile in this form: the host sized them so, capi_fcch.cpp)
		if (a.n_stat_tiles != a.n_lag_tiles)
			return hipErrorInvalidValue;
		return ntaps == 117 ? launch_sweep<117>(a, tl, st) : launch_sweep<468>(a, tl, st);
	}
	hipLaunchKernelGGL(k_fcch_stats, dim3(a.n_stat_tiles, a.n), dim3(256), 0, st, a);
	if (ntaps == 117)
		launch_corr<117>(a, st);
	else
		launch_corr<468>(a, st);
	if (a.toa)
		hipLaunchKernelGGL(k_fcch_pick, dim3(a.n), dim3(64), 0, st, a, tl);
	return hipGetLastError();
}

hipError_t launch_fcch_rough(const FcchRoughArgs &a, int ntaps, hipStream_t st)
{
	AcqTail none;
	std::memset(&none, 0, sizeof(none));
	return launch_fcch_rough_tail(a, ntaps, none, st);
}

hipError_t launch_fcch_multi_tail(const FcchMultiArgs &a, const AcqTail &tl, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	if (a.Lw > 8192 || a.N > 32 || (tl.step == 3 && a.N != kAcqPeaks))
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_fcch_multi, dim3(a.n), dim3(256), (size_t)a.Lw * 4, st, a, tl);
	return hipGetLastError();
}

hipError_t launch_fcch_multi(const FcchMultiArgs &a, hipStream_t st)
{
	AcqTail none;
	std::memset(&none, 0, sizeof(none));
	return launch_fcch_multi_tail(a, none, st);
}

hipError_t launch_fcch_fine_tail(const FcchFineArgs &a, int nsym, const AcqTail &tl, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	if (nsym == 117)
		hipLaunchKernelGGL((k_fcch_fine<117>), dim3(a.n), dim3(64 * kFineWaves), 0, st, a, tl);
	else if (nsym == 468)
		hipLaunchKernelGGL((k_fcch_fine<468>), dim3(a.n), dim3(64 * kFineWaves), 0, st, a, tl);
	else
		return hipErrorInvalidValue;
	return hipGetLastError();
}

hipError_t launch_fcch_fine(const FcchFineArgs &a, int nsym, hipStream_t st)
{
	AcqTail none;
	std::memset(&none, 0, sizeof(none));
	return launch_fcch_fine_tail(a, nsym, none, st);
}

// ---------------------------------------------------------------------------
// k_acq_glue -- the steps above as a launch of their own between two sweeps (what the chain was before the sweeps'
// last threads took them over; kept as the other side of the comparison: profiling build, GMR1_HIP_ACQ_UNFUSED)
// ---------------------------------------------------------------------------
__global__ void k_acq_glue(int step, AcqArgs a)
{
	const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (step == 1 || step == 2) {
		if (t >= a.n)
			return;
		if (step == 1)
			acq_step1(a, t, a.toa1[t], a.rv1[t]);
		else
			acq_step2(a, t, a.ftoa[t], a.fe[t]);
		return;
	}
	if (t >= a.n * kAcqPeaks)
		return;
	if (step == 3) {
		// (one wave covers whole carriers: 64 = 4 x kAcqPeaks, so a carrier's status is read by all its slots before slot 0
		// stores it)
		const int k = t / kAcqPeaks;
		acq_step3(a, k, t % kAcqPeaks, a.peaks + (size_t)k * kAcqPeaks, a.count[k]);
		return;
	}
	acq_step4(a, t, a.ctoa[t], a.cfe[t]);
}

hipError_t launch_acq_glue(int step, const AcqArgs &a, hipStream_t st)
{
	if (a.n <= 0)
		return hipSuccess;
	const int n = (step <= 2) ? a.n : a.n * kAcqPeaks;
	hipLaunchKernelGGL(k_acq_glue, dim3((n + 255) / 256), dim3(256), 0, st, step, a);
	return hipGetLastError();
}

}  // namespace gmr1
