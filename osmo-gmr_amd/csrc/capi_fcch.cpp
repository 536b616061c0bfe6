// capi_fcch.cpp -- C-ABI entry points of the FCCH acquisition kernels.
#include "capi_common.h"
#include "profile_env.h"
#include "fcch_acq.h"

using namespace gmr1;

namespace {

int fcch_tab_of(const struct gmr1_fcch_burst *bt)
{
	for (int i = 0; i < kFcchTabs; i++)
		if (bt == kFcchBuiltin[i])
			return i;
	// a caller-provided descriptor with the parameters of a built-in one is accepted too
	for (int i = 0; i < kFcchTabs; i++)
		if (bt && bt->len == kFcchBuiltin[i]->len && bt->freq == kFcchBuiltin[i]->freq)
			return i;
	return -1;
}

const AcqTail &no_tail()
{
	static const AcqTail none = [] { AcqTail t; std::memset(&t, 0, sizeof(t)); return t; }();
	return none;
}

int rough_dev(hipStream_t st, int tab, int n, int sps, int len, const float *iq, const uint64_t *offset,
              const float *freq_shift, int32_t *toa, int32_t *rv, float *energy, size_t energy_stride,
              const AcqTail &tl = no_tail())
{
	if (tab < 0 || tab >= kFcchTabs)
		return fail(-EINVAL, "fcch: unknown burst type");
	if (n < 0 || !iq || !offset || (!toa && !energy))
		return fail(-EINVAL, "fcch_rough: NULL argument");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "fcch_rough: sps=%d out of range", sps);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n == 0) return 0;
	const int ntaps = kFcchBuiltin[tab]->len;
	const int ndec = len / sps;
	const int nlags = ndec - ntaps + 1;
	if (nlags < 5)
		return fail(-EINVAL, "fcch_rough: window of %d samples is too short", len);
	FcchRoughArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.len = len; a.sps = sps; a.tab = tab;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift;
	a.n_lag_tiles = fcch_lag_tiles(nlags);
	// the one-pass sweep keeps its statistics partials per lag tile (fcch_kernels.hip: k_fcch_sweep)
	a.n_stat_tiles = fcch_one_pass() ? a.n_lag_tiles : fcch_stat_tiles(len);
	a.dec_stride = ((size_t)ndec + 15) & ~(size_t)15;
	const size_t b_dec = (size_t)n * a.dec_stride * 8;
	const size_t b_par = (((size_t)n * a.n_stat_tiles * 16) + 255) & ~(size_t)255;
	const size_t b_best = (((size_t)n * a.n_lag_tiles * 32) + 255) & ~(size_t)255;
	void *ws;
	WsLease lease;
	if ((r = lease.acquire(s, st))) return r;
	r = dev_workspace(s, b_dec + b_par + b_best, &ws);
	if (r) return r;
	a.dec = static_cast<float2 *>(ws);
	a.partial = reinterpret_cast<float *>(static_cast<char *>(ws) + b_dec);
	a.tile_best = reinterpret_cast<float *>(static_cast<char *>(ws) + b_dec + b_par);
	a.energy = energy; a.energy_stride = energy_stride;
	a.toa = toa; a.rv = rv;
	// the folded sweep's statistics records: a buffer of its own (per device, grow-only, zeroed when made), written by these
	// kernels only, so that a record carrying this launch's epoch can only be this launch's.  Under the workspace lease: one
	// sweep at a time per device.
	{
		// 32 bytes a record slot, slot = stream * tiles + tile; the epoch words only ever hold epochs (or the zero they were made
		// with), so whatever geometry the last sweep had, a word carrying THIS launch's epoch was written by this launch
		struct FoldBuf { char *p = nullptr; size_t slots = 0; uint32_t epoch = 0; };
		static FoldBuf fold_of[64];
		int dev = 0;
		HIP_TRY(hipGetDevice(&dev));
		if (dev >= 0 && dev < 64 && a.n_stat_tiles == a.n_lag_tiles) {
			FoldBuf &fb = fold_of[dev];
			const size_t n_rec = (size_t)n * a.n_stat_tiles;
			if (fb.slots < n_rec) {
				// (a sweep of an earlier call may still be polling the old buffer)
				HIP_TRY(hipDeviceSynchronize());
				if (fb.p) HIP_TRY(hipFree(fb.p));
				fb.p = nullptr; fb.slots = 0;
				const size_t cap = n_rec + n_rec / 2;
				HIP_TRY(hipMalloc(&fb.p, cap * 32));
				HIP_TRY(hipMemset(fb.p, 0, cap * 32));
				fb.slots = cap;
				fb.epoch = 0;
			}
			if (++fb.epoch == 0) {                     // (once in four billion sweeps)
				HIP_TRY(hipDeviceSynchronize());
				HIP_TRY(hipMemset(fb.p, 0, fb.slots * 32));
				fb.epoch = 1;
			}
			a.fold_partial = reinterpret_cast<float *>(fb.p);
			a.epoch = fb.epoch;
			// (a poll is three coherent 8-byte loads and a short sleep, a microsecond or two: the bound is some tens of milliseconds.
			// Profiling build: GMR1_HIP_FCCH_FOLD_POLLS=0 makes every tile give up at once -- the fallback path, for the tests)
			static const int polls = [] { const char *e = profile_env("GMR1_HIP_FCCH_FOLD_POLLS"); return e ? atoi(e) : 1 << 14; }();
			a.fold_polls = polls;
		}
	}
	HIP_TRY(launch_fcch_rough_tail(a, ntaps, tl, st));
	return 0;
}

int fine_dev(hipStream_t st, int tab, int mode, int n, int sps, const float *iq, const uint64_t *offset,
             const float *freq_shift, int32_t *toa, float *freq_err, float *snr, const AcqTail &tl = no_tail())
{
	if (tab < 0 || tab >= kFcchTabs)
		return fail(-EINVAL, "fcch: unknown burst type");
	if (n < 0 || !iq || !offset || (mode == 0 && (!toa || !freq_err)) || (mode == 1 && !snr))
		return fail(-EINVAL, "fcch_fine/snr: NULL argument");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "fcch_fine/snr: sps=%d out of range", sps);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	FcchFineArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.tab = tab; a.mode = mode;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift;
	a.toa = toa; a.freq_err = freq_err; a.snr = snr;
	HIP_TRY(launch_fcch_fine_tail(a, kFcchBuiltin[tab]->len, tl, st));
	return 0;
}

// shared host staging: copies iq + offsets (+ freq_shift) in, runs `body`, copies results out
struct Staged {
	DBuf iq, off, fs;
	int stage(int n, const float *h_iq, uint64_t iq_len, const uint64_t *h_off, const float *h_fs,
	          uint64_t need)
	{
		for (int i = 0; i < n; i++)
			if (h_off[i] + need > iq_len)
				return fail(-EINVAL, "window %d runs past the end of iq", i);
		HIP_TRY(iq.alloc(iq_len * 8));
		HIP_TRY(off.alloc((size_t)n * 8));
		HIP_TRY(hipMemcpy(iq.p, h_iq, iq_len * 8, hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(off.p, h_off, (size_t)n * 8, hipMemcpyHostToDevice));
		if (h_fs) {
			HIP_TRY(fs.alloc((size_t)n * 4));
			HIP_TRY(hipMemcpy(fs.p, h_fs, (size_t)n * 4, hipMemcpyHostToDevice));
		}
		return 0;
	}
};

}  // namespace

extern "C" {

int gmr1_hip_fcch_rough_batch_dev(void *stream, int fcch_type, int n, int sps, int len,
                                  const float *iq, const uint64_t *offset, const float *freq_shift,
                                  int32_t *toa, int32_t *rv)
{
	return rough_dev((hipStream_t)stream, fcch_type, n, sps, len, iq, offset, freq_shift, toa, rv, nullptr, 0);
}

int gmr1_hip_fcch_rough_batch(int fcch_type, int n, int sps, int len,
                              const float *iq, uint64_t iq_len, const uint64_t *offset,
                              const float *freq_shift, int32_t *toa, int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !toa)
		return fail(-EINVAL, "fcch_rough: NULL argument");
	Staged st;
	r = st.stage(n, iq, iq_len, offset, freq_shift, (uint64_t)len);
	if (r) return r;
	DBuf d_toa, d_rv;
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	r = rough_dev(nullptr, fcch_type, n, sps, len, st.iq.as<float>(), st.off.as<uint64_t>(),
	              freq_shift ? st.fs.as<float>() : nullptr, d_toa.as<int32_t>(), d_rv.as<int32_t>(), nullptr, 0);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (rv) HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_fcch_fine_batch_dev(void *stream, int fcch_type, int n, int sps,
                                 const float *iq, const uint64_t *offset, const float *freq_shift,
                                 int32_t *toa, float *freq_error)
{
	return fine_dev((hipStream_t)stream, fcch_type, 0, n, sps, iq, offset, freq_shift, toa, freq_error, nullptr);
}

int gmr1_hip_fcch_snr_batch_dev(void *stream, int fcch_type, int n, int sps,
                                const float *iq, const uint64_t *offset, const float *freq_shift,
                                float *snr)
{
	return fine_dev((hipStream_t)stream, fcch_type, 1, n, sps, iq, offset, freq_shift, nullptr, nullptr, snr);
}

int gmr1_hip_fcch_fine_batch(int fcch_type, int n, int sps,
                             const float *iq, uint64_t iq_len, const uint64_t *offset,
                             const float *freq_shift, int32_t *toa, float *freq_error)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (fcch_type < 0 || fcch_type >= kFcchTabs || !iq || !offset || !toa || !freq_error)
		return fail(-EINVAL, "fcch_fine: bad argument");
	Staged st;
	r = st.stage(n, iq, iq_len, offset, freq_shift, (uint64_t)kFcchBuiltin[fcch_type]->len * sps);
	if (r) return r;
	DBuf d_toa, d_fe;
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_fe.alloc((size_t)n * 4));
	r = fine_dev(nullptr, fcch_type, 0, n, sps, st.iq.as<float>(), st.off.as<uint64_t>(),
	             freq_shift ? st.fs.as<float>() : nullptr, d_toa.as<int32_t>(), d_fe.as<float>(), nullptr);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(freq_error, d_fe.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_fcch_snr_batch(int fcch_type, int n, int sps,
                            const float *iq, uint64_t iq_len, const uint64_t *offset,
                            const float *freq_shift, float *snr)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (fcch_type < 0 || fcch_type >= kFcchTabs || !iq || !offset || !snr)
		return fail(-EINVAL, "fcch_snr: bad argument");
	Staged st;
	r = st.stage(n, iq, iq_len, offset, freq_shift, (uint64_t)kFcchBuiltin[fcch_type]->len * sps);
	if (r) return r;
	DBuf d_snr;
	HIP_TRY(d_snr.alloc((size_t)n * 4));
	r = fine_dev(nullptr, fcch_type, 1, n, sps, st.iq.as<float>(), st.off.as<uint64_t>(),
	             freq_shift ? st.fs.as<float>() : nullptr, nullptr, nullptr, d_snr.as<float>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(snr, d_snr.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

}  // extern "C"

namespace gmr1 {

int fcch_rough_tail(hipStream_t st, int tab, int n, int sps, int len, const float *iq, const uint64_t *offset,
                    const float *freq_shift, int32_t *toa, int32_t *rv, const AcqTail &t)
{
	return rough_dev(st, tab, n, sps, len, iq, offset, freq_shift, toa, rv, nullptr, 0, t);
}

int fcch_fine_tail(hipStream_t st, int tab, int mode, int n, int sps, const float *iq, const uint64_t *offset,
                   const float *freq_shift, int32_t *toa, float *freq_err, float *snr, const AcqTail &t)
{
	return fine_dev(st, tab, mode, n, sps, iq, offset, freq_shift, toa, freq_err, snr, t);
}

int fcch_rough_multi_tail(hipStream_t stream, int fcch_type, int n, int sps, int len,
                          const float *iq, const uint64_t *offset, const float *freq_shift,
                          int32_t *peaks_toa, int N, int32_t *count, const AcqTail &tl)
{
	if (fcch_type < 0 || fcch_type >= kFcchTabs || !peaks_toa || !count || N < 1 || N > 32)
		return fail(-EINVAL, "fcch_rough_multi: bad argument");
	if (sps < 1 || sps > 16)                                      // before any arithmetic that divides by it
		return fail(-EINVAL, "fcch_rough_multi: sps=%d out of range (1..16)", sps);
	if (len < ((650 * 23400 * sps) / 1000))                       // fcch.c:355-356
		return fail(-EINVAL, "fcch_rough_multi: needs at least 650 ms of signal");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	const int blen = kFcchBuiltin[fcch_type]->len;
	const int nlags = len / sps - blen + 1;
	const size_t estride = ((size_t)nlags + 63) & ~(size_t)63;
	// the energy plane lives behind the rough sweep's own scratch: ask for both at once
	const int ndec = len / sps;
	const size_t b_dec = (size_t)n * ((((size_t)ndec + 15) & ~(size_t)15) * 8);
	const size_t b_par = (((size_t)n * fcch_stat_tiles(len) * 16) + 255) & ~(size_t)255;
	const size_t b_best = (((size_t)n * fcch_lag_tiles(nlags) * 32) + 255) & ~(size_t)255;
	const size_t b_rough = b_dec + b_par + b_best;
	void *ws;
	WsLease lease;
	if ((r = lease.acquire(s, (hipStream_t)stream))) return r;
	r = dev_workspace(s, b_rough + (size_t)n * estride * 4, &ws);
	if (r) return r;
	float *energy = reinterpret_cast<float *>(static_cast<char *>(ws) + b_rough);
	r = rough_dev((hipStream_t)stream, fcch_type, n, sps, len, iq, offset, freq_shift, nullptr, nullptr,
	              energy, estride);
	if (r) return r;
	FcchMultiArgs m;
	std::memset(&m, 0, sizeof(m));
	m.n = n; m.sps = sps; m.burst_len = blen; m.N = N;
	m.nlags = nlags;
	m.Lp = (320 * 23400) / 1000;                                  // fcch.c:380-383
	m.Lw = m.Lp + blen;
	m.energy = energy; m.energy_stride = estride;
	m.toa = peaks_toa; m.count = count;
	HIP_TRY(launch_fcch_multi_tail(m, tl, stream));
	return 0;
}

}  // namespace gmr1

extern "C" {

int gmr1_hip_fcch_rough_multi_batch_dev(void *stream, int fcch_type, int n, int sps, int len,
                                        const float *iq, const uint64_t *offset, const float *freq_shift,
                                        int32_t *peaks_toa, int N, int32_t *count)
{
	return fcch_rough_multi_tail((hipStream_t)stream, fcch_type, n, sps, len, iq, offset, freq_shift, peaks_toa, N, count,
	                             no_tail());
}

int gmr1_hip_fcch_rough_multi_batch(int fcch_type, int n, int sps, int len,
                                    const float *iq, uint64_t iq_len, const uint64_t *offset,
                                    const float *freq_shift, int32_t *peaks_toa, int N, int32_t *count)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !peaks_toa || !count || N < 1 || N > 32)
		return fail(-EINVAL, "fcch_rough_multi: bad argument");
	if (len < ((650 * 23400 * sps) / 1000))
		return fail(-EINVAL, "fcch_rough_multi: needs at least 650 ms of signal");
	Staged st;
	r = st.stage(n, iq, iq_len, offset, freq_shift, (uint64_t)len);
	if (r) return r;
	DBuf d_toa, d_cnt;
	HIP_TRY(d_toa.alloc((size_t)n * N * 4));
	HIP_TRY(d_cnt.alloc((size_t)n * 4));
	HIP_TRY(hipMemset(d_toa.p, 0, (size_t)n * N * 4));
	r = gmr1_hip_fcch_rough_multi_batch_dev(nullptr, fcch_type, n, sps, len, st.iq.as<float>(), st.off.as<uint64_t>(),
	                                        freq_shift ? st.fs.as<float>() : nullptr, d_toa.as<int32_t>(), N,
	                                        d_cnt.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(peaks_toa, d_toa.p, (size_t)n * N * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(count, d_cnt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

// ---- reference-compatible single calls (fcch.h:47-61) ---------------------------------------
int gmr1_fcch_rough(const struct gmr1_fcch_burst *burst_type,
                    struct osmo_cxvec *search_win_in, int sps, float freq_shift, int *toa)
{
	if (!burst_type || !search_win_in || !search_win_in->data || !toa)
		return fail(-EINVAL, "gmr1_fcch_rough: NULL argument");
	const int tab = fcch_tab_of(burst_type);
	if (tab < 0)
		return fail(-EINVAL, "gmr1_fcch_rough: unknown FCCH burst type");
	const uint64_t off = 0;
	int32_t t = 0, rv = 0;
	int r = gmr1_hip_fcch_rough_batch(tab, 1, sps, search_win_in->len,
	                                  reinterpret_cast<const float *>(search_win_in->data),
	                                  (uint64_t)search_win_in->len, &off, &freq_shift, &t, &rv);
	if (r) return r;
	if (rv) return rv;
	*toa = t;
	return 0;
}

int gmr1_fcch_rough_multi(const struct gmr1_fcch_burst *burst_type,
                          struct osmo_cxvec *search_win_in, int sps, float freq_shift, int *peaks_toa, int N)
{
	if (!burst_type || !search_win_in || !search_win_in->data || !peaks_toa)
		return fail(-EINVAL, "gmr1_fcch_rough_multi: NULL argument");
	const int tab = fcch_tab_of(burst_type);
	if (tab < 0)
		return fail(-EINVAL, "gmr1_fcch_rough_multi: unknown FCCH burst type");
	if (N < 1 || N > 32)
		return fail(-EINVAL, "gmr1_fcch_rough_multi: N must be 1..32");
	const uint64_t off = 0;
	int32_t cnt = 0;
	int32_t tmp[32] = {0};
	int r = gmr1_hip_fcch_rough_multi_batch(tab, 1, sps, search_win_in->len,
	                                        reinterpret_cast<const float *>(search_win_in->data),
	                                        (uint64_t)search_win_in->len, &off, &freq_shift, tmp, N, &cnt);
	if (r) return r;
	for (int i = 0; i < N && i < cnt; i++)
		peaks_toa[i] = tmp[i];
	return cnt;
}

int gmr1_fcch_fine(const struct gmr1_fcch_burst *burst_type,
                   struct osmo_cxvec *burst_in, int sps, float freq_shift, int *toa, float *freq_error)
{
	if (!burst_type || !burst_in || !burst_in->data || !toa || !freq_error)
		return fail(-EINVAL, "gmr1_fcch_fine: NULL argument");
	const int tab = fcch_tab_of(burst_type);
	if (tab < 0)
		return fail(-EINVAL, "gmr1_fcch_fine: unknown FCCH burst type");
	if (sps < 1 || burst_in->len / sps != burst_type->len)      // fcch.c:546-551
		return fail(-EINVAL, "gmr1_fcch_fine: burst must be len*sps samples");
	const uint64_t off = 0;
	int32_t t = 0;
	float fe = 0.f;
	int r = gmr1_hip_fcch_fine_batch(tab, 1, sps, reinterpret_cast<const float *>(burst_in->data),
	                                 (uint64_t)burst_in->len, &off, &freq_shift, &t, &fe);
	if (r) return r;
	*toa = t;
	*freq_error = fe;
	return 0;
}

int gmr1_fcch_snr(const struct gmr1_fcch_burst *burst_type,
                  struct osmo_cxvec *burst_in, int sps, float freq_shift, float *snr)
{
	if (!burst_type || !burst_in || !burst_in->data || !snr)
		return fail(-EINVAL, "gmr1_fcch_snr: NULL argument");
	const int tab = fcch_tab_of(burst_type);
	if (tab < 0)
		return fail(-EINVAL, "gmr1_fcch_snr: unknown FCCH burst type");
	if (sps < 1 || burst_in->len / sps != burst_type->len)      // fcch.c:671-675
		return fail(-EINVAL, "gmr1_fcch_snr: burst must be len*sps samples");
	const uint64_t off = 0;
	float v = 0.f;
	int r = gmr1_hip_fcch_snr_batch(tab, 1, sps, reinterpret_cast<const float *>(burst_in->data),
	                                (uint64_t)burst_in->len, &off, &freq_shift, &v);
	if (r) return r;
	*snr = v;
	return 0;
}

}  // extern "C"
