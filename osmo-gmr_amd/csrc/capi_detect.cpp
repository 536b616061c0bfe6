// capi_detect.cpp -- C-ABI entry points of burst-type detection and modulation-order estimation.
#include "capi_common.h"

using namespace gmr1;

extern "C" {

int gmr1_hip_detect_batch_dev(void *stream, int n_types, const int *burst_ids, int n, int sps, int in_len,
                              const float *iq, const uint64_t *offset, const float *freq_shift,
                              const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv)
{
	if (n < 0 || !burst_ids || !iq || !offset || !rv)
		return fail(-EINVAL, "detect: NULL argument");
	if (n_types < 1 || n_types > 4)
		return fail(-EINVAL, "detect: 1..4 candidate burst types");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "detect: sps=%d out of range (1..16)", sps);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	DetectArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len = in_len; a.n_types = n_types;
	for (int i = 0; i < n_types; i++) {
		if (burst_ids[i] < 0 || burst_ids[i] >= GMR1_HIP_N_BURSTS)
			return fail(-EINVAL, "detect: bad burst id %d", burst_ids[i]);
		a.types[i] = burst_ids[i];
		const DevBurst &ht = g_host_types[burst_ids[i]];
		const int w = in_len - ht.len * sps + 1;
		if (w < 1 || w > kMaxWindow || in_len > kMaxInLen)
			return fail(-EINVAL, "detect: window of %d samples gives %d lags (1..%d supported)", in_len, w, kMaxWindow);
	}
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift; a.e_toa = e_toa;
	a.bt_id = bt_id; a.sync_id = sync_id; a.toa = toa; a.rv = rv;
	HIP_TRY(launch_detect(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_detect_batch(int n_types, const int *burst_ids, int n, int sps, int in_len,
                          const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                          const float *e_toa, int32_t *bt_id, int32_t *sync_id, float *toa, int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !rv)
		return fail(-EINVAL, "detect: NULL argument");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_et, d_bt, d_sid, d_toa, d_rv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_bt.alloc((size_t)n * 4));
	HIP_TRY(d_sid.alloc((size_t)n * 4));
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	if (e_toa) {
		HIP_TRY(d_et.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_et.p, e_toa, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	r = gmr1_hip_detect_batch_dev(nullptr, n_types, burst_ids, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                              freq_shift ? d_fs.as<float>() : nullptr, e_toa ? d_et.as<float>() : nullptr,
	                              d_bt.as<int32_t>(), d_sid.as<int32_t>(), d_toa.as<float>(), d_rv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (bt_id) HIP_TRY(hipMemcpy(bt_id, d_bt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (sync_id) HIP_TRY(hipMemcpy(sync_id, d_sid.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_mod_order_batch_dev(void *stream, int n, int sps, int in_len,
                                 const float *iq, const uint64_t *offset, const float *freq_shift, int32_t *order)
{
	if (n < 0 || !iq || !offset || !order)
		return fail(-EINVAL, "mod_order: NULL argument");
	if (sps < 1 || sps > 16 || in_len < 1 || in_len > kMaxInLen)
		return fail(-EINVAL, "mod_order: sps / window out of range");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	ModOrderArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len = in_len;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift; a.order = order;
	HIP_TRY(launch_mod_order(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_mod_order_batch(int n, int sps, int in_len, const float *iq, uint64_t iq_len,
                             const uint64_t *offset, const float *freq_shift, int32_t *order)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !order)
		return fail(-EINVAL, "mod_order: NULL argument");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_o;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_o.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	if (freq_shift) {
		HIP_TRY(d_fs.alloc((size_t)n * 4));
		HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	r = gmr1_hip_mod_order_batch_dev(nullptr, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                                 freq_shift ? d_fs.as<float>() : nullptr, d_o.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(order, d_o.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

// ---- reference-compatible single calls (pi4cxpsk.h:107-113) -------------------------------
int gmr1_pi4cxpsk_detect(struct gmr1_pi4cxpsk_burst **burst_types, float e_toa,
                         struct osmo_cxvec *burst_in, int sps, float freq_shift,
                         int *bt_id_p, int *sync_id_p, float *toa_p)
{
	if (!burst_types || !burst_types[0] || !burst_in || !burst_in->data)
		return fail(-EINVAL, "gmr1_pi4cxpsk_detect: NULL argument");
	int r = host_types();
	if (r) return r;
	int ids[4], nt = 0;
	for (; burst_types[nt]; nt++) {
		if (nt >= 4)
			return fail(-EINVAL, "gmr1_pi4cxpsk_detect: at most 4 candidate types");
		int id = -1;
		for (int i = 0; i < GMR1_HIP_N_BURSTS; i++)
			if (burst_types[nt] == kBuiltin[i])
				id = i;
		if (id < 0)
			return fail(-EINVAL, "gmr1_pi4cxpsk_detect: only the built-in burst types are supported");
		ids[nt] = id;
	}
	const uint64_t off = 0;
	int32_t bt = -1, sid = -1, rv = 0;
	float toa = 0.f;
	r = gmr1_hip_detect_batch(nt, ids, 1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                          (uint64_t)burst_in->len, &off, &freq_shift, &e_toa, &bt, &sid, &toa, &rv);
	if (r) return r;
	if (rv) return rv;
	if (bt_id_p) *bt_id_p = bt;
	if (sync_id_p) *sync_id_p = sid;
	if (toa_p) *toa_p = toa;
	return 0;
}

int gmr1_pi4cxpsk_mod_order(struct osmo_cxvec *burst_in, int sps, float freq_shift)
{
	if (!burst_in || !burst_in->data)
		return fail(-EINVAL, "gmr1_pi4cxpsk_mod_order: NULL argument");
	const uint64_t off = 0;
	int32_t order = 0;
	int r = gmr1_hip_mod_order_batch(1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                                 (uint64_t)burst_in->len, &off, &freq_shift, &order);
	if (r) return r;
	return order;
}

}  // extern "C"
