// capi_tch.cpp -- C ABI of the DKAB demodulator and the A5 keystream generator
// (reference include/osmocom/gmr1/sdr/dkab.h:39-41, include/osmocom/gmr1/l1/a5.h:37-41).

#include "capi_common.h"

#include <vector>

#include "../../include/gmr1_hip.h"
#include "../../include/osmocom/gmr1/sdr/dkab.h"
#include "../../include/osmocom/gmr1/l1/a5.h"

using namespace gmr1;

namespace {
constexpr int kDkabMaxLen = 4096;
}

extern "C" {

int gmr1_hip_dkab_demod_batch_dev(void *stream, int n, int sps, int in_len,
                                  const float *iq, const uint64_t *offset, const float *freq_shift,
                                  const int32_t *p, int8_t *ebits, float *toa, int32_t *rv)
{
	if (n < 0 || !iq || !offset || !p || !rv)
		return fail(-EINVAL, "dkab: iq/offset/p/rv are required");
	if (sps < 1 || sps > 16)
		return fail(-EINVAL, "dkab: sps=%d out of range (1..16)", sps);
	const int w = in_len - GMR1_DKAB_SYMS * sps + 1;
	if (w <= 0)
		return fail(-EINVAL, "dkab: window of %d samples is shorter than a DKAB (%d)", in_len, GMR1_DKAB_SYMS * sps);
	if (in_len > kDkabMaxLen)
		return fail(-EINVAL, "dkab: window of %d samples (<= %d supported)", in_len, kDkabMaxLen);
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	DkabArgs a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.sps = sps; a.in_len = in_len;
	a.iq = reinterpret_cast<const float2 *>(iq);
	a.offset = offset; a.freq_shift = freq_shift; a.p = p;
	a.ebits = ebits; a.toa = toa; a.rv = rv;
	HIP_TRY(launch_dkab(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_dkab_demod_batch(int n, int sps, int in_len,
                              const float *iq, uint64_t iq_len, const uint64_t *offset, const float *freq_shift,
                              const int32_t *p, int8_t *ebits, float *toa, int32_t *rv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!iq || !offset || !p || !rv)
		return fail(-EINVAL, "dkab: iq/offset/p/rv are required");
	for (int i = 0; i < n; i++)
		if (offset[i] + (uint64_t)in_len > iq_len)
			return fail(-EINVAL, "burst %d runs past the end of iq", i);
	DBuf d_iq, d_off, d_fs, d_p, d_eb, d_toa, d_rv;
	HIP_TRY(d_iq.alloc(iq_len * 8));
	HIP_TRY(d_off.alloc((size_t)n * 8));
	HIP_TRY(d_p.alloc((size_t)n * 4));
	HIP_TRY(d_eb.alloc((size_t)n * 8));
	HIP_TRY(d_toa.alloc((size_t)n * 4));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	if (freq_shift) HIP_TRY(d_fs.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_iq.p, iq, iq_len * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off.p, offset, (size_t)n * 8, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_p.p, p, (size_t)n * 4, hipMemcpyHostToDevice));
	if (freq_shift) HIP_TRY(hipMemcpy(d_fs.p, freq_shift, (size_t)n * 4, hipMemcpyHostToDevice));
	r = gmr1_hip_dkab_demod_batch_dev(nullptr, n, sps, in_len, d_iq.as<float>(), d_off.as<uint64_t>(),
	                                  freq_shift ? d_fs.as<float>() : nullptr, d_p.as<int32_t>(),
	                                  d_eb.as<int8_t>(), d_toa.as<float>(), d_rv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (ebits) HIP_TRY(hipMemcpy(ebits, d_eb.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	if (toa) HIP_TRY(hipMemcpy(toa, d_toa.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_dkab_demod(struct osmo_cxvec *burst_in, int sps, float freq_shift, int p, sbit_t *ebits, float *toa_p)
{
	if (!burst_in || !burst_in->data || !ebits || !toa_p)
		return fail(-EINVAL, "gmr1_dkab_demod: NULL argument");
	const uint64_t off = 0;
	int32_t rv = 0, pp = p;
	float toa = 0.f;
	int8_t eb[8];
	int r = gmr1_hip_dkab_demod_batch(1, sps, burst_in->len, reinterpret_cast<const float *>(burst_in->data),
	                                  (uint64_t)burst_in->len, &off, &freq_shift, &pp, eb, &toa, &rv);
	if (r) return r;
	*toa_p = toa;
	if (rv == 0)
		std::memcpy(ebits, eb, 8);
	return rv;
}

int gmr1_hip_a5_batch_dev(void *stream, int n, int alg, int nbits,
                          const uint8_t *keys, const uint32_t *fn, uint8_t *dl, uint8_t *ul)
{
	if (n < 0 || nbits < 0 || (alg == 1 && (!keys || !fn)))
		return fail(-EINVAL, "a5: keys / fn are required for A5/1");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	A5Args a;
	std::memset(&a, 0, sizeof(a));
	a.n = n; a.alg = alg; a.nbits = nbits; a.keys = keys; a.fn = fn; a.dl = dl; a.ul = ul;
	HIP_TRY(launch_a5(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_a5_batch(int n, int alg, int nbits, const uint8_t *keys, const uint32_t *fn, uint8_t *dl, uint8_t *ul)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0 || nbits <= 0) return 0;
	if (alg == 1 && (!keys || !fn))
		return fail(-EINVAL, "a5: keys / fn are required for A5/1");
	DBuf d_k, d_fn, d_dl, d_ul;
	const size_t nb = (size_t)n * nbits;
	if (keys) { HIP_TRY(d_k.alloc((size_t)n * 8)); HIP_TRY(hipMemcpy(d_k.p, keys, (size_t)n * 8, hipMemcpyHostToDevice)); }
	if (fn) { HIP_TRY(d_fn.alloc((size_t)n * 4)); HIP_TRY(hipMemcpy(d_fn.p, fn, (size_t)n * 4, hipMemcpyHostToDevice)); }
	// the buffers start out as the caller's, so that "unsupported n" leaves them untouched
	if (dl) { HIP_TRY(d_dl.alloc(nb)); HIP_TRY(hipMemcpy(d_dl.p, dl, nb, hipMemcpyHostToDevice)); }
	if (ul) { HIP_TRY(d_ul.alloc(nb)); HIP_TRY(hipMemcpy(d_ul.p, ul, nb, hipMemcpyHostToDevice)); }
	r = gmr1_hip_a5_batch_dev(nullptr, n, alg, nbits, keys ? d_k.as<uint8_t>() : nullptr, fn ? d_fn.as<uint32_t>() : nullptr,
	                          dl ? d_dl.as<uint8_t>() : nullptr, ul ? d_ul.as<uint8_t>() : nullptr);
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	if (dl) HIP_TRY(hipMemcpy(dl, d_dl.p, nb, hipMemcpyDeviceToHost));
	if (ul) HIP_TRY(hipMemcpy(ul, d_ul.p, nb, hipMemcpyDeviceToHost));
	return 0;
}

void gmr1_a5(int n, uint8_t *key, uint32_t fn, int nbits, ubit_t *dl, ubit_t *ul)
{
	(void)gmr1_hip_a5_batch(1, n, nbits, key, &fn, reinterpret_cast<uint8_t *>(dl), reinterpret_cast<uint8_t *>(ul));
}

void gmr1_a5_1(uint8_t *key, uint32_t fn, int nbits, ubit_t *dl, ubit_t *ul)
{
	gmr1_a5(1, key, fn, nbits, dl, ul);
}

}  // extern "C"
