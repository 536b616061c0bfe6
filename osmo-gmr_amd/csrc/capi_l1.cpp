// capi_l1.cpp -- C-ABI entry points of the traffic-channel layer-1 decoders (FACCH3, TCH3).
#include "capi_common.h"

#include <osmocom/gmr1/l1/facch3.h>
#include <osmocom/gmr1/l1/tch3.h>

using namespace gmr1;

extern "C" {

int gmr1_hip_facch3_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *ciph,
                                     uint8_t *l2, uint8_t *bits_s, int32_t *crc, int32_t *conv)
{
	if (n < 0 || !ebits || !l2 || !crc || !conv)
		return fail(-EINVAL, "facch3 decode: NULL argument");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	Facch3Args a;
	a.n = n; a.conv_acc = conv_acc(); a.ebits = ebits; a.ciph = ciph; a.l2 = l2; a.bits_s = bits_s; a.crc = crc; a.conv = conv;
	HIP_TRY(launch_facch3(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_facch3_decode_batch(int n, const int8_t *ebits, const uint8_t *ciph,
                                 uint8_t *l2, uint8_t *bits_s, int32_t *crc, int32_t *conv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!ebits || !l2 || !crc || !conv)
		return fail(-EINVAL, "facch3 decode: NULL argument");
	DBuf d_eb, d_ci, d_l2, d_s, d_crc, d_conv;
	HIP_TRY(d_eb.alloc((size_t)n * 416));
	HIP_TRY(d_l2.alloc((size_t)n * 10));
	HIP_TRY(d_s.alloc((size_t)n * 32));
	HIP_TRY(d_crc.alloc((size_t)n * 4));
	HIP_TRY(d_conv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_eb.p, ebits, (size_t)n * 416, hipMemcpyHostToDevice));
	if (ciph) {
		HIP_TRY(d_ci.alloc((size_t)n * 384));
		HIP_TRY(hipMemcpy(d_ci.p, ciph, (size_t)n * 384, hipMemcpyHostToDevice));
	}
	r = gmr1_hip_facch3_decode_batch_dev(nullptr, n, d_eb.as<int8_t>(), ciph ? d_ci.as<uint8_t>() : nullptr,
	                                     d_l2.as<uint8_t>(), d_s.as<uint8_t>(), d_crc.as<int32_t>(), d_conv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(l2, d_l2.p, (size_t)n * 10, hipMemcpyDeviceToHost));
	if (bits_s) HIP_TRY(hipMemcpy(bits_s, d_s.p, (size_t)n * 32, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(conv, d_conv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int gmr1_hip_tch3_decode_batch_dev(void *stream, int n, int m, const int8_t *ebits, const uint8_t *ciph,
                                   uint8_t *frames, uint8_t *bits_s, int32_t *conv)
{
	if (n < 0 || !ebits || !frames)
		return fail(-EINVAL, "tch3 decode: NULL argument");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	Tch3Args a;
	a.n = n; a.m = m ? 1 : 0; a.conv_acc = conv_acc(); a.ebits = ebits; a.ciph = ciph; a.frames = frames; a.bits_s = bits_s; a.conv = conv;
	HIP_TRY(launch_tch3(a, (hipStream_t)stream));
	return 0;
}

int gmr1_hip_tch3_decode_batch(int n, int m, const int8_t *ebits, const uint8_t *ciph,
                               uint8_t *frames, uint8_t *bits_s, int32_t *conv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return 0;
	if (!ebits || !frames)
		return fail(-EINVAL, "tch3 decode: NULL argument");
	DBuf d_eb, d_ci, d_fr, d_s, d_conv;
	HIP_TRY(d_eb.alloc((size_t)n * 212));
	HIP_TRY(d_fr.alloc((size_t)n * 20));
	HIP_TRY(d_s.alloc((size_t)n * 4));
	HIP_TRY(d_conv.alloc((size_t)n * 8));
	HIP_TRY(hipMemcpy(d_eb.p, ebits, (size_t)n * 212, hipMemcpyHostToDevice));
	if (ciph) {
		HIP_TRY(d_ci.alloc((size_t)n * 208));
		HIP_TRY(hipMemcpy(d_ci.p, ciph, (size_t)n * 208, hipMemcpyHostToDevice));
	}
	r = gmr1_hip_tch3_decode_batch_dev(nullptr, n, m, d_eb.as<int8_t>(), ciph ? d_ci.as<uint8_t>() : nullptr,
	                                   d_fr.as<uint8_t>(), d_s.as<uint8_t>(), d_conv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(frames, d_fr.p, (size_t)n * 20, hipMemcpyDeviceToHost));
	if (bits_s) HIP_TRY(hipMemcpy(bits_s, d_s.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (conv) HIP_TRY(hipMemcpy(conv, d_conv.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	return 0;
}

// ---- reference-compatible single calls ---------------------------------------------------------
int gmr1_facch3_decode(uint8_t *l2, ubit_t *bits_s, const sbit_t *bits_e, const ubit_t *ciph, int *conv_rv)
{
	int32_t crc = 0, conv = 0;
	int r = gmr1_hip_facch3_decode_batch(1, reinterpret_cast<const int8_t *>(bits_e), ciph, l2, bits_s, &crc, &conv);
	if (r) return r;
	if (conv_rv) *conv_rv = conv;
	return crc;
}

void gmr1_tch3_decode(uint8_t *frame0, uint8_t *frame1, ubit_t *bits_s,
                      const sbit_t *bits_e, const ubit_t *ciph, int m, int *conv0_rv, int *conv1_rv)
{
	// the reference returns void: a device failure leaves the outputs zeroed and is
	// reported through gmr1_hip_last_error()
	uint8_t fr[20] = {0};
	uint8_t st[4] = {0};
	int32_t conv[2] = {0, 0};
	(void)gmr1_hip_tch3_decode_batch(1, m, reinterpret_cast<const int8_t *>(bits_e), ciph, fr, st, conv);
	std::memcpy(frame0, fr, 10);
	std::memcpy(frame1, fr + 10, 10);
	if (bits_s) std::memcpy(bits_s, st, 4);
	if (conv0_rv) *conv0_rv = conv[0];
	if (conv1_rv) *conv1_rv = conv[1];
}

}  // extern "C"
