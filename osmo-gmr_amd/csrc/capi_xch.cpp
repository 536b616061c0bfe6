// capi_xch.cpp -- C ABI of the two layer-1 decoders gmr1_rx itself never calls: xCH over DC12 (reference
// include/osmocom/gmr1/l1/xch_dc12.h:37-38) and RACH (include/osmocom/gmr1/l1/rach.h:37-39).  Everything
// per burst runs on the GPU (xch_kernels.hip, nt9_kernels.hip); there is no CPU path.

#include "capi_common.h"

#include "../../include/gmr1_hip.h"
#include "../../include/osmocom/gmr1/l1/rach.h"
#include "../../include/osmocom/gmr1/l1/xch_dc12.h"

using namespace gmr1;

namespace {

int xch_dev(hipStream_t st, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	if (n < 0 || (n > 0 && (!ebits || !l2 || !crc)))
		return fail(-EINVAL, "xch_dc12: ebits / l2 / crc are required");
	if ((reinterpret_cast<uintptr_t>(ebits) & 3u) || (reinterpret_cast<uintptr_t>(l2) & 1u))
		return fail(-EINVAL, "xch_dc12: ebits must be 4-byte and l2 2-byte aligned");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	XchArgs a;
	a.n = n; a.ebits = ebits; a.l2 = l2; a.crc = crc; a.conv = conv;
	HIP_TRY(launch_xch(a, st));
	return 0;
}

int rach_dev(hipStream_t st, int n, const int8_t *ebits, const uint8_t *sb_mask, uint8_t *rach,
             int32_t *rv, int32_t *conv, int32_t *crc)
{
	if (n < 0 || (n > 0 && (!ebits || !sb_mask || !rach || !rv)))
		return fail(-EINVAL, "rach: ebits / sb_mask / rach / rv are required");
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	RachArgs a;
	a.n = n; a.conv_acc = conv_acc(); a.ebits = ebits; a.sb_mask = sb_mask; a.rach = rach; a.rv = rv; a.conv = conv; a.crc = crc;
	HIP_TRY(launch_rach(a, st));
	return 0;
}

int xch_host(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return n < 0 ? fail(-EINVAL, "xch_dc12: n < 0") : 0;
	if (!ebits || !l2 || !crc)
		return fail(-EINVAL, "xch_dc12: ebits / l2 / crc are required");
	DBuf d_e, d_l2, d_crc, d_cv;
	HIP_TRY(d_e.alloc((size_t)n * 432));
	HIP_TRY(d_l2.alloc((size_t)n * 24));
	HIP_TRY(d_crc.alloc((size_t)n * 4));
	HIP_TRY(d_cv.alloc((size_t)n * 4));
	HIP_TRY(hipMemcpy(d_e.p, ebits, (size_t)n * 432, hipMemcpyHostToDevice));
	r = xch_dev(nullptr, n, d_e.as<int8_t>(), d_l2.as<uint8_t>(), d_crc.as<int32_t>(), d_cv.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(l2, d_l2.p, (size_t)n * 24, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (conv) HIP_TRY(hipMemcpy(conv, d_cv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	return 0;
}

int rach_host(int n, const int8_t *ebits, const uint8_t *sb_mask, uint8_t *rach, int32_t *rv, int32_t *conv,
              int32_t *crc)
{
	DevState *s;
	int r = dev_state(&s);
	if (r) return r;
	if (n <= 0) return n < 0 ? fail(-EINVAL, "rach: n < 0") : 0;
	if (!ebits || !sb_mask || !rach || !rv)
		return fail(-EINVAL, "rach: ebits / sb_mask / rach / rv are required");
	DBuf d_e, d_m, d_r, d_rv, d_cv, d_crc;
	HIP_TRY(d_e.alloc((size_t)n * 494));
	HIP_TRY(d_m.alloc((size_t)n));
	HIP_TRY(d_r.alloc((size_t)n * 18));
	HIP_TRY(d_rv.alloc((size_t)n * 4));
	HIP_TRY(d_cv.alloc((size_t)n * 4));
	HIP_TRY(d_crc.alloc((size_t)n * 8));
	HIP_TRY(hipMemcpy(d_e.p, ebits, (size_t)n * 494, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_m.p, sb_mask, (size_t)n, hipMemcpyHostToDevice));
	r = rach_dev(nullptr, n, d_e.as<int8_t>(), d_m.as<uint8_t>(), d_r.as<uint8_t>(), d_rv.as<int32_t>(),
	             d_cv.as<int32_t>(), d_crc.as<int32_t>());
	if (r) return r;
	HIP_TRY(hipStreamSynchronize(nullptr));
	HIP_TRY(hipMemcpy(rach, d_r.p, (size_t)n * 18, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(rv, d_rv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (conv) HIP_TRY(hipMemcpy(conv, d_cv.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (crc) HIP_TRY(hipMemcpy(crc, d_crc.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	return 0;
}

}  // namespace

extern "C" {

int gmr1_hip_xch_dc12_decode_batch_dev(void *stream, int n, const int8_t *ebits, uint8_t *l2, int32_t *crc,
                                       int32_t *conv)
{
	return xch_dev((hipStream_t)stream, n, ebits, l2, crc, conv);
}

int gmr1_hip_xch_dc12_decode_batch(int n, const int8_t *ebits, uint8_t *l2, int32_t *crc, int32_t *conv)
{
	return xch_host(n, ebits, l2, crc, conv);
}

int gmr1_hip_rach_decode_batch_dev(void *stream, int n, const int8_t *ebits, const uint8_t *sb_mask,
                                   uint8_t *rach, int32_t *rv, int32_t *conv, int32_t *crc)
{
	return rach_dev((hipStream_t)stream, n, ebits, sb_mask, rach, rv, conv, crc);
}

int gmr1_hip_rach_decode_batch(int n, const int8_t *ebits, const uint8_t *sb_mask, uint8_t *rach, int32_t *rv,
                               int32_t *conv, int32_t *crc)
{
	return rach_host(n, ebits, sb_mask, rach, rv, conv, crc);
}

// reference-compatible single calls (xch_dc12.h:38, rach.h:38-39)
int gmr1_xch_dc12_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv)
{
	if (!l2 || !bits_e)
		return fail(-EINVAL, "gmr1_xch_dc12_decode: NULL argument");
	int32_t crc = -1, conv = 0;
	int r = xch_host(1, reinterpret_cast<const int8_t *>(bits_e), l2, &crc, &conv);
	if (r) return r;
	if (conv_rv) *conv_rv = conv;
	return crc;
}

int gmr1_rach_decode(uint8_t *rach, const sbit_t *bits_e, uint8_t sb_mask, int *conv_rv, int *crc_rv)
{
	if (!rach || !bits_e)
		return fail(-EINVAL, "gmr1_rach_decode: NULL argument");
	int32_t rv = -1, conv = 0, crc[2] = {0, 0};
	int r = rach_host(1, reinterpret_cast<const int8_t *>(bits_e), &sb_mask, rach, &rv, &conv, crc);
	if (r) return r;
	if (conv_rv) *conv_rv = conv;
	if (crc_rv) {
		crc_rv[0] = crc[0];
		crc_rv[1] = crc[1];
	}
	return rv;
}

}  // extern "C"
