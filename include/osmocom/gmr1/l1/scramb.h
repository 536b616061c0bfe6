/* GMR-1 scrambler (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/scramb.h:36-37) */
#ifndef __OSMO_GMR1_L1_SCRAMB_H__
#define __OSMO_GMR1_L1_SCRAMB_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[i] = in[i] with its sign / value flipped where the 15-bit LFSR (seed 0x4d4b, restarted at every call) gives 1.
 * out may equal in.  One blocking trip to the GPU per call -- the decoders and encoders of this library have the
 * scrambler folded into their own kernels; these two exist for callers that use the primitive on its own.  A device
 * failure leaves out untouched and is reported by gmr1_hip_last_error(). */
void gmr1_scramble_sbit(sbit_t *out, const sbit_t *in, int len);
void gmr1_scramble_ubit(ubit_t *out, const ubit_t *in, int len);

#ifdef __cplusplus
}
#endif

#endif
