/* BCCH channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/bcch.h:38) */
#ifndef __OSMO_GMR1_L1_BCCH_H__
#define __OSMO_GMR1_L1_BCCH_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 424 soft bits -> 24 bytes; returns 0 when the CRC16 matches; *conv_rv (optional) = Viterbi metric */
int gmr1_bcch_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv);

/* bcch.h:37: 24 bytes -> 424 burst bits (CRC16, K=5 rate 1/2, intra-burst interleaver, scrambler).  The bits are
 * computed on the GPU; a device failure leaves bits_e untouched and is reported by gmr1_hip_last_error(). */
void gmr1_bcch_encode(ubit_t *bits_e, const uint8_t *l2);

#ifdef __cplusplus
}
#endif

#endif
