/* CCCH channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/ccch.h:38) */
#ifndef __OSMO_GMR1_L1_CCCH_H__
#define __OSMO_GMR1_L1_CCCH_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 432 soft bits -> 24 bytes; returns 0 when the CRC16 matches; *conv_rv (optional) = Viterbi metric */
int gmr1_ccch_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv);

/* ccch.h:37: 24 bytes -> 432 burst bits (the BCCH coding between 4 + 4 padding bits, then scrambled) */
void gmr1_ccch_encode(ubit_t *bits_e, const uint8_t *l2);

#ifdef __cplusplus
}
#endif

#endif
