/* RACH channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/rach.h:37-39) */
#ifndef __OSMO_GMR1_L1_RACH_H__
#define __OSMO_GMR1_L1_RACH_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 494 soft bits of one RACH burst -> 18 bytes (2 class-1 bytes, then 123 class-2 bits, LSB first; the upper
 * five bits of rach[17] = 0).  sb_mask: the spot beam's SB mask.  Returns 0 when both CRCs pass;
 * *conv_rv = Viterbi path metric, crc_rv[0] / crc_rv[1] = CRC8 / CRC12 verdicts (both optional). */
int gmr1_rach_decode(uint8_t *rach, const sbit_t *bits_e, uint8_t sb_mask, int *conv_rv, int *crc_rv);

/* rach.h:37: 18 bytes -> 494 burst bits (CRC8 ^ sb_mask, CRC12, K=5 rate 1/4, class-1 part sent twice) */
void gmr1_rach_encode(ubit_t *bits_e, const uint8_t *rach, uint8_t sb_mask);

#ifdef __cplusplus
}
#endif

#endif
