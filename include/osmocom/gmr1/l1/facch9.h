/* FACCH9 channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/facch9.h:39-41) */
#ifndef __OSMO_GMR1_L1_FACCH9_H__
#define __OSMO_GMR1_L1_FACCH9_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 662 soft bits of one NT9 burst -> 38 bytes of L2 (300 bits, LSB first; upper nibble of l2[37] = 0),
 * 10 SACCH and 4 status soft bits.  ciph: optional 658 keystream bits.  Returns the CRC16 verdict
 * (0 = pass); *conv_rv = Viterbi path metric. */
int gmr1_facch9_decode(uint8_t *l2, sbit_t *bits_sacch, sbit_t *bits_status,
                       const sbit_t *bits_e, const ubit_t *ciph, int *conv_rv);

/* facch9.h:37-39: 38 bytes (300 bits) -> 662 burst bits; bits_sacch 10, bits_status 4, ciph optional 658 */
void gmr1_facch9_encode(ubit_t *bits_e, const uint8_t *l2, const ubit_t *bits_sacch, const ubit_t *bits_status,
                        const ubit_t *ciph);

#ifdef __cplusplus
}
#endif

#endif
