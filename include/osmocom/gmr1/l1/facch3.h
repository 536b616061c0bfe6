/* FACCH3 channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/facch3.h:39-40) */
#ifndef __OSMO_GMR1_L1_FACCH3_H__
#define __OSMO_GMR1_L1_FACCH3_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 x 104 soft bits (4 bursts) -> 10 bytes of L2 and 4 x 8 status bits.
 * ciph: optional 4 x 96 keystream bits.  Returns 0 when the CRC16 matches. */
int gmr1_facch3_decode(uint8_t *l2, ubit_t *bits_s,
                       const sbit_t *bits_e, const ubit_t *ciph, int *conv_rv);

/* facch3.h:37-38: 10 bytes (76 bits) -> 4 x 104 burst bits; bits_s: 4 x 8 status bits; ciph: optional 4 x 96 keystream bits */
void gmr1_facch3_encode(ubit_t *bits_e, const uint8_t *l2, const ubit_t *bits_s, const ubit_t *ciph);

#ifdef __cplusplus
}
#endif

#endif
