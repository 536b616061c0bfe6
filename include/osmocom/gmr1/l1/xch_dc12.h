/* xCH over DC12 channel decoding (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/xch_dc12.h:37-38) */
#ifndef __OSMO_GMR1_L1_XCH_DC12_H__
#define __OSMO_GMR1_L1_XCH_DC12_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 432 soft bits of one DC12 burst -> 24 bytes of L2 (192 bits, LSB first).  Returns the CRC16 verdict
 * (0 = pass); *conv_rv = Viterbi path metric (optional). */
int gmr1_xch_dc12_decode(uint8_t *l2, const sbit_t *bits_e, int *conv_rv);

/* xch_dc12.h:37: 24 bytes -> 432 burst bits (CRC16, K=9 rate 1/3 tail-biting, P(12;13)); 0 or -errno */
int gmr1_xch_dc12_encode(ubit_t *bits_e, const uint8_t *l2);

#ifdef __cplusplus
}
#endif

#endif
