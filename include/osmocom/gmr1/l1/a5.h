/* GMR-1 A5 ciphering (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/a5.h:37-41) */
#ifndef __OSMO_GMR1_L1_A5_H__
#define __OSMO_GMR1_L1_A5_H__

#include <stdint.h>
#include <osmocom/gmr1/compat.h>

#ifdef __cplusplus
extern "C" {
#endif

/* n = 0: all-zero streams; n = 1: A5/1; other n: buffers untouched.  key: 8 bytes as received from the
 * SIM; nbits downlink bits to dl, the next nbits to ul; either may be NULL. */
void gmr1_a5(int n, uint8_t *key, uint32_t fn, int nbits, ubit_t *dl, ubit_t *ul);
void gmr1_a5_1(uint8_t *key, uint32_t fn, int nbits, ubit_t *dl, ubit_t *ul);

#ifdef __cplusplus
}
#endif

#endif
