/* GMR-1 CRC parameters (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/crc.h:36-38), for libosmocore's
 * osmo_crc8gen_* / osmo_crc16gen_* bit-wise routines.  The GPU codecs evaluate the same polynomials as per-bit
 * syndrome tables. */
#ifndef __OSMO_GMR1_L1_CRC_H__
#define __OSMO_GMR1_L1_CRC_H__

#include <osmocom/gmr1/compat.h>
#ifdef GMR1_HIP_USE_SYSTEM_OSMOCOM
#include <osmocom/core/crcgen.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

extern const struct osmo_crc8gen_code  gmr1_crc8;    /* D^8 + D^7 + D^4 + D^3 + D + 1          (RACH)                  */
extern const struct osmo_crc16gen_code gmr1_crc12;   /* D^12 + D^11 + D^3 + D^2 + D + 1        (RACH)                  */
extern const struct osmo_crc16gen_code gmr1_crc16;   /* D^16 + D^12 + D^5 + 1, init 0          (BCCH, CCCH, FACCH, xCH) */

#ifdef __cplusplus
}
#endif

#endif
