/*
 * osmocom/gmr1/compat.h -- the few libosmocore / libosmo-dsp TYPES that appear
 * in the GMR-1 C API signatures (sbit_t, ubit_t, struct osmo_cxvec).
 *
 * When the real libraries are installed, build with
 * -DGMR1_HIP_USE_SYSTEM_OSMOCOM and their headers are used instead; the
 * layouts below follow them (libosmocore include/osmocom/core/bits.h,
 * libosmo-dsp include/osmocom/dsp/cxvec.h) so the ABI is the same either way.
 */
#ifndef OSMO_GMR1_COMPAT_H
#define OSMO_GMR1_COMPAT_H

#include <stdint.h>

#ifdef GMR1_HIP_USE_SYSTEM_OSMOCOM
#include <osmocom/core/bits.h>
#include <osmocom/dsp/cxvec.h>
#ifdef __cplusplus
typedef struct { float re, im; } gmr1_cfloat;
#else
#include <complex.h>
typedef float complex gmr1_cfloat;
#endif
#else /* own definitions */

typedef int8_t  sbit_t;   /* soft bit: +127 = confident 0, -127 = confident 1, 0 = unknown */
typedef uint8_t ubit_t;   /* unpacked bit, 0 or 1 */
typedef uint8_t pbit_t;   /* packed bits */

#ifdef __cplusplus
typedef struct { float re, im; } gmr1_cfloat;   /* same layout as C99 float complex */
#else
#include <complex.h>
typedef float complex gmr1_cfloat;
#endif

#define CXVEC_FLG_REAL_ONLY (1 << 0)

struct osmo_cxvec {
	int len;            /* valid samples              */
	int max_len;        /* capacity                   */
	int flags;          /* CXVEC_FLG_*                */
	gmr1_cfloat *data;  /* samples (may point at _data) */
	gmr1_cfloat _data[0];
};

#endif /* GMR1_HIP_USE_SYSTEM_OSMOCOM */

#endif /* OSMO_GMR1_COMPAT_H */
