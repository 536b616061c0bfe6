/* GMR-1 SDR global definitions (API of osmocom/osmo-gmr include/osmocom/gmr1/sdr/defs.h) */
#ifndef __OSMO_GMR1_SDR_DEFS_H__
#define __OSMO_GMR1_SDR_DEFS_H__

#define GMR1_SYM_RATE 23400   /* GMR-1 symbol rate, symbols / second */

#endif
