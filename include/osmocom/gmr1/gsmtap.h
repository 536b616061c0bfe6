/* GSMTAP helper of the GMR-1 tools (API of osmocom/osmo-gmr include/osmocom/gmr1/gsmtap.h:35-37).
 *
 * Declaration only.  As in the reference (src/Makefile.am:8 compiles src/gsmtap.c into gmr1_rx itself) the
 * definition belongs to the program, because it allocates a libosmocore `struct msgb` (msgb_alloc / msgb_put),
 * which this library does not link.  A caller without libosmocore builds the same 16-byte gsmtap_hdr + L2 packet
 * with gmr1_hip_gsmtap_pack() (gmr1_hip.h). */
#ifndef __OSMO_GMR1_GSMTAP_H__
#define __OSMO_GMR1_GSMTAP_H__

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct msgb;

/* chan_type: one of GSMTAP_GMR1_*; fn / tn: frame and timeslot number; l2: `len` payload bytes */
struct msgb *gmr1_gsmtap_makemsg(uint8_t chan_type, uint32_t fn, uint8_t tn, const uint8_t *l2, int len);

#ifdef __cplusplus
}
#endif

#endif
