/* Normal burst formats (API of osmocom/osmo-gmr include/osmocom/gmr1/sdr/nb.h:37-46) */
#ifndef __OSMO_GMR1_SDR_NB_H__
#define __OSMO_GMR1_SDR_NB_H__

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_pi4cxpsk_burst;

extern struct gmr1_pi4cxpsk_burst gmr1_bcch_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_dc2_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_dc6_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_dc12_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_nt3_speech_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_nt3_facch_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_nt6_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_nt9_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_rach_burst;
extern struct gmr1_pi4cxpsk_burst gmr1_sdcch_burst;

#ifdef __cplusplus
}
#endif

#endif
