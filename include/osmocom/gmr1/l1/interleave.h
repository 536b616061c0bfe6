/* Inter-burst de-interleaver state (API of osmocom/osmo-gmr include/osmocom/gmr1/l1/interleave.h:40-56).
 * The struct layout is the
 * reference's (callers declare it themselves, gmr1_rx.c:90); what bits_cpp points to is private to this
 * library: the raw soft bits and key stream of the previous N - 1 bursts, which the GPU decoder gathers
 * from (the de-interleaving itself is part of the kernel's gather, nt9_kernels.hip). */
#ifndef __OSMO_GMR1_L1_INTERLEAVE_H__
#define __OSMO_GMR1_L1_INTERLEAVE_H__

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct gmr1_interleaver {
	int N;              /* interleaver depth (3) */
	int K;              /* interleaver width (648) */
	int n;              /* current burst number */
	uint8_t *bits_cpp;  /* state storage */
};

/* Intra-burst interleaver (interleave.h:36-37): 8 N bits (bytes: ubits or sbits alike), out != in.  One blocking trip
 * to the GPU per call, like the scrambler primitives (scramb.h). */
void gmr1_interleave_intra(void *out, const void *in, int N);
void gmr1_deinterleave_intra(void *out, const void *in, int N);

/* Inter-burst interleaver on its own (interleave.h:53-56): K bits per call, state in `il` exactly as the reference
 * keeps it (N rows of K bits).  An object used with these two calls must not also be handed to gmr1_tch9_encode /
 * gmr1_tch9_decode, which keep their own history in it. */
void gmr1_interleave_inter(struct gmr1_interleaver *il, void *bits_epp, void *bits_ep);
void gmr1_deinterleave_inter(struct gmr1_interleaver *il, void *bits_ep, void *bits_epp);

/* 0 / -ENOMEM; -EINVAL unless (N, K) = (3, 648), the only geometry GMR-1 uses (gmr1_rx.c:273) */
int  gmr1_interleaver_init(struct gmr1_interleaver *il, int N, int K);
void gmr1_interleaver_fini(struct gmr1_interleaver *il);

#ifdef __cplusplus
}
#endif

#endif
