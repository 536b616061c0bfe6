// The receive loop's acquisition (gmr1_rx.c:605-744) with the glue between two sweeps done by the PRODUCING sweep's last
// thread instead of a kernel of its own (k_acq_glue, gmr1_dev.h): every launch in that chain is small and dependent, so a
// launch saved is its whole latency saved.  The stand-alone entry points pass step 0 and behave as before.
#pragma once
#include "gmr1_dev.h"

namespace gmr1 {

struct AcqTail {
	int step;                 // 0 none; 1..4: k_acq_glue's step, applied by the thread that holds the sweep's result
	const int32_t *skip_dead; // fine / SNR over the candidate slots: AcqArgs::live -- a slot without a candidate is not computed
	AcqArgs g;
};

hipError_t launch_fcch_rough_tail(const FcchRoughArgs &a, int ntaps, const AcqTail &t, hipStream_t stream);
hipError_t launch_fcch_multi_tail(const FcchMultiArgs &a, const AcqTail &t, hipStream_t stream);
hipError_t launch_fcch_fine_tail(const FcchFineArgs &a, int nsym, const AcqTail &t, hipStream_t stream);

// host side (capi_fcch.cpp): the _batch_dev entry points' bodies with a tail
int fcch_rough_tail(hipStream_t st, int tab, int n, int sps, int len, const float *iq, const uint64_t *offset,
                    const float *freq_shift, int32_t *toa, int32_t *rv, const AcqTail &t);
int fcch_rough_multi_tail(hipStream_t st, int tab, int n, int sps, int len, const float *iq, const uint64_t *offset,
                          const float *freq_shift, int32_t *peaks_toa, int N, int32_t *count, const AcqTail &t);
int fcch_fine_tail(hipStream_t st, int tab, int mode, int n, int sps, const float *iq, const uint64_t *offset,
                   const float *freq_shift, int32_t *toa, float *freq_err, float *snr, const AcqTail &t);

}  // namespace gmr1
