// Launcher of the debug-tap kernel (rx_debug_kernels.inc): the four intermediate vectors the reference dumps under
// ENABLE_DEBUG_SIGNAL (include/osmocom/gmr1/sdr/defs.h:35-39, pi4cxpsk.c:251,345,545,582), for ONE burst.
#pragma once
#include "gmr1_dev.h"

namespace gmr1 {

struct RxTapsOut {
	float *corr;      // [w]            w = in_len - symbols * sps + 1
	float2 *burst;    // [in_len]
	float2 *align;    // [symbols]
	float2 *final_;   // [symbols]
};

// a.n == 1; a.ssyms and a.rv required; every pointer is device memory
hipError_t launch_rx_taps(const RxArgs &a, const RxTapsOut &o, hipStream_t stream);

}  // namespace gmr1
