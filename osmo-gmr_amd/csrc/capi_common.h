// capi_common.h -- helpers shared by the C-ABI translation units (host only).
#pragma once

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <hip/hip_runtime.h>

#include "gmr1_dev.h"
#include "host_tables.h"

namespace gmr1 {

int fail(int code, const char *fmt, ...);
const char *last_error();
// 1 when the layer-1 chains follow libosmocore's accelerated Viterbi decoder (gmr1_hip_set_conv_decoder), else 0
int conv_acc();

#define HIP_TRY(expr)                                                                  \
	do {                                                                               \
		hipError_t e_ = (expr);                                                        \
		if (e_ != hipSuccess)                                                          \
			return ::gmr1::fail(e_ == hipErrorNoDevice ? -ENODEV : -EIO, "%s: %s", #expr, \
			                    hipGetErrorString(e_));                                \
	} while (0)

struct DevState {
	bool ready = false;       // constant tables uploaded to this device
	void *ws = nullptr;       // grow-only scratch for kernels that need workspace
	size_t ws_bytes = 0;
	// users of the workspace (and of the receive loop's side stream and events) take turns: WsLease
	std::recursive_mutex ws_mu;
	hipEvent_t ws_ev = nullptr;   // recorded behind the last user's kernels
	int ws_depth = 0;
};

// One user of the device's shared workspace at a time, from any thread on any stream.  The host part of a call runs under
// the device's lock (a second thread waits its turn); the device part is ordered by an event: a call on another stream
// first makes ITS stream wait for the kernels of the previous user, so nobody's scratch is overwritten or freed under
// running kernels.  Re-entrant (a workspace user may call another one on the same stream: the receive loop calls the
// FCCH sweeps).  Calls that need no workspace (every burst-level _batch_dev entry in its usual shape) take no lease.
class WsLease {
public:
	WsLease() = default;
	WsLease(const WsLease &) = delete;
	WsLease &operator=(const WsLease &) = delete;
	~WsLease();
	int acquire(DevState *s, hipStream_t st);

private:
	DevState *s_ = nullptr;
	hipStream_t st_ = nullptr;
};

extern DevBurst g_host_types[kNumTypes];
// guards the descriptor-table slots caller-defined burst types are uploaded into (one demodulator slot, four detector
// slots): held from the upload until the kernel that reads them has finished
std::mutex &custom_slots_mutex();
int host_types();
// state of the CURRENT device; uploads the constant tables on first use
int dev_state(DevState **out);
// grow-only device scratch of the current device; the caller holds a WsLease from here to its last launch
int dev_workspace(DevState *s, size_t bytes, void **out);

// fused BCCH / CCCH receive with the optional burst_energy() output (capi.cpp)
// the fused BCCH / DC6 launch arguments without the per-call pointers (rx_base_args, capi.cpp): for the one-burst server
int rx_fused_base_args(int sps, const float *iq, RxArgs *out);
int rx_bcch_ccch_dev_impl(hipStream_t stream, int n, int sps,
                          const float *iq, const uint64_t *offset, const uint8_t *kind,
                          const float *freq_shift,
                          uint8_t *l2, int32_t *crc, int32_t *conv,
                          float *toa, float *freq_err, float *energy,
                          int8_t *ebits, float *ssyms, int32_t *rv, long long plane_stride = 0);

// process_bcch of n_chains chains in one launch (capi.cpp / launch_rx_loop); every pointer in `la` is device memory
int rx_loop_dev_impl(hipStream_t stream, int n_chains, int sps, const float *iq, const RxLoopArgs &la);

// gmr1_hip_rx_run_dev that also reports how many records each carrier contributed (capi_rx.cpp; the sharded entry keeps
// the records on the device and still needs the per-carrier counts)
int rx_run_dev_counted(void *stream, int n_arfcn, int sps, const float *iq, const uint64_t *offset, const uint64_t *length,
                       const uint16_t *arfcn, struct gmr1_hip_rx_record *out, int max_records, int *n_records,
                       int32_t *status, int32_t *n_chains, int32_t *rec_per_carrier);

// TCH9 bursts of several interleaver runs of unequal length in one launch (capi_nt9.cpp)
int tch9_runs_dev_impl(hipStream_t st, int mode, int n, const int32_t *seq_pos, const int8_t *ebits, const uint8_t *ciph,
                       uint8_t *l2, int32_t *conv);

// gmr1_hip_demod_batch_dev of a built-in burst type, plus burst_energy() of each window (capi.cpp)
int demod_dev_energy(hipStream_t st, int burst_id, int n, int sps, int in_len, const float *iq,
                     const uint64_t *offset, const float *freq_shift, int8_t *ebits, int ebits_stride,
                     int32_t *sync_id, float *toa, float *energy, int32_t *rv);

// RAII device buffer for the host-pointer variants
struct DBuf {
	void *p = nullptr;
	~DBuf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
	template <typename T> T *as() { return static_cast<T *>(p); }
};

}  // namespace gmr1
