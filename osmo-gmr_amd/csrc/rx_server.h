// The reference's one-burst calls without a launch per call: a one-wave "server" kernel that stays on the device between
// calls and takes its requests from a mailbox in pinned host memory (rx_server_kernels.inc, capi.cpp: OneBurst).
#pragma once
#include "gmr1_dev.h"

namespace gmr1 {

struct OneMail {
	uint32_t req;       // host -> device: sequence number of the request in the block (changes = a new request)
	uint32_t done;      // device -> host: the last request answered
	uint32_t ended;     // device -> host: the generation of the server that ended last (idle / lifetime / superseded)
	uint32_t gen;       // host -> device: the server generation that is to serve (from 1); an older one ends when it sees a newer
};

// `a`: the fused BCCH / DC6 launch arguments with n = 1 (every pointer into the pinned block, device addresses), 4 samples per
// symbol; idle_us / life_us: the server ends after that long without a request / in total (it ALWAYS ends: nothing may
// spin on the device for ever)
hipError_t launch_one_server(const RxArgs &a, OneMail *mail_dev, uint32_t gen, unsigned idle_us, unsigned life_us, hipStream_t stream);

}  // namespace gmr1
