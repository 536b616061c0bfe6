// host_tables.h -- burst-format tables on the host and their flattening.
#pragma once

#include <osmocom/gmr1/sdr/pi4cxpsk.h>
#include <osmocom/gmr1/sdr/nb.h>
#include <osmocom/gmr1/sdr/fcch.h>

#include "gmr1_dev.h"

namespace gmr1 {

extern gmr1_pi4cxpsk_burst *const kBuiltin[GMR1_HIP_N_BURSTS];
extern const char *const kBuiltinName[GMR1_HIP_N_BURSTS];

void tables_init();
// pointer-linked reference-style struct -> flat copy; 0 or -EINVAL
int flatten(const gmr1_pi4cxpsk_burst *b, gmr1_hip_burst_flat *out, const char *name);
// flat copy -> kernel descriptor; 0 or -EINVAL (limits: kMaxCoef sync symbols per sequence)
int to_dev(const gmr1_hip_burst_flat &f, DevBurst *d);

// FCCH reference waveforms and DFT tables, computed on the host with the reference's own float
// formulas (fcch.c:92-121,167-193,575-580) so they match a CPU build bit for bit
void fcch_tables_init(FcchTables *t);
extern const struct gmr1_fcch_burst *const kFcchBuiltin[kFcchTabs];

}  // namespace gmr1
