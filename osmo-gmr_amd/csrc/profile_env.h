// profile_env.h -- the one gate for profiling switches (host side).
#pragma once

#include <cstdlib>

namespace gmr1 {

// Profiling switches (phase cut-offs, alternative kernels, occupancy caps) change what the library computes or returns.
// They exist only in a library built with -DGMR1_HIP_PROFILE (osmo-gmr_amd/build.py --profile -> libgmr1_hip_prof.so,
// which the tools under tools/ load through GMR1_HIP_LIBRARY); the product build never looks at these variables.
inline const char *profile_env(const char *name)
{
#ifdef GMR1_HIP_PROFILE
	return getenv(name);
#else
	(void)name;
	return nullptr;
#endif
}

}  // namespace gmr1
