// diag_env.h -- diagnostic switches of the kernels' launch code.
//
// The release library (`make all`) reads NO environment variable: IRIS_DIAG_ENV(name, dflt) is the
// constant `dflt` there, so a stray variable in a caller's environment cannot change what a forward
// computes or how it is launched (tests/test_host_logic.py scans the built .so for switch names).
// The diagnostic builds (`make diag`, `make stamps`: -DIRIS_MRF_DIAG, loaded only through
// IRIS_HIFIGAN_LIB by the scripts under tools/) read each switch once.
#pragma once
#include <stdlib.h>

namespace iris {

#ifdef IRIS_MRF_DIAG
inline int diag_env_read(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define IRIS_DIAG_ENV(name, dflt) ([] { static const int v__ = ::iris::diag_env_read(name, dflt); return v__; }())
#else
#define IRIS_DIAG_ENV(name, dflt) (dflt)
#endif

}  // namespace iris
