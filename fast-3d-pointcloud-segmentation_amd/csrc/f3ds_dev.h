// f3ds_dev.h -- the gate in front of the development switches.
//
// The library has ~25 result-neutral F3DS_* environment switches (kernel layouts, stage-0 path, sweep machinery,
// scratch sizing, tracing) that exist for tests, A/B runs and profiling.  None of them is read unless F3DS_DEV is set
// to something other than "0": a stray F3DS_MERGE_NW or F3DS_VOX_TILES in a user's environment cannot change which
// kernels a production process runs.  With the gate open f3ds_version_string() ends in " +dev" and bench.py reports
// no `value`.  (F3DS_RCCL_LIB -- where librccl lives -- and HIP's own GPU_MAX_HW_QUEUES are configuration, not
// development switches, and stay ungated.)
#ifndef F3DS_DEV_H_
#define F3DS_DEV_H_
#include <cstdlib>

namespace f3ds {
inline bool dev_mode() {
    const char* e = getenv("F3DS_DEV");
    return e && !(e[0] == '0' && e[1] == '\0') && e[0] != '\0';
}
// getenv for a development switch: the variable only exists while the gate is open
inline const char* dev_getenv(const char* name) { return dev_mode() ? getenv(name) : nullptr; }
}  // namespace f3ds
#endif  // F3DS_DEV_H_
