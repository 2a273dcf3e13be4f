// tests/asan -- link-time stand-ins for the kernel launchers of libatsc_hip.so (the .hip files), so that
// the HOST sources of the library (atsc_host.cpp, atsc_stream.cpp, atsc_vsri.cpp) build with
// AddressSanitizer + UBSan on a box without a GPU.  Nothing here is ever reached: atsc_ctx_create fails
// with ATSC_E_NO_DEVICE before a launch can happen.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include "../../include/atsc_hip.h"
#include "../../atsc_amd/csrc/atsc_internal.h"

namespace atsc {
hipError_t launch_compress_class(int, uint32_t, uint32_t, const double *, const DevFrame *, const uint32_t *,
                                 const DevPlan *, const float2 *, const KParams &, uint8_t *, DevResult *,
                                 atsc_frame_diag *, const UniArgs &, hipStream_t, hipEvent_t, hipEvent_t)
{
    return hipErrorNotSupported;
}
hipError_t launch_compress_large(uint32_t, const double *, const DevFrame *, const uint32_t *, const DevPlan *,
                                 const float2 *, const KParams &, uint8_t *, DevResult *, atsc_frame_diag *,
                                 unsigned char *, uint64_t, uint32_t, hipStream_t, const LargePre *)
{
    return hipErrorNotSupported;
}
uint32_t resident_grid(int, uint32_t, uint32_t) { return 0; }
// same bound as atsc_large.hip needs is irrelevant here: the parser only takes the maximum
uint64_t large_ws_bytes(uint32_t n, uint32_t L, uint32_t kcap) { return 64ull * n + 32ull * L + 16ull * kcap; }
hipError_t launch_decompress_large(uint32_t, const struct DevDFrame *, const uint32_t *, const DevPlan *,
                                   const float2 *, const uint8_t *, double *, int *, unsigned char *, uint64_t,
                                   uint32_t, int, int, hipStream_t, const LargePre *, uint32_t)
{
    return hipErrorNotSupported;
}
hipError_t launch_order_by_cost(const uint32_t *, uint32_t *, const uint32_t *, uint8_t *, uint32_t *,
                                const uint32_t *, const uint32_t *, int, hipStream_t)
{
    return hipErrorNotSupported;
}
hipError_t launch_nonfinite_flag(const double *, uint64_t, uint32_t *, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_pack(const DevFrame *, const DevResult *, uint64_t, uint32_t *, uint64_t *, const uint8_t *,
                       uint8_t *, uint64_t, uint64_t *, uint8_t *, double *, const uint32_t *, uint32_t,
                       hipStream_t, uint64_t *)
{
    return hipErrorNotSupported;
}
hipError_t launch_copy_words(void *, const void *, uint64_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_decompress(const struct DevDFrame *, uint64_t, const uint32_t *, int, uint32_t, uint32_t,
                             const DevPlan *, const float2 *, const uint8_t *, double *, int *, hipStream_t)
{
    return hipErrorNotSupported;
}
}  // namespace atsc
