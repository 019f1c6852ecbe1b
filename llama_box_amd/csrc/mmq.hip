// mmq.hip — prefill path: quantised weights x many activation columns through MFMA (placeholder until the
// LDS-dequant MFMA kernel lands; graph.cpp falls back to column chunks of the bandwidth kernel).
#include "kernels.h"
namespace mi355x {
bool mmq_supported(int, int64_t, int64_t, int64_t) { return false; }
size_t mmq_workspace_bytes(int, int64_t, int64_t, int64_t) { return 0; }
void launch_mmq(hipStream_t, int, const uint8_t *, int64_t, int, int, int, const void *, float *, int64_t) { abort(); }
}  // namespace mi355x
