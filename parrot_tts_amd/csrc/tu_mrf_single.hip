// tu_mrf_single.hip -- one translation unit of libparrot_hip.so: the whole-MRF instantiations of resblock_split_kernel for the
// single-MFMA (reduced precision) schemes.
#include "resblock_split.h"
namespace parrot {
hipError_t launch_mrf_split_bf16(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_mrf_split_s<SchBf16>(C, p, s); }
hipError_t launch_mrf_split_f16(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_mrf_split_s<SchF16>(C, p, s); }
}  // namespace parrot
