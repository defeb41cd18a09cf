// tu_mrf.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the whole-MRF
// instantiations of resblock_split_kernel (parity-grade schemes) behind the entry points below.
#include "resblock_split.h"
namespace parrot {
hipError_t launch_mrf_split_f16x3(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_mrf_split_s<SchF16x3>(C, p, s); }
}  // namespace parrot
