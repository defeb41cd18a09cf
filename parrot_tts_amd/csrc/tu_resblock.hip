// tu_resblock.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "resblock_split.h"
namespace parrot {
hipError_t launch_resblock_split_f16x3(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split_s<SchF16x3>(C, p, s); }
hipError_t launch_resblock_split_bf16x6(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split_s<SchBf16x6>(C, p, s); }
}  // namespace parrot
