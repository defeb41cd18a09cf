// tu_split_single.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "resblock_split.h"
namespace parrot {
hipError_t launch_conv_split_bf16(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split_generic<SchBf16>(variant, p, s); }
hipError_t launch_conv_split_f16(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split_generic<SchF16>(variant, p, s); }
hipError_t launch_resblock_split_bf16(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split_s<SchBf16>(C, p, s); }
hipError_t launch_resblock_split_f16(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split_s<SchF16>(C, p, s); }
}  // namespace parrot
