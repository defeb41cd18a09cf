// tu_split16.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_s<SchF16x3>(variant, p, s); }
hipError_t launch_conv_split16_small_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_small_s<SchF16x3>(variant, p, s); }
}  // namespace parrot
