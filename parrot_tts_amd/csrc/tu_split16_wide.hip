// tu_split16_wide.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the
// 64-row and 128 x 160 tile instantiations of conv_split16.h.
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_wide_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_wide_s<SchF16x3>(variant, p, s); }
}  // namespace parrot
