// tu_split16_small.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the
// small-tile instantiations of conv_split16.h (launches that do not give every CU a workgroup).
#ifndef S16_TRACE  // (trace builds: tu_split16.hip holds these too, beside the trace buffers)
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_small_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_small_s<SchF16x3>(variant, p, s); }
}  // namespace parrot
#endif
