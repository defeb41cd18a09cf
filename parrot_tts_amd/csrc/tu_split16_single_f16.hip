// tu_split16_single_f16.hip -- one translation unit of libparrot_hip.so: conv_split16_kernel for the single-MFMA fp16 operating point
// (see tu_split16_single.hip).
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_f16(int variant, const ConvParams& p, hipStream_t s) {
    return (variant == 2 || variant == 3)   ? launch_conv_split16_small_s<SchF16>(variant, p, s)
           : (variant == 1 || variant == 4) ? launch_conv_split16_wide_s<SchF16>(variant, p, s)
                                            : launch_conv_split16_s<SchF16>(variant, p, s);
}
hipError_t launch_conv_split16_xpl_f16(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_xpl_s<SchF16>(variant, p, s); }
}  // namespace parrot
