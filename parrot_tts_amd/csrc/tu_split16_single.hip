// tu_split16_single.hip -- one translation unit of libparrot_hip.so: conv_split16_kernel (16x16x32 MFMA, 32-channel chunks) for the
// single-MFMA bf16 operating point (BASELINE configs[2]): the wide k >= 7 layers of the reduced-precision companion mode, plain and
// operand-plane (XPL) instantiations.  (fp16: tu_split16_single_f16.hip -- two units, compiled in parallel.)
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_bf16(int variant, const ConvParams& p, hipStream_t s) {
    return (variant == 2 || variant == 3)   ? launch_conv_split16_small_s<SchBf16>(variant, p, s)
           : (variant == 1 || variant == 4) ? launch_conv_split16_wide_s<SchBf16>(variant, p, s)
                                            : launch_conv_split16_s<SchBf16>(variant, p, s);
}
hipError_t launch_conv_split16_xpl_bf16(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_xpl_s<SchBf16>(variant, p, s); }
}  // namespace parrot
