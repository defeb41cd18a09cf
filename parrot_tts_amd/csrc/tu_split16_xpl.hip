// tu_split16_xpl.hip -- one translation unit of libparrot_hip.so: the operand-plane (XPL) instantiations of conv_split16_kernel for
// the fp16x3 scheme (PARROT_PLANES=1; off by default there).  A unit of its own: the 128 x 160 instantiations compile for minutes.
#include "conv_split16.h"
namespace parrot {
hipError_t launch_conv_split16_xpl_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_xpl_s<SchF16x3>(variant, p, s); }
}  // namespace parrot
