// tu_split16.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "conv_split16.h"
#include "resblock_split16.h"
namespace parrot {
hipError_t launch_conv_split16_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_s<SchF16x3>(variant, p, s); }
hipError_t launch_resblock_split16_f16x3(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split16_s<SchF16x3>(C, p, s); }
}  // namespace parrot
#ifdef S16_TRACE
extern "C" int parrot_debug_s16_trace(unsigned long long* out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(parrot::g_s16_trace), sizeof(unsigned long long) * 4 * 64);
}
#endif
