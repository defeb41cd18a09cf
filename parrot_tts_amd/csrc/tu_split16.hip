// tu_split16.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "conv_split16.h"
#include "resblock_split16.h"
namespace parrot {
hipError_t launch_conv_split16_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_s<SchF16x3>(variant, p, s); }
hipError_t launch_resblock_split16_f16x3(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_split16_s<SchF16x3>(C, p, s); }
}  // namespace parrot
#ifdef S16_TRACE
namespace parrot {
hipError_t launch_conv_split16_small_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_small_s<SchF16x3>(variant, p, s); }
hipError_t launch_conv_split16_wide_f16x3(int variant, const ConvParams& p, hipStream_t s) { return launch_conv_split16_wide_s<SchF16x3>(variant, p, s); }
}  // namespace parrot
extern "C" int parrot_debug_s16_trace(unsigned long long* out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(parrot::g_s16_trace), sizeof(unsigned long long) * 4 * 64);
}
extern "C" int parrot_debug_s16_wg(unsigned long long* out_host, int n_wg) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(parrot::g_s16_wg), sizeof(unsigned long long) * 4 * (n_wg < parrot::S16_WG_MAX ? n_wg : parrot::S16_WG_MAX));
}
extern "C" int parrot_debug_s16_wg_clear() { 
    static unsigned long long zeros[parrot::S16_WG_MAX * 4];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(parrot::g_s16_wg), zeros, sizeof zeros);
}
#endif
