// tu_mrf.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the whole-MRF
// instantiations of resblock_split_kernel (parity-grade schemes) behind the entry points below.
#include "resblock_split.h"
namespace parrot {
hipError_t launch_mrf_split_f16x3(int C, const ResblockSplitParams& p, hipStream_t s) { return launch_mrf_split_s<SchF16x3>(C, p, s); }
#ifdef RBS_TRACE
}  // namespace parrot
// (experiment builds only) copy the phase stamps of the traced workgroup to the host: 8 waves x RBS_TRACE_STAMPS
extern "C" int parrot_debug_rbs_trace(unsigned long long* out_host, int n) {
    if (n > 8 * parrot::RBS_TRACE_STAMPS) n = 8 * parrot::RBS_TRACE_STAMPS;
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(parrot::g_rbs_trace), (size_t)n * sizeof(unsigned long long));
}
namespace parrot {
#endif
}  // namespace parrot
