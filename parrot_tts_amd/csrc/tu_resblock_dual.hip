// tu_resblock_dual.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the
// dual-window anti-phase fused ResBlock pair kernels (resblock_dual.h) for the default scheme.
#include "resblock_dual.h"
#include "resblock_pdual.h"
namespace parrot {
hipError_t launch_resblock_dual_f16x3(int C, int nwin, const ResblockSplitParams& p, hipStream_t s) { return launch_resblock_dual_s<SchF16x3>(C, nwin, p, s); }
hipError_t launch_resblock_pdual_f16x3(int C, const ResblockSplitParams& p, int n_cus, hipStream_t s) { return launch_resblock_pdual_s<SchF16x3>(C, p, n_cus, s); }
}  // namespace parrot
#ifdef RBD_TRACE
extern "C" int parrot_debug_rbd_trace(unsigned long long* out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(parrot::g_rbd_trace), sizeof(unsigned long long) * 8 * 64);
}
#endif
