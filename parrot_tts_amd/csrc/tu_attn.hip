// tu_attn.hip -- one translation unit of libparrot_hip.so (parrot_tts_amd/build.py compiles them in parallel): the kernel
// instantiations behind the entry points below.
#include "attn.h"
namespace parrot {
hipError_t launch_attn_flash(const AttnParams& p, int B, hipStream_t s) {
    switch (p.hd) {
        case 128: return launch_attn_flash_t<128>(p, B, s);
        case 64: return launch_attn_flash_t<64>(p, B, s);
        case 32: return launch_attn_flash_t<32>(p, B, s);
        case 16: return launch_attn_flash_t<16>(p, B, s);
        default: return hipErrorInvalidValue;
    }
}
}  // namespace parrot
