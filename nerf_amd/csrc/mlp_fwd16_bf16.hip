// Instantiations of the 16x16x32-MFMA inference kernel: bf16, width 129..256, skip connection,
// depth 4 or 8, three activations.
#include <algorithm>

#include "mlp_fwd16_impl.hpp"

namespace nerfx {
int launch_mlp_m16(int activ, int hd, const MlpArgs& a, hipStream_t s) {
#define NERFX_M16(ACT_) \
    case ACT_: return hd == 2 ? launch_cfg16<ACT_, 2>(a, s) : launch_cfg16<ACT_, 4>(a, s);
    switch (activ) {
        NERFX_M16(NERFX_ACT_RELU)
        NERFX_M16(NERFX_ACT_LEAKY_RELU)
        NERFX_M16(NERFX_ACT_SILU)
    }
#undef NERFX_M16
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}
}  // namespace nerfx
