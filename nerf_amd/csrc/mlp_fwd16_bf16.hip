// Instantiations of the 16x16x32-MFMA inference kernel: bf16, width 129..256, skip connection,
// depth 4 or 8, three activations.
#include <algorithm>

#include "mlp_fwd16_impl.hpp"

namespace nerfx {
// `is_static`: the stream lengths are whole ring lengths (ModelPlan::m16_static), which the depth-4
// models' are: they take the kernels whose ring arithmetic is folded at compile time.
int launch_mlp_m16(int activ, int hd, bool is_static, const MlpArgs& a, hipStream_t s) {
#define NERFX_M16(ACT_)                                                                                  \
    case ACT_:                                                                                           \
        if (hd == 2 && is_static)                                                                        \
            return a.rgb ? launch_cfg16<ACT_, 2, M16_STATIC_COLOUR>(a, s) : launch_cfg16<ACT_, 2, M16_STATIC_DENSITY>(a, s); \
        return hd == 2 ? launch_cfg16<ACT_, 2, M16_DYNAMIC>(a, s) : launch_cfg16<ACT_, 4, M16_DYNAMIC>(a, s);
    switch (activ) {
        NERFX_M16(NERFX_ACT_RELU)
        NERFX_M16(NERFX_ACT_LEAKY_RELU)
        NERFX_M16(NERFX_ACT_SILU)
    }
#undef NERFX_M16
    return fail(NERFX_E_ARG, "unknown activation %d", activ);
}
}  // namespace nerfx
