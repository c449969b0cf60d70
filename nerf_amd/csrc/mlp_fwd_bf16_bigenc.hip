// bf16 MFMA instantiations of the fused MLP, encoders wider than the register cache
// (one translation unit per precision and encoder class so that they compile in parallel).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_bf16_bigenc(int activ, int wt, const MlpArgs& a, hipStream_t s) {
    return launch_activation<NERFX_PREC_BF16, RingBf16, true>(activ, wt, a, s);
}
}  // namespace nerfx
