// bf16 MFMA instantiations of the fused MLP's TRAINING forward (records the tape).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_bf16_train(int activ, int wt, const MlpArgs& a, hipStream_t s) {
    return launch_activation<NERFX_PREC_BF16, RingBf16, true, true>(activ, wt, a, s);
}
}  // namespace nerfx
