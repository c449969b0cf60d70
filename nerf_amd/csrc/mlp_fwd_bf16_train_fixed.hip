// bf16 MFMA instantiations of the TRAINING forward with compile-time depth (width 256, skip on).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_bf16_train_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s) {
    return launch_fixed_depth<NERFX_PREC_BF16, RingBf16Train, true>(activ, hd, a, s);
}
}  // namespace nerfx
