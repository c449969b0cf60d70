// bf16 MFMA instantiations of the MLP's input-gradient chain.
#include "mlp_bwd_impl.hpp"
namespace nerfx {
int launch_mlp_bwd_bf16(int activ, int wt, const BwdArgs& a, hipStream_t s) {
    return launch_bwd_activation<NERFX_PREC_BF16, RingBf16Bwd>(activ, wt, a, s);
}
}  // namespace nerfx
