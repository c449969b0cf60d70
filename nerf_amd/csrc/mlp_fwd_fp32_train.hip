// fp32 MFMA instantiations of the fused MLP's TRAINING forward (records the tape).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_fp32_train(int activ, int wt, const MlpArgs& a, hipStream_t s) {
    return launch_activation<NERFX_PREC_FP32, RingFp32, true, true>(activ, wt, a, s);
}
}  // namespace nerfx
