// fp32 MFMA instantiations of the fused MLP
// (one translation unit per precision and encoder class so that they compile in parallel).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_fp32(int activ, int wt, const MlpArgs& a, hipStream_t s) {
    return launch_activation<NERFX_PREC_FP32, RingFp32, false>(activ, wt, a, s);
}
}  // namespace nerfx
