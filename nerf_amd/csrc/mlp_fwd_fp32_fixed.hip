// fp32 MFMA instantiations of the fused MLP with the depth fixed at compile time
// (width 256, skip connection, depth 4 or 8): straight-line block bodies.
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_fp32_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s) {
    return launch_fixed_depth<NERFX_PREC_FP32, RingFp32>(activ, hd, a, s);
}
}  // namespace nerfx
