// bf16 MFMA instantiations of the fused MLP for encoders wider than four k-units (Fourier
// features) with the depth fixed at compile time (width 256, skip connection, depth 4 or 8).
#include "mlp_fwd_impl.hpp"
namespace nerfx {
int launch_mlp_bf16_bigenc_fixed(int activ, int hd, const MlpArgs& a, hipStream_t s) {
    return launch_fixed_depth<NERFX_PREC_BF16, RingBf16, false, true>(activ, hd, a, s);
}
}  // namespace nerfx
