#include "psl_decode.h"
namespace psl {
int launch_decode_bwd(psl_ctx* ctx, const DecodeArgs& a, const psl_render_grads& g, hipStream_t s) {
  set_error("decode backward not built yet");
  return PSL_ERR_UNSUPPORTED;
}
}  // namespace psl
