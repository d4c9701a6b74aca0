// Temporary: stages not yet implemented return MA_ERR_ARG with a message (replaced as kernels land).
#include "ma_internal.h"
namespace ma {
int launch_genotype(ma_ctx* ctx, const DBatch&, const ma_asm_out_t&, const ma_var_out_t&, const ma_geno_out_t&) {
  ctx->err = "ma_genotype_batch: not implemented yet";
  return MA_ERR_ARG;
}
}  // namespace ma
