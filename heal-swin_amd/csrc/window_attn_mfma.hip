// MFMA path of the fused shift + window attention op (Ws = 64, head_dim = 32, bf16).  Placeholder until
// the kernel lands: reports "unsupported" so the dispatcher uses the fp32-VALU path.
#include "window_attn.h"

namespace hs {
bool attn_mfma_supported(const AttnParams&, int) { return false; }
int launch_attn_fwd_mfma(const AttnParams&, hipStream_t) { return fail(HS_ERR_UNSUPPORTED, "mfma path not built"); }
int launch_attn_bwd_mfma(const AttnParams&, hipStream_t) { return fail(HS_ERR_UNSUPPORTED, "mfma path not built"); }
}  // namespace hs
