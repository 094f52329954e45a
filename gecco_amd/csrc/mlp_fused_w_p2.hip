// The one-launch point MLP of the "w2" mode at feature_dim 128 and 256: its own translation unit, so that the instantiations of
// mlp_fused_w.hip build side by side (that file holds the kernel and the entry points).
#define MFW_PART 2
#include "mlp_fused_w.hip"
