// The three-part-bf16 forward kernels (mlp3_fwd.inc) for 32 neurons per hidden layer.
#define DNS_FWD_NN 32
#include "mlp3_fwd.inc"
