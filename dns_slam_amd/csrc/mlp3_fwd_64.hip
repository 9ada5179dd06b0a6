// The three-part-bf16 forward kernels (mlp3_fwd.inc) for 64 neurons per hidden layer.
#define DNS_FWD_NN 64
#include "mlp3_fwd.inc"
