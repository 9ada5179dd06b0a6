// half-rows MLP backward, 32 neurons x 2 hidden layer(s) (one translation unit per pair: parallel compilation)
#define DNS_BWD_NN 32
#define DNS_BWD_NL 2
#include "mlp_half_bwd.inc"
