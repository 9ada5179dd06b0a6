// MLP backward kernels for 2 hidden layer(s) of 32 neurons (see mlp_split_bwd.inc).
#define DNS_BWD_NN 32
#define DNS_BWD_NL 2
#include "mlp_split_bwd.inc"
