// MLP backward kernels for 1 hidden layer(s) of 32 neurons (see mlp_split_bwd.inc).
#define DNS_BWD_NN 32
#define DNS_BWD_NL 1
#include "mlp_split_bwd.inc"
