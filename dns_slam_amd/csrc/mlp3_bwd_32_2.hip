// One (n_neurons, n_hidden_layers) instantiation of the three-part-bf16 backward kernel (mlp3_bwd.inc).
#define DNS_BWD_NN 32
#define DNS_BWD_NL 2
#include "mlp3_bwd.inc"
