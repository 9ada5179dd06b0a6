#define DNS_BWD_NN 32
#define DNS_BWD_NL 1
#include "track_fused.inc"
