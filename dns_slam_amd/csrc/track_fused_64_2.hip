#define DNS_BWD_NN 64
#define DNS_BWD_NL 2
#define DNS_TF_WITH_BEGIN
#include "track_fused.inc"
