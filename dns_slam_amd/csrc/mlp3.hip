// Host-side launcher of the three-part-bf16 MLP backward (device code: mlp3.hpp, mlp3_bwd.inc; one translation unit per
// (n_neurons, n_hidden_layers): mlp3_bwd_*.hip).  Arguments are validated by the C-ABI wrappers in mlp.hip.
#include "mlp3.hpp"

namespace dns {
#ifdef DNS_BWD_TRACE
namespace sp { extern unsigned long long* g_bwd_trace; }
#endif

int launch_mlp_bwd3(const float* x, uint32_t ldx, const float* x2, uint32_t ldx2, uint32_t n_in1, const float* dy,
                    uint32_t lddy, const float* params, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                    uint32_t n_hidden_layers, float* d_x, uint32_t lddx, float* d_x2, uint32_t lddx2, float* d_params,
                    uint32_t n_slots, const int32_t* row_index, const int32_t* tile_group, uint32_t param_stride, int acc1,
                    int acc2, hipStream_t st) {
  using namespace m3;
  BwdArgs a;
  a.x = x; a.ldx = ldx; a.seg = {x2, ldx2, x2 ? n_in1 : n_in}; a.dy = dy; a.lddy = lddy; a.params = params;
  a.n_in = n_in; a.n_out = n_out; a.dx = d_x; a.lddx = lddx;
  a.dseg = {(d_x && x2) ? d_x2 : nullptr, lddx2, (uint32_t)acc1, (uint32_t)acc2};
  a.d_params = d_params;
#ifdef DNS_BWD_TRACE
  a.trace = sp::g_bwd_trace;
#endif
  a.n_slots = n_slots; a.row_index = row_index; a.tile_group = tile_group; a.param_stride = param_stride;
  const uint32_t n_btiles = (n_slots + 127u) / 128u;
  uint32_t tpb = (n_btiles + 255u) / 256u;       // one workgroup of 4 waves per CU, contiguous tile ranges
  if (tpb < 1) tpb = 1;
  a.tiles_per_block = tpb;
  const uint32_t blocks = (n_btiles + tpb - 1) / tpb;
  if (n_neurons == 32 && n_hidden_layers == 1) return launch_bwd3_32_1(a, blocks, st);
  if (n_neurons == 32 && n_hidden_layers == 2) return launch_bwd3_32_2(a, blocks, st);
  if (n_neurons == 64 && n_hidden_layers == 1) return launch_bwd3_64_1(a, blocks, st);
  return launch_bwd3_64_2(a, blocks, st);
}

}  // namespace dns
