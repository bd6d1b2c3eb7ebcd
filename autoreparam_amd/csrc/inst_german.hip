// Chain-kernel instantiations for german_credit_lognormalcentered: the padded
// 64-column design matrix is split over K lanes x 64/K consecutive features.
#include "host_common.h"

namespace arp {
// chain kernels from the lanes sized for 4 waves per workgroup, the VI kernel from the 4-lane (matrix-core)
// instantiation in its row-part form, sized for the VI workgroup
static LaneOps with_vi(LaneOps o, const LaneOps& vi) {
  o.vi = vi.vi; o.vi_block = vi.vi_block; o.vi_parts = vi.vi_parts; o.vi_occ = vi.vi_occ; o.vi_dmax = vi.vi_dmax;
  return o;
}
const LaneOps& german_bf3_ops() {
  static const LaneOps o = with_vi(Launch<GermanLane<4, 16, kBlock / 64, false, true>>::ops(),
                                   Launch<GermanLane<4, 16, kGermanViBlock / 64, true, true>>::vi_only());
  return o;
}
const std::vector<LaneOps>& german_ops() {
  static const std::vector<LaneOps> t = {
      with_vi(Launch<GermanLane<4, 16>>::ops(), Launch<GermanLane<4, 16, kGermanViBlock / 64, true>>::vi_only()),
      Launch<GermanLane<8, 8>>::ops(), Launch<GermanLane<16, 4>>::ops(),
  };
  return t;
}
}  // namespace arp
