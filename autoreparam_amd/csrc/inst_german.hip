// Chain-kernel instantiations for german_credit_lognormalcentered: the padded
// 64-column design matrix is split over K lanes x 64/K consecutive features.
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& german_ops() {
  static const std::vector<LaneOps> t = {
      Launch<GermanLane<4, 16>>::ops(), Launch<GermanLane<8, 8>>::ops(), Launch<GermanLane<16, 4>>::ops(),
  };
  return t;
}
}  // namespace arp
