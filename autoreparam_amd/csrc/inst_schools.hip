// Chain-kernel instantiations for eight schools (8 groups) and election88 (S+1 = 52 groups).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& schools_ops() {
  static const std::vector<LaneOps> t = {
      Launch<SchoolsLane<1, 8>>::ops(), Launch<SchoolsLane<2, 4>>::ops(),
      Launch<SchoolsLane<4, 2>>::ops(), Launch<SchoolsLane<8, 1>>::ops(),
  };
  return t;
}
const std::vector<LaneOps>& funnel_ops() {
  static const std::vector<LaneOps> t = {Launch<FunnelLane<1, 1>>::ops()};
  return t;
}
}  // namespace arp
