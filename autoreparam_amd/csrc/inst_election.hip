// Chain-kernel instantiations for election88: S+1 = 52 groups (51 state effects
// plus the cells that see no state effect).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& election_ops() {
  static const std::vector<LaneOps> t = {
      election_lane_ops<4, 13>(), election_lane_ops<8, 7>(),
      election_lane_ops<16, 4>(),
  };
  return t;
}
}  // namespace arp
