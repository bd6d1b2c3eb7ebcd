// Chain-kernel instantiations for election88: S+1 = 52 groups (51 state effects
// plus the cells that see no state effect).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& election_ops() {
  static const std::vector<LaneOps> t = {
      Launch<ElectionLane<4, 13>>::ops(), Launch<ElectionLane<8, 7>>::ops(),
      Launch<ElectionLane<16, 4>>::ops(),
  };
  return t;
}
}  // namespace arp
