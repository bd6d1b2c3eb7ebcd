// Chain-kernel instantiations for the electric company model: P+1 = 97 groups (96 pair
// effects plus the observations that see no pair effect).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& electric_ops() {
  static const std::vector<LaneOps> t = {
      Launch<ElectricLane<16, 7>>::ops(),
  };
  return t;
}
}  // namespace arp
