// Chain-kernel instantiations for the electric company model: P+1 = 97 groups (96 pair
// effects plus the observations that see no pair effect), over 16 or 8 lanes per chain.
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& electric_ops() {
  static const std::vector<LaneOps> t = {
      Launch<ElectricLane<16, 7>>::ops(),
      Launch<ElectricLane<8, 13>>::ops(),
  };
  return t;
}
}  // namespace arp
