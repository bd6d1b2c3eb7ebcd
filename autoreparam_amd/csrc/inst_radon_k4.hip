// Chain-kernel instantiations for the radon model, 4 lanes per chain: per-lane slice
// sizes NL = ceil(J / K) for the county counts of the reference's radon datasets
// (MN 85, PA 68, IN 91, MO 115, ND 53, MA 13, AZ 15) plus round-ups.
#include "host_common.h"

namespace arp {
std::vector<LaneOps> radon_ops_k4() {
  return {radon_lane_ops<4, 4>(), radon_lane_ops<4, 14>(), radon_lane_ops<4, 17>(), radon_lane_ops<4, 22>(), radon_lane_ops<4, 23>(), radon_lane_ops<4, 29>()};
}
}  // namespace arp
