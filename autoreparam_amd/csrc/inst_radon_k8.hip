// Chain-kernel instantiations for the radon model, 8 lanes per chain: per-lane slice
// sizes NL = ceil(J / K) for the county counts of the reference's radon datasets
// (MN 85, PA 68, IN 91, MO 115, ND 53, MA 13, AZ 15) plus round-ups.
#include "host_common.h"

namespace arp {
std::vector<LaneOps> radon_ops_k8() {
  return {radon_lane_ops<8, 2>(), radon_lane_ops<8, 7>(), radon_lane_ops<8, 9>(), radon_lane_ops<8, 11>(), radon_lane_ops<8, 12>(), radon_lane_ops<8, 15>()};
}
}  // namespace arp
