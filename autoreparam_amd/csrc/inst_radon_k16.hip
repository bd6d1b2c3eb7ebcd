// Chain-kernel instantiations for the radon model, 16 lanes per chain: per-lane slice
// sizes NL = ceil(J / K) for the county counts of the reference's radon datasets
// (MN 85, PA 68, IN 91, MO 115, ND 53, MA 13, AZ 15) plus round-ups.
#include "host_common.h"

namespace arp {
std::vector<LaneOps> radon_ops_k16() {
  return {radon_lane_ops<16, 1>(), radon_lane_ops<16, 4>(), radon_lane_ops<16, 5>(), radon_lane_ops<16, 6>(), radon_lane_ops<16, 8>()};
}
}  // namespace arp
