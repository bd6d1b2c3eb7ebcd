// Chain-kernel instantiations for the radon model, 8 lanes per chain: per-lane slice
// sizes NL = ceil(J / K) for the county counts of the reference's radon datasets
// (MN 85, PA 68, IN 91, MO 115, ND 53) plus round-ups.
#include "host_common.h"

namespace arp {
std::vector<LaneOps> radon_ops_k8() {
  return {Launch<RadonLane<8, 7>>::ops(), Launch<RadonLane<8, 9>>::ops(), Launch<RadonLane<8, 11>>::ops(), Launch<RadonLane<8, 12>>::ops(), Launch<RadonLane<8, 15>>::ops()};
}
}  // namespace arp
