// Instantiations of the chain kernels for the radon model: lanes-per-chain K and
// per-lane slice size NL = ceil(J / K) for the county counts of the reference's
// radon datasets (MN 85, PA 68, IN 91, MO 115, ND 53) plus round-ups.
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& radon_ops() {
  static const std::vector<LaneOps> t = {
#define R(K, NL) Launch<RadonLane<K, NL>>::ops()
      R(16, 4), R(16, 5), R(16, 6), R(16, 8),
      R(8, 7), R(8, 9), R(8, 11), R(8, 12), R(8, 15),
      R(4, 14), R(4, 17), R(4, 22), R(4, 23), R(4, 29),
#undef R
  };
  return t;
}
}  // namespace arp
