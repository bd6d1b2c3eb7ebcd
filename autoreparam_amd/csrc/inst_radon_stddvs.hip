// Chain-kernel instantiations for radon_stddvs: lanes-per-chain K and counties per lane
// NLS = ceil(J / K) for the reference's radon datasets (MN 85, PA 68, IN 91, MO 115, ND 53, MA 13, AZ 15).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& radon_sd_ops() {
  static const std::vector<LaneOps> t = {
#define R(K, N) Launch<RadonSdLane<K, N>>::ops()
      R(16, 1), R(8, 2),   // MA 13, AZ 15 counties
      R(16, 4), R(16, 5), R(16, 6), R(16, 8),
      R(8, 7), R(8, 9), R(8, 11), R(8, 12), R(8, 15),
#undef R
  };
  return t;
}
}  // namespace arp
