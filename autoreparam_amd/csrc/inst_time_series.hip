// Chain-kernel instantiations for the local linear trend model: 2T = 120 trend latents in
// consecutive runs of 30 (15 time steps) over the 4 lanes of a chain.
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& time_series_ops() {
  static const std::vector<LaneOps> t = {
      Launch<TimeSeriesLane<4, 30>>::ops(),
  };
  return t;
}
}  // namespace arp
