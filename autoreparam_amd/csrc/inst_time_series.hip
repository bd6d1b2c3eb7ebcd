// Chain-kernel instantiations for the local linear trend model: 2T = 120 trend latents in consecutive runs of
// 16 / 30 / 8 (8 / 15 / 4 time steps) over the 8 / 4 / 16 lanes of a chain (T = 60 padded to 64 at 8 and 16 lanes).
#include "host_common.h"

namespace arp {
const std::vector<LaneOps>& time_series_ops() {
  static const std::vector<LaneOps> t = {
      Launch<TimeSeriesLane<4, 30>>::ops(),
      Launch<TimeSeriesLane<8, 16>>::ops(),
      Launch<TimeSeriesLane<16, 8>>::ops(),
  };
  return t;
}
}  // namespace arp
