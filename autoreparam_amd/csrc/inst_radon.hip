// Radon launcher table: the per-K instantiations live in inst_radon_k{4,8,16}.hip so
// that they compile in parallel.
#include "host_common.h"

namespace arp {
std::vector<LaneOps> radon_ops_k4();
std::vector<LaneOps> radon_ops_k8();
std::vector<LaneOps> radon_ops_k16();

const std::vector<LaneOps>& radon_ops() {
  static const std::vector<LaneOps> t = [] {
    std::vector<LaneOps> v = radon_ops_k4();
    for (auto f : {radon_ops_k8, radon_ops_k16})
      for (const auto& o : f()) v.push_back(o);
    return v;
  }();
  return t;
}
}  // namespace arp
