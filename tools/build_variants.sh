#!/bin/bash
# timing experiments only: tagged library variants built with extra -D flags (see the ARP_EXP_* hooks in csrc/)
# usage: tools/build_variants.sh tag1=-DFLAG1 tag2="-DFLAG2 -DFLAG3" ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  tag="${spec%%=*}"; flags="${spec#*=}"
  ARP_BUILD_TAG="_$tag" ARP_HIPCC_FLAGS="$flags" python3 -c "from autoreparam_amd import build; print(build.build())" &
done
wait
