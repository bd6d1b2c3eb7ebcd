export ARP_DEBUG=1 VI_BENCH_KINDS=1
for lr in 1 2; do for R in 1 2 4 8; do echo "== german lrs=$lr R=$R"; VI_BENCH_LRS=$lr ARP_VI_R=$R timeout 120 python tools/vi_bench.py german 2>&1 | grep -v DEBUG | grep german; done; done
for G in 4 8 16 32; do echo "== radon_PA G=$G"; ARP_VI_G=$G timeout 120 python tools/vi_bench.py radon_PA 2>&1 | grep "^radon"; done
for G in 4 8 16 32; do echo "== election G=$G"; ARP_VI_G=$G timeout 120 python tools/vi_bench.py election 2>&1 | grep "^election"; done
for lr in 1 2; do echo "== radon lrs=$lr"; VI_BENCH_LRS=$lr timeout 120 python tools/vi_bench.py radon_PA funnel 2>&1 | grep "NCP"; done
