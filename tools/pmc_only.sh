#!/bin/bash
# only the HBM-traffic PMC passes of tools/evidence.sh:  bash tools/pmc_only.sh r03
R=${1:-r03}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="python bench.py --steps 3 --warmup 1 --no-settle --no-cpu-baseline --no-other-configs --roofline-kernel none"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o b -- $S --dump-launch-order $O/order.json > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o b -- $S > $O/write.log 2>&1
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W $O/order.json $O/${R}_pmc_hbm_traffic.csv $O/${R}_traffic.json 2>&1 | tee $O/traffic.txt
rm -rf $O/fetch $O/write
