#!/bin/bash
# HBM traffic (PMC) + kernel stats of one tools/bench_models.py config (run through gpurun; copy the results to profiles/ by hand):
#   bash tools/traffic_cfg.sh r04 cfg3 "cfg3"          # <round tag> <key: cfg2_n4096|cfg3|cfg4|cfg5> <--only substring>
# Two separate counter passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only beside them) and one --kernel-trace --stats pass.
R=${1:-r04}; K=${2:-cfg3}; SUB=${3:-cfg3}; O=gpurun_out/$R; mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python tools/bench_models.py --steps 3 --only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$K -o b -- $CMD "$SUB" --dump-launch-order $O/order_$K.json > $O/f_$K.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w_$K -o b -- $CMD "$SUB" --dump-launch-order $O/order2_$K.json > $O/w_$K.log 2>&1
F=$(find $O/f_$K -name '*counter_collection.csv' | head -1); W=$(find $O/w_$K -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W $O/order_$K.json $O/${R}_pmc_hbm_traffic_$K.csv $O/${R}_traffic_$K.json "tools/bench_models.py --only '$SUB'" "$CMD '$SUB'" 2>&1 | tee $O/traffic_$K.txt | head -30
sed -i "s#\"source\": \"$O/#\"source\": \"profiles/#" $O/${R}_traffic_$K.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$K -o b -- python tools/bench_models.py --steps 10 --only "$SUB" > $O/kt_$K.log 2>&1
cp $(find $O/kt_$K -name '*kernel_stats.csv' | head -1) $O/${R}_${K}_kernel_stats.csv
grep '"config"' $O/kt_$K.log | cut -c1-300
rm -rf $O/f_$K $O/w_$K $O/kt_$K
