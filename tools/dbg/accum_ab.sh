#!/bin/bash
# A/B of the fp64-flushed forward accumulation (csrc/frag.hip) on configs 3 and 4: parity rows and step time, same box.
O=gpurun_out/accum; mkdir -p $O
python -m pytest tests/test_frag_hip.py -m gpu -q -s 2>&1 | tail -4 > $O/frag_tests.txt
for v in msg ssg; do python tools/dbg/partseg_local_err.py $v > $O/local_$v.txt 2>&1; done
python -m pytest tests/test_parity_partseg_gpu.py tests/test_parity_dgcnn_gpu.py -m gpu -s -q 2>&1 | grep -v Warning > $O/parity_on.txt
PCL_PARTSEG_FLUSH=0 PCL_DGCNN_FLUSH=0 python -m pytest tests/test_parity_dgcnn_gpu.py -m gpu -s -q 2>&1 | grep -v Warning > $O/parity_off_dgcnn.txt
for f in 0 1; do
  if [ $f = 0 ]; then export PCL_PARTSEG_FLUSH=0 PCL_DGCNN_FLUSH=0; else unset PCL_PARTSEG_FLUSH PCL_DGCNN_FLUSH; fi
  python tools/bench_models.py --steps 30 --only "cfg3" 2>&1 | grep -E "ms/step|cfg3" | tail -2 > $O/time_cfg3_$f.txt
  python tools/bench_models.py --steps 30 --only "cfg4 PointNet++ MSG part-seg B=16 N=2048 (BASELINE" 2>&1 | grep -E "ms/step|cfg4" | tail -2 > $O/time_cfg4_$f.txt
done
tail -3 $O/frag_tests.txt; cat $O/local_msg.txt | tail -11; grep -E "fp3|fp2|fp1|logits|gradients:|passed|failed|closer" $O/parity_on.txt; grep -E "gradients:|passed|failed" $O/parity_off_dgcnn.txt; cat $O/time_*.txt
