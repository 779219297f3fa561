#!/bin/bash
# multi-radius ball query (PCL_MULTI_BALL_QUERY=0 | 1): its parity test, the MSG networks' tests, then cfg4 interleaved on one box
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ball_query" 2>&1 | tail -3
python -m pytest tests/test_networks_gpu.py -m gpu -x -q -k "msg or MSG" 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 1; do
    PCL_MULTI_BALL_QUERY=$v python tools/bench_models.py --steps 40 --only "cfg4 PointNet++ MSG part-seg B=16 N=2048 (BASELINE" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print('cfg4 rep $rep [multi=$v]', j.get('ms_per_step'), j.get('ms_per_step_inline'))"
  done
done
