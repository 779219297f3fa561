#!/bin/bash
# run-to-run spread of the MSG part-seg step with and without the sampling prefetch (three fresh processes), then the bench line's own rows
for i in 1 2 3; do
  python tools/bench_models.py --steps 20 --only "cfg4 PointNet++ MSG" 2>&1 | grep -o '"config": "[^"]*", "ms_per_step": [0-9.]*' | cut -c1-170
done
python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['ms_per_step'], [(r['key'], r['ms_per_step'], r.get('ms_per_step_inline'), r['roofline'].get('traffic')) for r in d['other_configs']], d['roofline']['traffic'])"
