"""stdin: bench.py output -> 'value ms_per_step' of the JSON line (helper for A/B loops on the GPU box)."""
import json, sys
for line in sys.stdin:
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(d["value"], d["ms_per_step"], *(sys.argv[1:]))
