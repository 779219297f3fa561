#!/bin/bash
# Is the host on the critical path at the forward -> backward turn?  A busy-wait between the loss and loss.backward() (bench.py
# PCL_HOST_DELAY_AT=bwd) against the same wait at the head of the step; interleaved on one box.
for rep in 1 2; do
  for v in "PCL_HOST_DELAY_US=0" "PCL_HOST_DELAY_US=60 PCL_HOST_DELAY_AT=bwd" "PCL_HOST_DELAY_US=150 PCL_HOST_DELAY_AT=bwd" "PCL_HOST_DELAY_US=400 PCL_HOST_DELAY_AT=bwd" "PCL_HOST_DELAY_US=400 PCL_HOST_DELAY_AT=step"; do
    r=$(env $v python bench.py --no-cpu-baseline --no-other-configs --roofline-kernel none --steps 60 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j.get('host_enqueue_ms'))")
    echo "rep $rep [$v] $r"
  done
done
