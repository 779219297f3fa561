"""Run a command with the CPU affinity set to one NUMA node:  python tools/pinrun.py <node> <cmd...>"""
import os, sys
def cpus(node):
    out = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-"); out.update(range(int(a), int(b or a) + 1))
    return out
os.sched_setaffinity(0, cpus(int(sys.argv[1])) & os.sched_getaffinity(0))
os.execvp(sys.argv[2], sys.argv[2:])
