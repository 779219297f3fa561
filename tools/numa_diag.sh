lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)"
for d in /sys/class/drm/renderD*; do echo $d $(cat $d/device/numa_node) $(cat $d/device/local_cpulist); done
for n in /sys/class/kfd/kfd/topology/nodes/*; do echo $n $(grep -E "simd_count|drm_render_minor" $n/properties | tr '\n' ' '); done
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], '...')"
cat /sys/devices/system/node/node*/cpulist
cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null
