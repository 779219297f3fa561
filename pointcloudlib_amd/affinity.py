"""Keep a rank's host thread on the CPU socket of its GPU.

One process per GPU enqueues ~140 small launches per step; on the dual-socket MI355X hosts (2 NUMA nodes, GPUs split 4+4)
an unpinned process that the scheduler moves between sockets enqueues ~15 % slower and the step turns host-bound
(measured: 2.9-3.1 ms instead of 2.57 ms for the first process starts on a fresh box; pinned to one node: 8 of 8 starts
at 2.57 ms).  `pin_to_gpu_node()` restricts the CPU affinity (threads created later inherit it; first-touch keeps host
memory local) to the NUMA node of the rank's GPU, found through the KFD topology in sysfs -- no torch, no HIP call, so
it can run before anything else is imported.  It never raises: on a host without the sysfs entries it does nothing.
"""
import os

_KFD = "/sys/class/kfd/kfd/topology/nodes"


def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.update(range(int(a), int(b or a) + 1))
    return out


def _node_cpus(node):
    with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
        return _cpulist(f.read())


def gpu_numa_nodes():
    """NUMA node of every GPU this process can open, in HIP enumeration order (KFD topology order)."""
    nodes = []
    try:
        entries = sorted(os.listdir(_KFD), key=int)
    except (OSError, ValueError):
        return nodes
    for n in entries:
        try:
            with open(f"{_KFD}/{n}/properties") as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", 0)) == 0:
                continue                                                  # a CPU node
            with open(f"/sys/class/drm/renderD{int(props['drm_render_minor'])}/device/numa_node") as f:
                nodes.append(int(f.read()))
        except (OSError, ValueError, KeyError):
            continue                                                      # another tenant's GPU: not readable
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        try:
            nodes = [nodes[int(v)] for v in vis.split(",")]
        except (ValueError, IndexError):
            pass
    return nodes


def pin_to_gpu_node(local_rank=0):
    """Restrict this process to the CPUs of GPU `local_rank`'s NUMA node (or, if that cannot be found, of the node it is
    running on now).  Returns a short description for logs, or None if nothing was changed."""
    try:
        allowed = os.sched_getaffinity(0)
        gpus = gpu_numa_nodes()
        node, why = None, ""
        forced = os.environ.get("PCL_PIN_NODE")              # lab switch (tools/driver_dist.sh): "-1" = do not pin, "n" = pin to node n
        if forced is not None and forced.lstrip("-").isdigit():
            if int(forced) < 0:
                return None
            node, why = int(forced), "PCL_PIN_NODE"
        elif local_rank < len(gpus) and gpus[local_rank] >= 0:
            node, why = gpus[local_rank], f"gpu{local_rank}"
        else:
            with open("/proc/self/stat") as f:
                cur = int(f.read().rsplit(")", 1)[1].split()[36])          # field 39: CPU last run on
            for n in sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()):
                if cur in _node_cpus(n):
                    node, why = n, "current cpu"
                    break
        if node is None:
            return None
        cpus = _node_cpus(node) & allowed
        if not cpus or cpus == allowed:
            return None
        os.sched_setaffinity(0, cpus)
        return f"numa node {node} ({why}), {len(cpus)} cpus"
    except (OSError, ValueError, IndexError):
        return None
