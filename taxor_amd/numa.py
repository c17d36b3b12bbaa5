"""CPU / memory locality of a rank's GPU, found WITHOUT touching the GPU (sysfs only), so that a rank can bind its host
threads before its first HIP call: with eight ranks feeding eight GPUs from host buffers (the drop-in call,
taxor_gpu_search_batch), a staging buffer first-touched on the other socket crosses the inter-socket fabric on every
H2D copy, and the parser threads of eight ranks pile onto the same cores.

The amdgpu/KFD driver lists compute nodes under /sys/class/kfd/kfd/topology/nodes/<n>/properties; nodes with
simd_count > 0 are GPUs, in the order HIP enumerates them (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES select and reorder
by ordinal).  `domain` + `location_id` give the PCI address, whose sysfs directory carries numa_node and local_cpulist."""
import os

KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"
PCI_DEVICES = "/sys/bus/pci/devices"


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11}"""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def _props(path):
    out = {}
    try:
        with open(path) as f:
            for line in f:
                kv = line.split()
                if len(kv) == 2:
                    try:
                        out[kv[0]] = int(kv[1])
                    except ValueError:
                        pass
    except OSError:
        pass
    return out


def kfd_gpus(kfd_nodes=KFD_NODES):
    """PCI addresses ('dddd:bb:dd.f') of the GPU nodes in KFD order."""
    gpus = []
    try:
        nodes = sorted((int(n) for n in os.listdir(kfd_nodes) if n.isdigit()))
    except OSError:
        return gpus
    for n in nodes:
        p = _props(os.path.join(kfd_nodes, str(n), "properties"))
        if p.get("simd_count", 0) <= 0:
            continue                                   # a CPU node
        loc = p.get("location_id", 0)
        gpus.append("%04x:%02x:%02x.%x" % (p.get("domain", 0) & 0xFFFF, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return gpus


def visible_ordinals(n_gpus, env=None):
    """device ordinal -> KFD GPU index after ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES (integer lists only; UUID
    lists are left alone and the identity map is returned)"""
    env = os.environ if env is None else env
    order = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if not v:
            continue
        try:
            sel = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return list(range(n_gpus))
        order = [order[i] for i in sel if 0 <= i < len(order)]
    return order


def gpu_locality(device, kfd_nodes=KFD_NODES, pci_devices=PCI_DEVICES, env=None):
    """(numa_node, cpu set) of HIP device `device`; (None, None) when the platform does not say (no KFD, a VM without
    NUMA information, numa_node == -1)."""
    gpus = kfd_gpus(kfd_nodes)
    order = visible_ordinals(len(gpus), env)
    if device < 0 or device >= len(order):
        return None, None
    d = os.path.join(pci_devices, gpus[order[device]])
    try:
        with open(os.path.join(d, "numa_node")) as f:
            node = int(f.read().strip())
        with open(os.path.join(d, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None, None
    if node < 0 or not cpus:
        return None, None
    return node, cpus


def bind_to_gpu(device, **kw):
    """Restrict this process (and every thread it starts later) to the CPUs local to HIP device `device`; ranks whose GPUs
    hang off the same NUMA node share its cores and the kernel balances them.  Memory follows by first touch.  Must run
    before the first HIP call (no re-exec, no numactl).  Returns a dict describing what was done, for the bench line."""
    node, cpus = gpu_locality(device, **kw)
    if node is None:
        return {"bound": False, "reason": "no NUMA locality information for this device in sysfs"}
    try:
        allowed = os.sched_getaffinity(0)
        use = cpus & allowed
        if not use:
            return {"bound": False, "numa_node": node, "reason": "the GPU's local CPUs are outside this process's affinity mask"}
        os.sched_setaffinity(0, use)
    except (OSError, AttributeError) as e:
        return {"bound": False, "numa_node": node, "reason": f"sched_setaffinity: {e}"}
    return {"bound": True, "numa_node": node, "cpus": len(use)}
