"""Host-side placement of a rank's worker threads (one process per GPU, sentences sharded, no data-path collective).

On an 8-GPU node every rank builds its lattices on a few worker threads (jlm_amd/decoder.py ``_prefetched``) and copies them to ITS
GPU from page-locked blocks.  The GPUs hang off different sockets / NUMA nodes: a worker running on the far node builds the arrays in
far memory and the copy crosses the socket link.  ``worker_cpus(device_index)`` is the set of CPUs of the GPU's own NUMA node that
this process may use, and the lattice workers pin themselves to it when they start (``pin_current_thread``: the calling THREAD only --
``sched_setaffinity(0, ...)`` takes a thread id on Linux -- so the host application's own threads keep their affinity).
``JLM_NUMA_PIN=0`` switches it off; it is a no-op wherever sysfs does not say (single-node boxes report node -1 or one node).
"""
import os

SYSFS = "/sys"


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    out = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.update(range(int(a), int(b) + 1))
        else:
            out.add(int(part))
    return out


def pci_numa_node(pci_bus_id, sysfs=SYSFS):
    """NUMA node of a PCI device ('0000:c1:00.0'), or -1 when the kernel does not know"""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def node_cpus(node, sysfs=SYSFS):
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return set()


def device_pci_bus_id(device_index):
    """'dddd:bb:dd.f' of a visible GPU, or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or dev is None:
            return None
        return "%04x:%02x:%02x.0" % (dom, bus, dev)
    except Exception:
        return None


MIN_PIN_CPUS = 4          # fewer CPUs of the GPU's node than this in the process mask: pinning would crowd the workers, leave them alone


def worker_cpus(device_index, sysfs=SYSFS, pci_bus_id=None, allowed=None, min_cpus=None):
    """(numa node, CPUs of that node this process may use) for the GPU; (-1, set()) when there is nothing to pin to.
    A process mask that only grazes the node (fewer than ``min_cpus`` of its CPUs, default MIN_PIN_CPUS or the worker count the
    caller passes) gives (node, set()): the workers then run unpinned rather than all on one or two CPUs."""
    if os.environ.get("JLM_NUMA_PIN", "1") == "0":
        return -1, set()
    bus = pci_bus_id or device_pci_bus_id(device_index)
    if not bus:
        return -1, set()
    node = pci_numa_node(bus, sysfs)
    if node < 0:
        return -1, set()
    if allowed is None:
        try:
            allowed = os.sched_getaffinity(0)
        except AttributeError:
            return node, set()
    cpus = node_cpus(node, sysfs) & set(allowed)
    need = MIN_PIN_CPUS if min_cpus is None else max(int(min_cpus), 1)
    if len(cpus) < need:
        return node, set()
    return node, cpus


def pin_current_thread(cpus):
    """Restrict the CALLING thread to `cpus` (no-op for an empty set or where the platform has no affinity call)"""
    if not cpus:
        return False
    try:
        os.sched_setaffinity(0, cpus)
        return True
    except (AttributeError, OSError):
        return False
