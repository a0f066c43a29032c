"""CPU-only: placement of a rank's lattice workers on its GPU's NUMA node (jlm_amd/numa.py) against a fake sysfs tree."""
import os

from jlm_amd import numa


def _sysfs(tmp_path):
    root = str(tmp_path)
    for node, cpus in ((0, "0-7,16-23"), (1, "8-15,24-31")):
        d = os.path.join(root, "devices", "system", "node", "node%d" % node)
        os.makedirs(d)
        with open(os.path.join(d, "cpulist"), "w") as f:
            f.write(cpus + "\n")
    for bus, node in (("0000:05:00.0", 0), ("0000:c5:00.0", 1), ("0000:e5:00.0", -1)):
        d = os.path.join(root, "bus", "pci", "devices", bus)
        os.makedirs(d)
        with open(os.path.join(d, "numa_node"), "w") as f:
            f.write("%d\n" % node)
    return root


def test_cpulist_and_worker_cpus(tmp_path, monkeypatch):
    monkeypatch.delenv("JLM_NUMA_PIN", raising=False)
    assert numa.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    root = _sysfs(tmp_path)
    allowed = set(range(4, 28))                       # the process's own affinity mask
    assert numa.worker_cpus(0, sysfs=root, pci_bus_id="0000:05:00.0", allowed=allowed) == (0, {4, 5, 6, 7, 16, 17, 18, 19, 20, 21, 22, 23})
    assert numa.worker_cpus(1, sysfs=root, pci_bus_id="0000:C5:00.0", allowed=allowed) == (1, set(range(8, 16)) | {24, 25, 26, 27})
    assert numa.worker_cpus(2, sysfs=root, pci_bus_id="0000:e5:00.0", allowed=allowed) == (-1, set())      # the kernel does not know
    assert numa.worker_cpus(3, sysfs=root, pci_bus_id="0000:ff:00.0", allowed=allowed) == (-1, set())      # no such device
    # a process mask that only grazes the GPU's node: pinning would crowd the workers onto a CPU or two -- left unpinned
    assert numa.worker_cpus(0, sysfs=root, pci_bus_id="0000:05:00.0", allowed={4, 5, 8, 9, 10, 11}) == (0, set())
    assert numa.worker_cpus(0, sysfs=root, pci_bus_id="0000:05:00.0", allowed={4, 5, 8, 9, 10, 11}, min_cpus=2) == (0, {4, 5})
    monkeypatch.setenv("JLM_NUMA_PIN", "0")
    assert numa.worker_cpus(0, sysfs=root, pci_bus_id="0000:05:00.0", allowed=allowed) == (-1, set())


def test_pin_touches_the_calling_thread_only():
    import threading
    if not hasattr(os, "sched_getaffinity"):
        return
    before = os.sched_getaffinity(0)
    one = {min(before)}
    seen = {}

    def worker():
        seen["ok"] = numa.pin_current_thread(one)
        seen["mask"] = os.sched_getaffinity(0)
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen["ok"] and seen["mask"] == one
    assert os.sched_getaffinity(0) == before          # the main thread keeps its mask
    assert numa.pin_current_thread(set()) is False
