"""MI355X-native LSTM inference + lattice beam-search decoders behind the JLM reference's API."""
import os
import sys


def _want_hw_queues(n=8):
    """The decode engine keeps four HIP streams busy (four batches in flight; JLM_STREAMS=2: two, each with a
    side stream for its edge logits).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4,
    one of them the null stream's); streams that share a queue serialise, and with 4 queues a side
    stream lands on the other batch's queue -- the two-stream overlap is gone (3.8 instead of 2.8 ms per
    step, tools/probes/streams_probe.py; three streams on 4 queues: 2.28 vs 2.23 ms device-resident).  The variable is read when the HIP runtime initialises, so it
    can only be defaulted here if that has not happened yet; the engine asks hw_queues_ok() and falls
    back to running the edge logits on the batch's own stream."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is None and os.environ.get("JLM_NO_ENV_DEFAULTS", "0") == "1":
        return False          # the host process decides its own environment: nothing is written
    if cur is not None:
        try:
            return int(cur) >= n
        except ValueError:
            return False
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return True


_HW_QUEUES_OK = _want_hw_queues()


def hw_queues_ok():
    return _HW_QUEUES_OK


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    show 256 hardware threads behind a 16-CPU quota; running 64 lattice threads there gets the whole
    process throttled for the rest of the scheduling period, the enqueueing thread included)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                                          # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    # one process per GPU (torchrun): the ranks of a node share the quota
    try:
        n //= max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        pass
    return max(1, n)
