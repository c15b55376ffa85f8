"""The package's side streams: ONE small pool per device, shared by everything that forks work off the caller's stream
(the two half-circuit chains of a cut contraction, the slices of a sliced contraction in flight side by side, hipGraph
captures).

Why a pool and not a ``torch.cuda.Stream()`` wherever one is needed: a HIP stream gets its hardware queue with its first
work, queues are handed out round-robin over FOUR per process, and two streams that share a queue run one after the other.
Measured in round 6 (scripts/round6/gpu_r6o.sh): the sliced value_and_grad with four slices in flight takes 7.8 ms per call
-- and 10.9 ms when two OTHER streams had been used earlier in the process (those of a second cut circuit: bench.py's
HEA-A leg), because one of its side streams then shares the queue of the stream it forks from; with four earlier streams
the assignment has wrapped around and the time is 7.7 ms again.  With one pool, created and touched in one go the first
time any of it is asked for, the package's streams sit on consecutive queues and their number does not grow with the number
of compiled circuits and contraction trees.
"""

from typing import Dict, List

POOL_SIZE = 3      # the caller's stream + 3 = the four hardware queues of a process
_POOLS: Dict[int, List] = {}


def side_stream(device, i: int = 0):
    """Side stream ``i`` (modulo the pool size) of ``device``."""
    import torch

    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _POOLS.get(idx)
    if pool is None:
        pool = [torch.cuda.Stream(device=idx) for _ in range(POOL_SIZE)]
        for s_ in pool:                       # first work = the stream's hardware queue: all of them now, in order
            with torch.cuda.stream(s_):
                torch.zeros(1, device=f"cuda:{idx}").add_(1)
        _POOLS[idx] = pool
    return pool[i % POOL_SIZE]
