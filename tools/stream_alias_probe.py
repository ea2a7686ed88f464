"""Development probe: which HIP streams share a hardware queue?  A long kernel on stream i, then a tiny kernel on stream j: if j's kernel
finishes only after i's, the two streams are multiplexed on one hardware queue (HIP maps its streams onto GPU_MAX_HW_QUEUES queues)."""
import os
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n)]
x = torch.zeros(1 << 20, device=dev)
y = [torch.zeros(16, device=dev) for _ in streams]
torch.cuda.synchronize()
LONG = 20_000_000  # ~10 ms at 2 GHz


def aliased(i, j):
    torch.cuda.synchronize()
    e_long = torch.cuda.Event(enable_timing=True)
    e_short = torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(streams[i]):
        e0.record()
        torch.cuda._sleep(LONG)
        e_long.record()
    with torch.cuda.stream(streams[j]):
        y[j].add_(1.0)
        e_short.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e_short) > 0.5 * e0.elapsed_time(e_long)


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("stream 0 = the default (null) stream; 1.. = torch.cuda.Stream() in creation order")
for i in range(len(streams)):
    print("long on %2d: blocked ->" % i, [j for j in range(len(streams)) if j != i and aliased(i, j)])
