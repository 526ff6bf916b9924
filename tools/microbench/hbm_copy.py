#!/usr/bin/env python3
"""What HBM delivers on this box next to the 8 TB/s the C4 roofline is priced against (SURVEY.md §8d asks
for a stream-copy figure): device-to-device copy, read-only reduction and fill of 4 GiB, best of 10."""
import json

import torch

n = 1 << 30  # float32 elements = 4 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)


def best_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


gib = a.numel() * 4
out = {"bytes": gib,
       "copy_TBps_read_plus_write": 2 * gib / best_ms(lambda: b.copy_(a)) / 1e9,
       "read_TBps_sum": gib / best_ms(lambda: a.sum()) / 1e9,
       "write_TBps_fill": gib / best_ms(lambda: b.fill_(1.0)) / 1e9}
print(json.dumps(out))
