"""Host <-> device copy rates of this box (pinned and pageable), and the host's memcpy rate: what bounds a STREAM of windows (BASELINE config 4
taken literally: every window's five arrays arrive in host memory, reference src/slam.cpp:899-921)."""
import json
import time

import numpy as np
import torch


def rate(fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    out = {}
    for mb in (64, 512):
        n = mb << 20
        dev = torch.empty(n, dtype=torch.uint8, device="cuda")
        pin = torch.empty(n, dtype=torch.uint8).pin_memory()
        pag = torch.empty(n, dtype=torch.uint8)
        pag.fill_(1); pin.fill_(2)
        out["%dMB" % mb] = {
            "h2d_pinned_GBs": rate(lambda: dev.copy_(pin, non_blocking=True), n),
            "h2d_pageable_GBs": rate(lambda: dev.copy_(pag), n),
            "d2h_pinned_GBs": rate(lambda: pin.copy_(dev, non_blocking=True), n),
            "host_memcpy_pageable_to_pinned_GBs": rate(lambda: pin.copy_(pag), n),
        }
    a = np.ones(64 << 20, dtype=np.uint8); b = np.empty_like(a)
    t0 = time.perf_counter()
    for _ in range(5):
        np.copyto(b, a)
    out["numpy_memcpy_1_thread_GBs"] = 5 * a.nbytes / (time.perf_counter() - t0) / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
