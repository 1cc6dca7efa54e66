"""The streamed leg alone (bench.py::streamed_block's loop) with per-call host timings: where a batch period goes - submit (pack + layout +
enqueue), waiting in collect, copying out - batch by batch.  Run under `rocprofv3 --kernel-trace --memory-copy-trace` to see whether
uploads, solves and downloads of consecutive batches overlap on the device."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=1024)
    ap.add_argument("--lines", type=int, default=2000)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--mode", default="pinned", help="pinned: page-locked arrays, zero-copy ingest + device build; pageable: staging copy + device build; host: the host packer")
    args = ap.parse_args()
    B = args.windows
    windows = [synth.make_window(i, num_lines=args.lines) for i in range(B)]
    warm = 2 * args.depth
    nsets = warm + args.batches
    base = capi.WindowSet(windows, pinned=(args.mode in ("pinned", "packed")), packed=(args.mode == "packed"))
    sets = [base.derive(list(range((k * 37) % B, B)) + list(range((k * 37) % B))) for k in range(nsets)]
    st = capi.LBAStream(depth=args.depth, host_threads=args.host_threads, **({"device_build": -1} if args.mode == "host" else {}))
    tick = []
    for k in range(warm):                # every slot built once and refilled once (a slot's first refill allocates its staging blocks)
        if k >= args.depth:
            st.collect(tick[k - args.depth], want_summaries=False)
        tick.append(st.submit(sets[k]))
    for k in range(warm - args.depth, warm):
        st.collect(tick[k], want_summaries=False)
    rows, tick = [], []
    t00 = time.perf_counter()
    for k in range(args.batches):
        tw = 0.0
        if k >= args.depth:
            a = time.perf_counter(); st.collect(tick[k - args.depth], want_summaries=False); tw = time.perf_counter() - a
        a = time.perf_counter(); tick.append(st.submit(sets[warm + k])); ts = time.perf_counter() - a
        rows.append((k, 1e3 * tw, 1e3 * ts))
    for k in range(max(0, args.batches - args.depth), args.batches):
        a = time.perf_counter(); st.collect(tick[k], want_summaries=False); rows.append((k, 1e3 * (time.perf_counter() - a), 0.0))
    dt = time.perf_counter() - t00
    for r in rows:
        print("batch %2d  collect %.2f ms  submit %.2f ms" % r)
    periods = sorted(r[1] + r[2] for r in rows[args.depth:args.batches])
    print(json.dumps({"ms_per_batch": 1e3 * dt / args.batches, "steady_ms_per_batch": periods[len(periods) // 2] if periods else None, "stats": st.stats(), "build_stats": st.build_stats()}))
    st.close()


if __name__ == "__main__":
    main()
