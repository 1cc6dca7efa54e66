"""tests/golden/bench_digest.json: crc32 of the solved parameters of the windows bench.py's cross-rank check looks at - the first four
windows of every shard of a 1 / 2 / 4 / 8-rank run (window id = rank * windows_per_gpu + i) - each solved ALONE on one GPU with the sweep
and chunk cut the headline batch takes.  A window's bytes are a function of the window, the sweep and the cut, so an N-rank run of
bench.py can be held against this 1-rank record without re-solving anything.  Regenerate after any change to
the sweeps' arithmetic:   python tools/make_bench_digest.py   (needs the GPU)."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slslam_amd import capi, synth  # noqa: E402


def main():
    lines, per_gpu, k = 2000, 1024, 4
    # what the headline batch resolves to: ask a batch of that size
    b = capi.LBABatch()
    for i in range(per_gpu):
        b.add(synth.make_window(i, num_lines=lines))
    b.finalize()
    elim, cut = b.elimination(), b.window_chunks(0)
    b.solve(); b.download()
    in_batch = {i: "%08x" % (zlib.crc32(b.parameters(i).tobytes()) & 0xffffffff) for i in range(k)}
    b.close()
    out = {"lines": lines, "windows_per_gpu": per_gpu, "lba_elimination": elim, "chunks_per_window": cut, "checked_per_rank": k, "crc32": {}}
    for r in range(8):
        for i in range(k):
            wid = r * per_gpu + i
            x, s, _ = capi.lba_solve(synth.make_window(wid, num_lines=lines), lba_elimination=elim, chunks_per_window=cut)
            out["crc32"][str(wid)] = "%08x" % (zlib.crc32(x.tobytes()) & 0xffffffff)
    for i in range(k):
        assert out["crc32"][str(i)] == in_batch[i], "window %d: alone and in the headline batch differ" % i
    path = os.path.join(ROOT, "tests", "golden", "bench_digest.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, out["lba_elimination"], out["chunks_per_window"], len(out["crc32"]), "windows")


if __name__ == "__main__":
    main()
