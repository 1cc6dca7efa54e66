"""GPU tests of the multi-GPU fan-out (BASELINE configs[3]: independent windows sharded over the GPUs of a node, RCCL).
Only one GPU is visible to the test box, so: (1) the whole per-rank path - batch -> device export -> all-gather /
all-reduce through torch.distributed's "nccl" backend (RCCL) - runs at world size 1 on the real device; (2) two ranks
sharing the one GPU run the same script in subprocesses (RCCL refuses two ranks on one device, so that leg uses gloo for the
rendezvous and the collectives, with the solves on the GPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [120, 40, 200, 75, 150, 60]

WORKER = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from slslam_amd import synth
from slslam_amd.dist import solve_shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
backend = os.environ["TEST_BACKEND"]
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
sizes = %(sizes)r
mk = lambda i: synth.make_window(500 + i, num_lines=sizes[i], num_kf=12, num_free=6)
r = solve_shard(mk, len(sizes), rank, world, 0, gather=(backend == "nccl" or world == 1), chunks_per_window=int(os.environ.get("TEST_CHUNKS", "0")))
lo, hi = r["range"]
local = np.concatenate([r["batch"].parameters(i - lo) for i in range(lo, hi)]) if hi > lo else np.zeros(0)
out = {"rank": rank, "range": [lo, hi], "iterations": r["iterations"], "initial_cost": r["initial_cost"], "final_cost": r["final_cost"],
       "local": local.tolist()}
if r["gathered"] is not None:
    out["gathered"] = [g.cpu().numpy().tolist() for g in r["gathered"]]
r["batch"].close()
print("RESULT " + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def _run(world, backend, chunks=0):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TEST_BACKEND=backend, TEST_CHUNKS=str(chunks), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "sizes": SIZES}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        import json
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:]))
    return sorted(outs, key=lambda o: o["rank"])


def test_world_size_one_through_rccl(hip):
    """batch -> slslam_lba_batch_export_device -> all_gather_into_tensor / all_reduce on the "nccl" backend (= RCCL),
    one rank on the real device: what every rank of the 8-GPU job executes."""
    (o,) = _run(1, "nccl")
    assert o["range"] == [0, len(SIZES)]
    assert len(o["gathered"]) == 1
    assert np.array_equal(np.array(o["gathered"][0]), np.array(o["local"]))      # device export == host download
    # same numbers as a plain one-process batch
    from slslam_amd import synth
    b = hip.LBABatch()
    for i, l in enumerate(SIZES):
        b.add(synth.make_window(500 + i, num_lines=l, num_kf=12, num_free=6))
    b.finalize(); b.solve(); b.download()
    ref = np.concatenate([b.parameters(i) for i in range(len(SIZES))])
    its = sum(b.summary(i)["num_successful_steps"] + b.summary(i)["num_unsuccessful_steps"] for i in range(len(SIZES)))
    assert np.array_equal(ref, np.array(o["local"])) and o["iterations"] == its
    b.close()


def test_two_ranks_on_one_gpu_match_one_rank(hip):
    """Two ranks, contiguous shards (`shard_range`), both solving on the one visible GPU: the concatenated results are
    BITWISE those of the one-rank run (1 / 2 / 4 / 8-rank runs give bit-identical per-window results).  A
    window's result is a function of its inputs, the options and the number of chunks it is cut into (the chunk partials are
    summed in chunk order); the library's automatic chunk count depends on the batch a window keeps company with, so the job
    fixes it (chunks_per_window) - what a multi-rank caller that wants rank-count-independent bytes does.  The all-reduced
    summary is the same on every rank."""
    (one,) = _run(1, "gloo", chunks=4)
    two = _run(2, "gloo", chunks=4)
    assert [o["range"] for o in two] == [[0, 3], [3, 6]]
    both = np.concatenate([np.array(o["local"]) for o in two])
    assert np.array_equal(both, np.array(one["local"]))
    for o in two:
        assert o["iterations"] == one["iterations"]
        assert abs(o["final_cost"] - one["final_cost"]) <= 1e-12 * one["final_cost"]


def test_bench_two_ranks_shared_gpu(hip):
    """The N > 1 path of the bench line itself (what the driver launches on an 8-GPU node), two ranks sharing the one visible GPU:
    `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with the gloo rendezvous (RCCL refuses two ranks on one
    device).  Checks what a first multi-GPU run must show: both ranks in the communicator, one clock per rank, the window ids of the
    cross-rank check spanning both shards, and the all-gathered results of BOTH ranks equal, byte for byte, to rank 0's re-solve of
    the same window ids in one-window batches (same chunk count, same elimination sweep)."""
    import json
    import socket as _s
    with _s.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SLSLAM_BENCH_SHARE_GPU="1", SLSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    windows, lines = 24, 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--windows", str(windows), "--lines", str(lines), "--no-cpu-baseline", "--no-overlap-run", "--no-extra-configs", "--stream-batches", "6"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["collective_backend"] == "gloo"
    assert len(d["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in d["per_rank_ms_per_step"])
    assert d["config"]["total_windows"] == 2 * windows and d["scaling"] == "weak"
    #  the STREAMED leg runs on every rank - its shard of every set through a stream object of its own, page-locked
    # arrays, refills built on the device, host threads = the CPU quota's share - and the line carries the per-rank blocks
    sb = d["streamed"]
    assert sb["ranks"] == 2 and len(sb["per_rank"]) == 2 and sb["host_threads_per_rank"] <= 2
    for x in sb["per_rank"]:
        assert x["value"] > 0 and x["device_builds"] == 6 and x["zero_copy_batches"] == 6 and x["windows_handed_to_the_host_path"] == 0
    assert abs(sb["value"] - sum(x["value"] for x in sb["per_rank"])) <= 1e-6 * sb["value"]
    chk = d["results_check"]
    ids = chk["window_ids"]
    assert min(ids) == 0 and max(ids) >= windows and len(ids) == 2 * chk["checked_per_rank"]      # both shards: [0, 24) and [24, 48)
    assert chk["bitwise_equal_to_rank0_resolve"] is True, chk
    assert d["lm_iterations"] > 0 and d["value"] > 0


def test_c_level_fan_out_through_rccl(tmp_path):
    """include/slslam_dist.h: the fan-out a C++ host drives - libslslam_dist.so on libslslam_hip.so + librccl,
    no Python, no torch - at world size 1 on the real device (RCCL refuses two ranks on one GPU and the box has one): communicator id
    through a file, ncclCommInitRank, the shard solved as one batch, ONE ncclAllReduce of the three sums of reference src/slam.cpp:949-952,
    ONE ncclAllGather of the solved parameters.  Results equal, to the byte, the same windows solved as a batch through the C ABI."""
    from slslam_amd import capi, synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "slslam_amd", "dist_c")])
    libdir = os.path.join(ROOT, "slslam_amd", "_lib")
    demo = os.path.join(ROOT, "tests", "_build", "dist_demo")
    os.makedirs(os.path.dirname(demo), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", demo,
                           os.path.join(ROOT, "tests", "host_cxx", "dist_demo.cpp"), "-L", libdir, "-lslslam_dist", "-lslslam_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    ws = [synth.make_window(700 + i, num_lines=n, num_kf=12, num_free=6) for i, n in enumerate(SIZES)]
    with open(tmp_path / "wins.bin", "wb") as f:
        np.array([len(ws)], dtype=np.int32).tofile(f)
        for w in ws:
            np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"])], dtype=np.int32).tofile(f)
            np.asarray(w["camera_index"], dtype=np.int32).tofile(f)
            np.asarray(w["line_index"], dtype=np.int32).tofile(f)
            np.asarray(w["fixed_index"], dtype=np.int32).tofile(f)
            np.asarray(w["observations"], dtype=np.float64).tofile(f)
            np.asarray(w["parameters"], dtype=np.float64).tofile(f)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([demo, "0", "1", str(tmp_path / "id.bin"), str(tmp_path / "wins.bin"), str(tmp_path / "out.bin"), "0", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    #  a failing shard - injected, and a malformed window - still enters the all-reduce, every rank gets an error,
    # nobody enters the all-gather, and the communicator serves the real solve below
    assert "error paths ok" in p.stdout, p.stdout + p.stderr
    # the streamed form (slslam_dist_stream_*): the shard three times through a depth-2 stream from page-locked arrays - refills built on the
    # device - with the per-set all-reduce; sums and parameters equal to the batch call
    assert "streamed fan-out ok" in p.stdout, p.stdout + p.stderr
    out = np.fromfile(tmp_path / "out.bin")
    sums, slot, count = out[:3], int(out[3]), int(out[4])
    gathered = out[5:5 + slot]
    b = capi.LBABatch()
    for w in ws:
        b.add(w)
    b.finalize()
    b.solve(); b.download()
    want = np.concatenate([b.parameters(i) for i in range(len(ws))])
    summ = [b.summary(i) for i in range(len(ws))]
    b.close()
    assert count == want.size == slot
    assert np.array_equal(gathered[:count], want)
    assert sums[0] == sum(s["num_successful_steps"] + s["num_unsuccessful_steps"] for s in summ)
    assert abs(sums[1] - sum(s["initial_cost"] for s in summ)) <= 1e-12 * sums[1]
    assert abs(sums[2] - sum(s["final_cost"] for s in summ)) <= 1e-12 * sums[2]
    # the shard rule is the Python layer's
    import ctypes as C
    from slslam_amd.dist import shard_range
    L = C.CDLL(os.path.join(libdir, "libslslam_dist.so"))
    L.slslam_dist_shard_range.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.slslam_dist_shard_range.restype = None
    for n, world in ((8192, 8), (10, 4), (3, 8), (0, 2), (1025, 3)):
        for r in range(world):
            lo, hi = C.c_longlong(), C.c_longlong()
            L.slslam_dist_shard_range(n, r, world, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == tuple(shard_range(n, r, world))
