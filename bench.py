#!/usr/bin/env python
"""bench.py — LBA iterations/sec on synthetic 10-keyframe / ~2000-line windows (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank solves its own batch of
independent sliding windows (LBAProblem::build'ed once, resident in HBM) with the full
Levenberg-Marquardt loop (max 10 iterations, Huber loss, Ceres-1.7 policy) through the C ABI.
Work per GPU is fixed as N grows (weak scaling); there is no data-path collective, only one
all-reduce of the run summary per timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = LM iterations (successful + unsuccessful trust-region
steps, as slam.cpp:949-950 counts them) of all ranks / max-over-ranks wall time of the K steps.
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams beyond that SHARE a queue: with torch's
# stream pool alive (the informational two-stream run creates it) the three slot streams of the streamed leg then queued behind each other - uploads
# waiting for another batch's solve, 6-8 ms per batch (tools/streamed_dbg.py).  Eight queues keep them apart; set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from slslam_amd import capi, synth  # noqa: E402
from slslam_amd.dist import allgather_parameters, allreduce_summary, export_parameters_device, shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: fp64 vector (and matrix) peak


def baseline_metric():
    """The metric string of BASELINE.json (the driver matches on it)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "LBA iterations/sec (10-KF window, ~2k lines) at 1/2/4/8 GPU; traj RMSE vs ref"


def algorithmic_bytes_linearise(counts):
    """Algorithmic HBM bytes of ONE launch of the dominant kernel (linearise + Schur sweep):
    SURVEY.md 8d per-window figure for one of the three observation sweeps,
    M*(64+8) + L*32 + C*48 + (6 Cf)^2*8, summed over the windows the launch processes."""
    return sum(72 * m + 32 * l + 48 * c + 8 * (6 * cf) ** 2 for (c, cf, l, m) in counts)


def algorithmic_flops_linearise(windows):
    """Algorithmic fp64 flops of ONE launch of the dominant kernel (secondary figure): per
    observation ~600 (residual + analytic Jacobian) + ~520 (block products), per line 288 k^2 for the Schur outer
    products over its k free-camera observations."""
    total = 0.0
    for w in windows:
        free_obs = np.asarray(w["fixed_index"]).reshape(-1, 2)[:, 0] == 0
        kf = np.bincount(np.asarray(w["line_index"])[free_obs], minlength=int(w["num_lines"]))
        total += 1120.0 * len(w["camera_index"]) + 288.0 * float((kf.astype(np.float64) ** 2).sum())
    return total


def cpu_baseline(windows, budget_s=12.0):
    """The oracle (oracle/*.c: Jet<10> autodiff + Huber + Ceres-1.7 LM + block Schur, one thread,
    as the reference pins Ceres to num_threads = 1) timed on this host on a bounded sample of the
    SAME windows.  Returns (iterations/s, description, solved parameters list)."""
    from oracle import pyoracle          # cpu_baseline leg only
    its, t0, outs = 0, time.perf_counter(), []
    for w in windows:
        x, s, _ = pyoracle.lba_solve(w, linear_solver=1)
        its += s["num_successful_steps"] + s["num_unsuccessful_steps"]
        outs.append(x)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return its / dt, "%d of the bench windows (%.1f s, %d LM iterations), 1 thread" % (len(outs), dt, its), outs


def host_cpu_info():
    """What this process may actually use of the host: scheduler affinity, the cgroup CPU bandwidth quota (cgroup v2 cpu.max or
    v1 cpu.cfs_quota_us / cpu.cfs_period_us; None = unlimited), the cgroup's effective cpuset and the host's 1-minute load average
    (a shared host shows up there: other tenants' runnable threads)."""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["sched_affinity"] = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    info["cgroup_cpu_quota"] = quota
    for path in ("/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.effective_cpus"):
        try:
            txt = open(path).read().strip()
            n = 0
            for part in filter(None, txt.split(",")):
                a, _, b = part.partition("-")
                n += (int(b) - int(a) + 1) if b else 1
            info["cgroup_cpuset_effective"] = n
            break
        except Exception:
            continue
    try:
        info["loadavg_1min"] = os.getloadavg()[0]
    except OSError:
        pass
    usable = info["sched_affinity"]
    if quota:
        usable = max(1, min(usable, int(quota + 0.999)))
    if info.get("cgroup_cpuset_effective"):
        usable = min(usable, info["cgroup_cpuset_effective"])
    info["usable_threads"] = usable
    return info


def cpu_baseline_all_cores(windows, budget_s=15.0):
    """Same oracle, fanned out over independent windows on every host core: one window per
    OpenMP task inside the C library, as many windows as fit the time budget.  Threads = min(scheduler affinity, cgroup CPU
    quota, cgroup cpuset): what the container may use, not what the host advertises."""
    from oracle import pyoracle          # cpu_baseline leg only
    cores = host_cpu_info()["usable_threads"]
    per_core = max(1, int(budget_s / 0.25 / 2))      # ~0.2-0.25 s per 2000-line window and thread
    # bounded sample: on a shared host the usable parallelism can be far below the advertised core count
    sample = windows[:min(len(windows), cores * per_core, 384)]
    pyoracle.lib()
    t0 = time.perf_counter()
    _, sums = pyoracle.lba_solve_many(sample, cores, linear_solver=1)
    dt = time.perf_counter() - t0
    its = sum(s["num_successful_steps"] + s["num_unsuccessful_steps"] for s in sums)
    return its / dt, cores, "%d of the bench windows (%.1f s wall incl. marshalling, %d LM iterations), one window per OpenMP task, %d threads" % (
        len(sample), dt, its, min(cores, len(sample)))


FP64_MEASURED_CEILING_TFLOPS = {1: 33.5, 2: 48.5, 3: 53.6, 4: 56.3, 8: 62.1}   # tools/micro/mfma_f64_bench.hip on MI355X: v_fma_f64
                                                                              # stream, by waves per SIMD (spec: 78.6)


def algorithmic_bytes_backsub(counts):
    """Algorithmic HBM bytes of ONE launch of the second observation sweep (back-substitution + cost at the candidate point,
    SURVEY.md 8d's second and third sweeps folded into one pass): observations 72 B each, every line's parameters read
    and its candidate written (32 + 32 B), accepted and candidate camera poses (2 x 48 B), the camera step (6 Cf x 8 B)."""
    return sum(72 * m + 64 * l + 96 * c + 48 * cf for (c, cf, l, m) in counts)


def reduced_solve_mfma_count(n):
    """v_mfma_f64_16x16x4_f64 instructions of one blocked (6 Cf)^2 Cholesky in k_reduced_solve: per block step 4 per panel
    tile and 4 per trailing tile (n = 60: 64)."""
    nt = (n + 15) // 16
    return sum(4 * (nt - kb - 1) + 4 * ((nt - kb - 1) * (nt - kb)) // 2 for kb in range(nt))


def ceres_probe():
    """Is a Ceres Solver installation visible on this host (headers or shared library)?  The reference links Ceres 1.7.0
    (README:8, src/CMakeLists.txt:14); without it the CPU baseline is the in-repo oracle (kind "port")."""
    import glob
    import subprocess
    hits = []
    for pat in ("/usr/include/ceres/ceres.h", "/usr/local/include/ceres/ceres.h", "/opt/*/include/ceres/ceres.h",
                "/usr/lib/x86_64-linux-gnu/libceres*", "/usr/local/lib/libceres*", "/usr/lib/cmake/Ceres/*", "/usr/local/share/Ceres/*",
                "/usr/local/lib/cmake/Ceres/*"):
        hits += glob.glob(pat)
    try:
        ld = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout
        hits += [l.strip() for l in ld.splitlines() if "libceres" in l]
    except Exception:
        pass
    return {"ceres_available": bool(hits), "found": hits[:4]}


def ceres_reference_leg(windows, gpu_params, nsample=8):
    """The true-Ceres leg of the CPU baseline, run ONLY where ceres_probe() finds an installation (none of the boxes seen so far): builds
    tools/ceres_harness.cpp (this repo's own functor and wiring on the public Ceres API) with g++, solves a sample of the bench windows on one
    thread as the reference configures it, and compares the results with the GPU's.  Any failure (no compiler flags that work, link errors, a
    Ceres without sparse Cholesky) returns None and the line keeps kind = "port"."""
    import glob
    import subprocess
    import tempfile
    try:
        d = tempfile.mkdtemp(prefix="slslam_ceres_")
        exe = os.path.join(d, "ceres_harness")
        inc = [i for pat in ("/usr/include/eigen3", "/usr/local/include/eigen3", "/opt/*/include/eigen3") for i in glob.glob(pat)]
        cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "tools")] + [x for i in inc for x in ("-I", i)] + \
              [os.path.join(ROOT, "tools", "ceres_harness.cpp"), "-o", exe, "-lceres", "-lglog", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"built": False, "error": r.stderr[-400:]}
        its, secs, worst = 0, 0.0, 0.0
        for i, w in enumerate(windows[:nsample]):
            fin, fout = os.path.join(d, "w%d.bin" % i), os.path.join(d, "x%d.bin" % i)
            with open(fin, "wb") as f:
                np.array([w["num_cameras"], w["num_lines"], len(w["camera_index"]), 10, 1], dtype=np.int32).tofile(f)
                for k, dt in (("camera_index", np.int32), ("line_index", np.int32), ("fixed_index", np.int32), ("observations", np.float64), ("parameters", np.float64)):
                    np.asarray(w[k], dtype=dt).tofile(f)
            rr = subprocess.run([exe, fin, fout, "1"], capture_output=True, text=True, timeout=900)
            if rr.returncode != 0:
                return {"built": True, "ran": False, "error": rr.stderr[-400:]}
            j = json.loads(rr.stdout.strip().splitlines()[-1])
            its += j["lm_iterations"]; secs += j["seconds"]
            worst = max(worst, float(np.abs(np.fromfile(fout) - gpu_params[i]).max()))
        return {"built": True, "ran": True, "value": its / secs if secs > 0 else None, "unit": "LM iterations/s", "cores": 1, "kind": "reference",
                "sample": "%d of the bench windows through Ceres (tools/ceres_harness.cpp), 1 thread" % min(nsample, len(windows)),
                "max_abs_parameter_diff_vs_gpu": worst}
    except Exception as e:      # the probe was wrong about the installation: the port stays the baseline
        return {"built": False, "error": repr(e)[:400]}


def cross_rank_result_check(batches, where, lo, B, rank, world, dev, local_rank, lines, elim, k_check=4, keep=0):
    """Results are a function of the window alone, so every rank's results can be checked against ANY rank's solve of the same
    window id, bit for bit.  Outside the timed region: every rank exports the solved parameters of
    the first `k_check` windows of its shard on the device, ONE all-gather (RCCL for N > 1) brings them to every rank, and rank 0
    solves the same window ids itself - in a small batch of its own, cut into the same number of chunks - and compares the
    bytes.  At N = 1 this is the same check against a differently composed batch.  Returns the dict for the JSON line."""
    import zlib
    import torch.distributed as dist
    k = min(k_check, B)
    vecs, chunks = [], []
    for i in range(k):
        si, li = where[i]
        batches[si].download()
        vecs.append(batches[si].parameters(li))
        chunks.append(batches[si].window_chunks(li))
    width = sum(v.size for v in vecs)
    local = torch.from_numpy(np.concatenate(vecs)).to(dev)
    meta = torch.tensor([lo] + chunks + [v.size for v in vecs], dtype=torch.int64, device=dev)
    if world > 1:
        allp = torch.empty(world * width, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allp, local)                      # the collective under test: RCCL all-gather over xGMI
        allm = torch.empty(world * meta.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allm, meta)
    else:
        allp, allm = local, meta
    if rank != 0:
        return None
    allp = allp.cpu().numpy().reshape(world, width)
    allm = allm.cpu().numpy().reshape(world, -1)
    ids, crcs, equal, maxdiff = [], [], True, 0.0
    elim_eff = batches[0].elimination() or elim          # the sweep the timed batches ran (the automatic choice depends on the batch size)
    for r in range(world):
        lo_r, ch, sizes = int(allm[r, 0]), [int(x) for x in allm[r, 1:1 + k]], [int(x) for x in allm[r, 1 + k:1 + 2 * k]]
        off = 0
        for i in range(k):
            got = allp[r, off:off + sizes[i]]
            off += sizes[i]
            bt = capi.LBABatch(device=local_rank)
            bt.add(synth.make_window(lo_r + i, num_lines=lines))
            bt.finalize(use_graph=0, chunks_per_window=ch[i], lba_elimination=elim_eff, lba_keep_jacobian=keep)
            bt.solve(); bt.download()
            mine = bt.parameters(0)
            bt.close()
            ids.append(lo_r + i)
            crcs.append("%08x" % (zlib.crc32(got.tobytes()) & 0xffffffff))
            if not np.array_equal(got, mine):
                equal = False
                maxdiff = max(maxdiff, float(np.abs(got - mine).max()))
    # ... and against the committed 1-rank record of the same window ids (tests/golden/bench_digest.json, tools/make_bench_digest.py): a
    # window's bytes are a function of the window, the sweep and the cut, whatever the number of ranks
    stored = None
    try:
        dg = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_digest.json")))
        if dg["lines"] == lines and dg["windows_per_gpu"] == B and dg["lba_elimination"] == elim_eff and all(int(x) == dg["chunks_per_window"] for x in allm[:, 1:1 + k].reshape(-1)):
            known = [(i, c) for i, c in zip(ids, crcs) if str(i) in dg["crc32"]]
            if known:
                stored = {"compared": len(known), "equal": all(dg["crc32"][str(i)] == c for i, c in known)}
    except Exception:
        stored = None
    return {"window_ids": ids, "crc32_of_gathered_parameters": crcs, "bitwise_equal_to_rank0_resolve": equal,
            "equal_to_stored_1_rank_digest": stored,
            "max_abs_diff": maxdiff, "checked_per_rank": k, "lba_elimination": elim_eff,
            "how": "first %d windows of every rank's shard: parameters all-gathered on the device, rank 0 solves the same ids in "
                   "1-window batches with the same chunk count and compares bytes" % k}


def time_batch(windows, device, steps, warmup, **opt):
    """Resident batch of `windows`, `steps` timed solves (hipGraph replay): (LM iterations / s, ms per solve)."""
    bt = capi.LBABatch(device=device)
    for w in windows:
        bt.add(w)
    bt.finalize(use_graph=1, **opt)
    for _ in range(max(warmup, 1)):
        bt.reset(); bt.solve()
    bt.iterations(clear=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bt.reset(); bt.solve()
    its = bt.iterations()
    dt = time.perf_counter() - t0
    bt.close()
    return its / dt, 1e3 * dt / steps


def streamed_block(windows, device, resident_value, resident_params, batches_timed=8, depth=3, host_threads=0, mode="pinned", **opt):
    """BASELINE config 4 taken literally - a STREAM of windows: every window arrives as the five host arrays the reference builds per
    window (src/slam.cpp:899-921) and its solved parameters go back into the caller's array (:957-972).  Inside the timed region, per
    batch of len(windows) windows: the LBAProblem::build stage, the same captured solve the resident headline replays, the results into the
    callers' arrays - `depth` batches in flight (slslam_lba_stream_*).  mode:
      "pinned"   the caller's arrays live in page-locked memory (slslam_pinned_alloc): the copy engine takes the observations and parameters
                 from where they are, the host threads only narrow the three index arrays on the way (16 -> 4 B per observation), the batch
                 is built on the device (csrc/lba_device_build.h) and the solved parameters are written back in place;
      "packed"   as "pinned", and the caller's packer writes the three index arrays narrowed to one 32-bit word per observation
                 (slslam_pack_indices / slslam_lba_stream_submit_packed): 68 instead of 80 bytes per observation over the link;
      "pageable" ordinary arrays: the host threads copy them into a pinned staging buffer (indices narrowed on the way), the device builds;
      "host"     the round-5 path (device_build = -1): packing on the host threads, pinned image, upload, download, copy-out.
    Every batch is a set of host arrays over the rank's windows, rotated so that every batch is laid out differently (the read-only input
    arrays are shared between the sets, every set has parameter arrays of its own); the first 2 x depth submits (batch builds, graph
    capture, every slot's first refill with its staging allocations) are warm-up.  Never the headline `value`: it measures the host, the host link and the GPU together."""
    B = len(windows)
    warm = 2 * depth                 # every slot is built once and REFILLED once before the clock starts (a slot's first refill allocates its staging blocks)
    nsets = warm + batches_timed
    base = capi.WindowSet(windows, pinned=(mode in ("pinned", "packed")), packed=(mode == "packed"))
    sets = []
    for k in range(nsets):
        r = (k * 37) % B
        sets.append(base.derive(list(range(r, B)) + list(range(r))))
    if mode == "host":
        opt = dict(opt, device_build=-1)
    st = capi.LBAStream(device=device, depth=depth, host_threads=host_threads, **opt)
    tick = []
    for k in range(warm):                           # warm-up: builds the slots' batches, then one refill of each
        if k >= depth:
            st.collect(tick[k - depth], want_summaries=False)
        tick.append(st.submit(sets[k]))
    for k in range(warm - depth, warm):
        st.collect(tick[k], want_summaries=False)
    s0, b0 = st.stats(), st.build_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tick, marks = [], []
    for k in range(batches_timed):
        marks.append(time.perf_counter())           # a batch PERIOD: collect of the batch `depth` back + submit of this one
        if k >= depth:
            st.collect(tick[k - depth], want_summaries=False)
        tick.append(st.submit(sets[warm + k]))
    marks.append(time.perf_counter())
    for k in range(max(0, batches_timed - depth), batches_timed):
        st.collect(tick[k], want_summaries=False)
    dt = time.perf_counter() - t0                   # ... and `value` also pays for draining the last `depth` batches
    periods = sorted(b - a for a, b in zip(marks[depth:-1], marks[depth + 1:]))
    steady = periods[len(periods) // 2] if periods else None
    s1, b1 = st.stats(), st.build_stats()
    its = s1["lm_iterations"] - s0["lm_iterations"]
    nwin = s1["windows"] - s0["windows"]
    # the streamed results are the resident batch's results, byte for byte (same windows, same sweep, same cut)
    equal, checked = True, 0
    k = warm + batches_timed - 1
    r = (k * 37) % B
    for j in (0, 1, B // 2, B - 1):
        i = (j + r) % B                              # window i of the rank sits at position j of set k
        if i in resident_params:
            checked += 1
            equal = equal and bool(np.array_equal(sets[k].parameters(j), resident_params[i]))
    m = sum(len(w["camera_index"]) for w in windows)
    npar = sum(8 * (6 * w["num_cameras"] + 4 * w["num_lines"]) for w in windows)
    # (page-locked arrays: the host threads narrow the three index arrays to one word per observation on the way - 64 + 4 B per observation cross the link)
    link_in = {"pinned": 68 * m + npar, "packed": 68 * m + npar, "pageable": 68 * m + npar}.get(mode)
    if link_in is None:
        link_in = 8 * 8 * m + 4 * 2 * m + npar + sum(40 * w["num_lines"] for w in windows)
    out = {"value": its / dt, "unit": "LM iterations/s", "fraction_of_resident": (its / dt) / resident_value if resident_value else None,
           "host_threads": s1["host_threads"], "mode": mode, "ms_per_batch": 1e3 * dt / batches_timed, "windows_per_batch": B,
           "batches_timed": batches_timed, "depth": depth,
           "steady_ms_per_batch": 1e3 * steady if steady else None,           # median period between submits once the pipeline is full (a long stream's rate)
           "steady_value": (its / batches_timed) / steady if steady else None,
           "refills": s1["refills"] - s0["refills"], "rebuilds": s1["builds"] - s0["builds"],
           "device_builds": b1["device_builds"] - b0["device_builds"], "zero_copy_batches": b1["zero_copy"] - b0["zero_copy"],
           "windows_handed_to_the_host_path": b1["fallback_windows"] - b0["fallback_windows"],
           "ms_per_batch_in_submit": (s1["ms_submit"] - s0["ms_submit"]) / batches_timed,
           "ms_per_batch_waiting_in_collect": (s1["ms_collect_wait"] - s0["ms_collect_wait"]) / batches_timed,
           "ms_per_batch_copying_results_out": (s1["ms_collect_copy"] - s0["ms_collect_copy"]) / batches_timed,
           "host_link_in_MB_per_batch": link_in / 1e6, "host_link_out_MB_per_batch": npar / 1e6, "lm_iterations": its,
           "bitwise_equal_to_resident_batch": equal if checked else None, "windows_compared": checked,
           "timed_region": {"pinned": "per batch: index arrays narrowed by the host threads + copy-engine ingest of the callers' page-locked observations and parameters + build on the device + hipGraph solve + results written in place, %d batches in flight",
                            "packed": "as pinned, the indices narrowed by the caller (one 32-bit word per observation), %d batches in flight",
                            "pageable": "per batch: staging copy (host threads, indices narrowed) + ingest + build on the device + hipGraph solve + D2H + copy-out, %d batches in flight",
                            "host": "per batch: pack (host threads) + pinned H2D + hipGraph solve + D2H + copy-out into the callers' arrays, %d batches in flight"}[mode] % depth}
    st.close()
    for ws in sets:
        ws.close()
    base.close()
    return out


def mixed_precision_block(windows, counts, device, resident_value, steps=5, warmup=2):
    """slslam_solver_options.lba_precision = 1 on the same resident batch: the steady elimination sweeps form the CAMERA Jacobian of an
    observation in packed fp32; the line Jacobian, every block product and accumulation, the cost and the trust-region bookkeeping stay fp64
    (include/slslam_hip.h).  MEASURED: it buys nothing (the step is not faster than the fp64 step) - float line Jacobians change LM accept /
    reject decisions, and what is left to fp32 is a few per cent of the sweep.  Reported as one ratio; its tolerance against the double path is
    asserted by tests/test_gpu_lba.py::test_mixed_precision_solves."""
    bt = capi.LBABatch(device=device)
    for w in windows:
        bt.add(w)
    bt.finalize(use_graph=1, lba_precision=1)
    for _ in range(max(warmup, 1)):
        bt.reset(); bt.solve()
    bt.iterations(clear=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bt.reset(); bt.solve()
    its = bt.iterations()
    dt = time.perf_counter() - t0
    bt.download()
    out = {"value": its / dt, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt / steps, "vs_double_path": (its / dt) / resident_value if resident_value else None,
           "lm_iterations": its, "lba_elimination": bt.elimination(), "dtype": "fp32 camera Jacobian only; fp64 line Jacobian, products, accumulation", "verdict": "buys nothing: kept as a tested option",
           "sum_final_cost": sum(bt.summary(i)["final_cost"] for i in range(len(windows)))}
    bt.close()
    return out


def single_window_latency(lines, device, **shape):
    """The reference's call protocol (slam.cpp:924-944: one window per keyframe, each built from the result of the one
    before): ms per 10-iteration solve of ONE window resident in HBM (hipGraph replay) and through slslam_lba_solve
    (host buffers: pack + upload + solve + download)."""
    w = synth.make_window(5, num_lines=lines, **shape)
    _, resident = time_batch([w], device, 30, 3)
    bt = capi.LBABatch(device=device); bt.add(w); bt.finalize(use_graph=1)
    path = {0: "tiled sweeps", 1: "fused motion-only", 2: "global memory (lba_big.h)", 3: "mixed"}[bt.path()]
    chunks = bt.window_chunks(0)
    bt.close()
    capi.lba_solve(w)
    t0 = time.perf_counter()
    for _ in range(10):
        _, s, _ = capi.lba_solve(w)
    host = 1e3 * (time.perf_counter() - t0) / 10
    return {"lines": lines, "observations": len(w["camera_index"]), "resident_ms_per_solve": resident, "host_buffer_ms_per_solve": host,
            "lm_iterations": s["num_successful_steps"] + s["num_unsuccessful_steps"], "path": path, "chunks": chunks}


def pose_graph_block():
    """BASELINE configs[4]: 260-pose graph with 8 loop closures (the myung-dong scale), 10 iterations, on one GPU."""
    from oracle import pyoracle          # cpu_baseline leg only
    g = synth.make_pose_graph(7, num_poses=260, num_loops=8)
    out = {"poses": 260, "edges": len(g["pose_index_1"]), "loop_closures": 8}
    res = {}
    for name, kw in (("structured", {}), ("dense_fp64", dict(po_dense_factor=1)), ("dense_fp32", dict(po_factor_fp32=1))):
        capi.po_solve(g, **kw)
        t0 = time.perf_counter()
        for _ in range(3):
            x, s, tm = capi.po_solve_timed(g, **kw)
        res[name] = x
        d = {"ms_per_solve_host_clock": 1e3 * (time.perf_counter() - t0) / 3, "device_ms": tm["total_ms"],
             "factorisation_ms_per_iteration": tm["factor_ms"],
             "unknowns": tm["unknowns"], "steps": [s["num_successful_steps"], s["num_unsuccessful_steps"]], "final_cost": s["final_cost"]}
        if name != "structured":
            flops = tm["unknowns"] ** 3 / 3.0
            peak = FP64_VECTOR_PEAK_TFLOPS if name == "dense_fp64" else 157.3
            tf = flops / (d["factorisation_ms_per_iteration"] * 1e-3) / 1e12
            d["mfma"] = {"algorithmic_flops": flops, "achieved_tflops": tf, "peak_tflops": peak, "utilisation": tf / peak}
        else:
            d["junction_unknowns"] = tm["junction_unknowns"]
        out[name] = d
    dx = (res["dense_fp32"] - res["dense_fp64"]).reshape(-1, 6)
    out["fp32_vs_fp64"] = {"max_rotation_diff_rad": float(np.abs(dx[:, :3]).max()), "max_translation_diff_m": float(np.abs(dx[:, 3:]).max())}
    out["structured_vs_dense_max_diff"] = float(np.abs(res["structured"] - res["dense_fp64"]).max())
    t0 = time.perf_counter()
    xo, so, _ = pyoracle.po_solve(g)
    out["cpu_oracle_ms"] = 1e3 * (time.perf_counter() - t0)                  # dense 1554^2 Cholesky: NOT what the reference configures
    out["max_diff_vs_oracle"] = float(np.abs(res["structured"] - xo).max())
    # the fair CPU figure: the reference configures SPARSE_NORMAL_CHOLESKY (src/po_problem.cpp:68); the oracle's envelope
    # Cholesky exploits the same sparsity (block tridiagonal chain + loop-closure fill), one thread as the reference pins it
    pyoracle.po_solve(g, linear_solver=2)
    t0 = time.perf_counter()
    for _ in range(5):
        xs, ss, _ = pyoracle.po_solve(g, linear_solver=2)
    out["cpu_oracle_sparse_ms"] = 1e3 * (time.perf_counter() - t0) / 5
    out["cpu_oracle_sparse_max_diff_vs_dense"] = float(np.abs(xs - xo).max())
    return out


def order_line(out):
    """The driver's record keeps the contract keys, `roofline`, `cpu_baseline` and the last ~2 KB of the line: the streamed figures (BASELINE
    config 4, the product's end-to-end rate) go LAST, compact, so that they are in it."""
    tail = ["streamed_host_packer", "streamed_pageable", "streamed_packed_indices", "streamed"]
    keep = ("value", "unit", "fraction_of_resident", "host_threads", "host_threads_per_rank", "ranks", "mode", "steady_value", "steady_ms_per_batch", "ms_per_batch",
            "batches_timed", "windows_per_batch", "depth", "device_builds", "zero_copy_batches", "windows_handed_to_the_host_path", "ms_per_batch_in_submit",
            "host_link_in_MB_per_batch", "bitwise_equal_to_resident_batch", "per_rank", "error")
    o = {k: v for k, v in out.items() if k not in tail}
    for k in tail:
        if k in out and isinstance(out[k], dict):
            o[k] = {q: out[k][q] for q in keep if q in out[k]}
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU")
    ap.add_argument("--lines", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-run", action="store_true", help="skip the informational two-stream measurement")
    ap.add_argument("--graph", action="store_true", help="(default since round 4) time the captured hipGraph replay; kept for old command lines")
    ap.add_argument("--eager", action="store_true", help="time eager launches with per-kernel hipEvents in the timed region (the round 1-3 mode)")
    ap.add_argument("--profile-steps", type=int, default=5, help="eager profiled steps after the timed region (per-kernel times, roofline)")
    ap.add_argument("--chunks", type=int, default=0, help="waves cooperating on one window (0 = library default)")
    ap.add_argument("--elim", type=int, default=0,
                    help="lba_elimination: 0 auto, 1 LDS-atomic sweep, 2 / 3 matrix-core sweep with 1 / 2 waves per chunk")
    ap.add_argument("--keep-jacobian", type=int, default=0,
                    help="lba_keep_jacobian: 0 (default) every sweep linearises, 1 the sweep after a rejected step replays the kept Jacobian blocks (measured slower)")
    ap.add_argument("--gather-results", action="store_true",
                    help="also all-gather the solved parameters of every rank inside the timed region (one RCCL all-gather per step)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the 500-line, latency and pose-graph blocks")
    ap.add_argument("--no-result-check", action="store_true", help="skip the cross-rank bitwise check of the results (outside the timed region)")
    ap.add_argument("--stream-depth", type=int, default=3, help="batches in flight in the streamed leg")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed leg (config 4 as a stream of host-side windows)")
    ap.add_argument("--stream-batches", type=int, default=32, help="timed batches of the streamed leg (the drain of the last `depth` batches is inside the timed region: the more batches, the closer to a long stream)")
    ap.add_argument("--host-threads", type=int, default=0, help="host threads of the streamed leg (0 = up to 16)")
    ap.add_argument("--streams", type=int, default=1,
                    help="the rank's windows are split into this many batches on separate HIP streams, so that the "
                         "latency-bound kernels of one batch (reduced solve, LM update) overlap the sweeps of the other")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    # one process per GPU; SLSLAM_BENCH_SHARE_GPU=1 lets several ranks share a device (dry runs of the
    # multi-rank flow on a 1-GPU box, together with SLSLAM_BENCH_BACKEND=gloo)
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not os.environ.get("SLSLAM_BENCH_SHARE_GPU"):
        raise SystemExit("LOCAL_RANK %d but only %d HIP device(s) visible" % (local_rank, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("SLSLAM_BENCH_BACKEND", "nccl")      # "nccl" is RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # who is in the job: communicator size after init and the device every rank sits on (one distinct GPU per rank expected)
    props = torch.cuda.get_device_properties(local_rank)
    me = "%s|%s|%s" % (os.uname().nodename, getattr(props, "pci_bus_id", local_rank), getattr(props, "uuid", ""))
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = dist.get_world_size()
    else:
        seen, ranks_seen = [me], 1

    # ---- synthetic inputs of the named shape, distinct per rank, resident in HBM before timing
    # BASELINE configs[3]: a stream of world x B independent windows, window i = synth.make_window(i), split contiguously
    # over the ranks (dist.shard_range): every rank holds B windows (weak scaling), no data-path collective
    B = args.windows
    lo, hi = shard_range(B * world, rank, world)
    windows = [synth.make_window(i, num_lines=args.lines) for i in range(lo, hi)]
    use_graph = not args.eager          # the production launch mode: one hipGraph replay per solve (one host call, insensitive to a busy host)
    ns = max(1, min(args.streams, B))
    bstreams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(ns - 1)]
    batches, owner = [], []
    for si in range(ns):
        bt = capi.LBABatch(device=local_rank)
        for wi in range(si, B, ns):
            bt.add(windows[wi])
            owner.append((wi, si, len(bt.sizes) - 1))
        bt.finalize(use_graph=1 if use_graph else 0, chunks_per_window=args.chunks, lba_elimination=args.elim, lba_keep_jacobian=args.keep_jacobian)
        bt.set_profiling(not use_graph)
        batches.append(bt)
    where = {wi: (si, li) for wi, si, li in owner}
    counts = [(w["num_cameras"], w["num_free_cameras"], w["num_lines"], len(w["camera_index"])) for w in windows]

    def run_step():
        for bt, st in zip(batches, bstreams):
            bt.reset(st.cuda_stream)
            bt.solve(st.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    for bt, st in zip(batches, bstreams):
        bt.iterations(st.cuda_stream, clear=True)
        bt.set_profiling(not use_graph)          # drop warm-up events

    barrier()
    t0 = time.perf_counter()
    gathered_bytes = 0
    for _ in range(args.steps):
        run_step()
        if args.gather_results:          # results needed on every rank: device export + ONE all-gather per step
            for bt, st in zip(batches, bstreams):
                with torch.cuda.stream(st):
                    local = export_parameters_device(bt, dev)
                st.synchronize()
                parts = allgather_parameters(local)
                gathered_bytes = sum(int(x.numel()) * 8 for x in parts)
    iters_local = sum(bt.iterations(st.cuda_stream) for bt, st in zip(batches, bstreams))   # synchronises the streams
    iters_total, _, _ = allreduce_summary(iters_local, 0.0, 0.0, device=dev)
    barrier()
    dt_local = time.perf_counter() - t0
    t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
    per_rank = [dt_local]
    if world > 1:
        tt = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(tt, t)
        per_rank = tt.tolist()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    check = cross_rank_result_check(batches, where, lo, B, rank, world, dev, local_rank, args.lines, args.elim, keep=args.keep_jacobian) if not args.no_result_check else None

    # ---- per-kernel times (roofline): outside the timed region, same process, same resident batches - a profiled pass of eager
    # launches with hipEvents around every kernel (events recorded by graph nodes cannot be read back on this runtime).  The sum
    # of the kernel times per step must agree with the graph-replay step measured above.
    profile_steps = 0
    if use_graph and args.profile_steps > 0:
        for bt in batches:
            bt.set_profiling(True)
        profile_steps = args.profile_steps
        for _ in range(profile_steps):
            run_step()
        torch.cuda.synchronize()

    # ---- second, informational measurement (not `value`): the same windows as two half-batches on two HIP
    # streams, each replaying its captured hipGraph, so that the latency-bound kernels of one half (reduced
    # solve, LM update) overlap the observation sweeps of the other.  Per-kernel durations are not comparable in
    # this mode (kernels of the two halves share the GPU), so the roofline numbers come from the region above.
    overlap = None
    if ns == 1 and not args.no_overlap_run and B >= 2:
        ostreams = [torch.cuda.current_stream(), torch.cuda.Stream()]
        obatches = []
        for si in range(2):
            bt = capi.LBABatch(device=local_rank)
            for wi in range(si, B, 2):
                bt.add(windows[wi])
            bt.finalize(use_graph=1, chunks_per_window=args.chunks, lba_elimination=args.elim, lba_keep_jacobian=args.keep_jacobian)
            obatches.append(bt)

        def orun():
            for bt, st in zip(obatches, ostreams):
                bt.reset(st.cuda_stream)
                bt.solve(st.cuda_stream)
        orun()
        torch.cuda.synchronize()
        for bt, st in zip(obatches, ostreams):
            bt.iterations(st.cuda_stream, clear=True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            orun()
        oit = sum(bt.iterations(st.cuda_stream) for bt, st in zip(obatches, ostreams))
        oit_total, _, _ = allreduce_summary(oit, 0.0, 0.0, device=dev)
        barrier()
        odt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(odt, op=dist.ReduceOp.MAX)
        overlap = {"value": oit_total / float(odt.item()), "unit": "LM iterations/s", "hip_streams": 2, "launch": "hipGraph replay",
                   "ms_per_step": 1e3 * float(odt.item()) / max(args.steps, 1), "lm_iterations": oit_total}
        for bt in obatches:
            bt.close()

    # ---- results of the last step (outside the timed region)
    for bt, st in zip(batches, bstreams):
        bt.download(st.cuda_stream)
    summaries = [batches[where[i][0]].summary(where[i][1]) for i in range(B)]
    init_cost = sum(s["initial_cost"] for s in summaries)
    final_cost = sum(s["final_cost"] for s in summaries)

    if rank == 0:
        out = {
            "metric": baseline_metric(),
            "value": iters_total / elapsed,
            "unit": "LM iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic 10 free + 10 fixed keyframe window, %d lines, ~%d observations; "
                                   "full LM solve (Huber, <=10 iterations, Schur + back-substitution)" % (
                                       args.lines, int(np.mean([c[3] for c in counts]))),
                       "windows_per_gpu": B, "total_windows": B * world, "lines": args.lines,
                       "parallelism": "windows [0, %d) split contiguously over %d GPU(s) (shard_range), no data-path collective%s" % (
                           B * world, world, "; one all-gather of the results per step (%d MB)" % (gathered_bytes >> 20) if args.gather_results else ""),
                       "launch": "hipGraph replay" if use_graph else "eager + hipEvents", "hip_streams": ns,
                       "lba_keep_jacobian": args.keep_jacobian},
            "lm_iterations": iters_total,
            "sum_initial_cost_rank0": init_cost, "sum_final_cost_rank0": final_cost,
            # multi-rank evidence: size of the communicator after init (RCCL when backend == "nccl"), distinct devices the ranks
            # sit on, every rank's own clock over the timed region, and the cross-rank bitwise check of the results
            "rccl_ranks_seen": ranks_seen, "collective_backend": backend, "distinct_devices_seen": len(set(seen)),
            "per_rank_ms_per_step": [1e3 * x / max(args.steps, 1) for x in per_rank],
            "results_check": check,
        }
        if overlap is not None:
            out["two_streams_overlapped"] = overlap
        kts = [bt.kernel_times() for bt in batches]
        kt = {k: (sum(x[k][0] for x in kts), sum(x[k][1] for x in kts)) for k in kts[0]}
        ms, n = kt["linearise_schur"]
        if n > 0:
            # one launch covers the windows of ONE stream's batch
            bytes_launch = algorithmic_bytes_linearise(counts) / ns
            achieved = bytes_launch / (ms / n * 1e-3) / 1e9
            traffic, tj = None, {}
            traffic_source = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("lines") == args.lines and tj.get("lba_elimination", 1) == batches[0].elimination():
                        traffic = tj.get("hbm_bytes_per_launch") * (B / ns) / tj.get("windows")
                        # NOT measured in this run: PMC counters need rocprofv3 around the process; the figure is the committed
                        # result of the separate --pmc passes (tools/gpu_r3_profile.sh), scaled to this batch size
                        traffic_source = "profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command, not this run" % tj.get("source", "round2_v3")
                except Exception:
                    traffic = None
            flops_launch = algorithmic_flops_linearise(windows) / ns
            tflops = flops_launch / (ms / n * 1e-3) / 1e12
            elim_eff = batches[0].elimination()
            kname = {1: "k_linearise_schur<false, 0>", 2: "k_eliminate_mfma<1>", 3: "k_eliminate_mfma<2>", 4: "k_eliminate_grouped<false>"}.get(elim_eff, "?")
            out["roofline"] = {"bound": "hbm", "kernel": kname, "lba_elimination": elim_eff, "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": ms / n, "launches": n,
                               # what actually bounds the kernel: fp64 issue at two waves per SIMD
                               # (256 VGPRs) - a v_fma_f64 stream reaches 48.5 TFLOP/s there, not the 78.6 of the spec -
                               # together with the LDS fp64 atomics of the per-wave partial system
                               "binding": ("fp64 issue at 2 waves/SIMD (v_fma_f64 stream: measured ceiling %.1f TFLOP/s) with the Schur products on v_mfma_f64_16x16x4_f64, whose 64 cycles each are not hidden behind the VALU work (DESIGN.md section 7d)" if elim_eff == 4 else
                                           "fp64 VALU issue at 2 waves/SIMD (measured ceiling %.1f TFLOP/s) + LDS ds_add_f64") % FP64_MEASURED_CEILING_TFLOPS[2],
                               # secondary view: the kernel's arithmetic intensity (~28 flop / algorithmic byte) is above the
                               # fp64 ridge (78.6 TF / 8 TB/s ~ 10 flop / B), so the vector-fp64 roofline is the nearer one
                               "fp64_vector": {"algorithmic_flops_per_launch": flops_launch, "achieved": tflops,
                                               "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_VECTOR_PEAK_TFLOPS}}
            out["roofline"]["fp64_vector"]["frac_of_measured_ceiling_2_waves_per_simd"] = tflops / FP64_MEASURED_CEILING_TFLOPS[2]
            ksteps = profile_steps if use_graph else args.steps
            out["kernel_ms_per_step"] = {k: v[0] / max(ksteps, 1) for k, v in kt.items() if v[1] > 0}
            ksum = sum(out["kernel_ms_per_step"].values())
            out["kernel_times_from"] = ("%d eager profiled steps (hipEvents around every launch) after the timed region, same resident batch" % ksteps) if use_graph \
                else "hipEvents in the timed region"
            # the two views of a step must agree: kernels back to back inside one graph launch vs the same kernels timed one by one
            out["launch_consistency"] = {"sum_kernel_ms_per_step": ksum, "timed_ms_per_step": out["ms_per_step"],
                                         "ratio": ksum / out["ms_per_step"], "within_5_percent": abs(ksum / out["ms_per_step"] - 1.0) <= 0.05}
            if not out["launch_consistency"]["within_5_percent"]:
                print("bench.py: WARNING sum of kernel times %.3f ms vs timed step %.3f ms differ by more than 5 %%" % (ksum, out["ms_per_step"]), file=sys.stderr)
            bms, bn = kt["backsub"]
            if bn > 0:
                bb = algorithmic_bytes_backsub(counts) / ns
                ach = bb / (bms / bn * 1e-3) / 1e9
                btraffic = None
                try:
                    if tj.get("lines") == args.lines and "k_backsub" in tj:
                        btraffic = tj["k_backsub"]["hbm_bytes_per_launch"] * (B / ns) / tj.get("windows")
                except Exception:
                    btraffic = None
                out["roofline_backsub"] = {"bound": "hbm", "kernel": "k_backsub", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ach / HBM_PEAK_GBS, "traffic": btraffic, "traffic_source": traffic_source if btraffic else None,
                                           "algorithmic_bytes_per_launch": bb,
                                           "avg_launch_ms": bms / bn, "launches": bn,
                                           "binding": "fp64 VALU issue at 2 waves/SIMD"}
            sms, sn = kt["reduced_solve"]
            if sn > 0:
                nm = sum(reduced_solve_mfma_count(6 * c[1]) for c in counts) / ns
                tf = nm * 2048.0 / (sms / sn * 1e-3) / 1e12
                out["reduced_solve_mfma"] = {"instruction": "v_mfma_f64_16x16x4_f64", "mfma_per_launch": nm, "avg_launch_ms": sms / sn,
                                             "achieved_tflops": tf, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "utilisation": tf / FP64_VECTOR_PEAK_TFLOPS,
                                             "note": "60x60 systems: latency-bound (one workgroup per window), 3 % of the step"}
        if world == 1 and not args.no_cpu_baseline:
            v, sample, outs = cpu_baseline(windows)
            hinfo = host_cpu_info()
            out["cpu_baseline"] = {"value": v, "unit": "LM iterations/s", "cores": 1, "kind": "port", "sample": sample,
                                   "host_cores_available": os.cpu_count(), "cgroup_cpu_quota": hinfo["cgroup_cpu_quota"],
                                   "host": hinfo}
            va, ca, sa = cpu_baseline_all_cores(windows)
            out["cpu_baseline_all_cores"] = {"value": va, "unit": "LM iterations/s", "cores": ca, "kind": "port", "sample": sa,
                                             # threads the process may run on (sched_getaffinity) vs how well they scale:
                                             # a shared host delivers far fewer than it advertises
                                             "parallel_efficiency": va / (ca * v) if v > 0 and ca > 0 else None,
                                             "equivalent_full_cores": va / v if v > 0 else None}
            probe = ceres_probe()
            out["cpu_baseline"].update(probe)
            if probe["ceres_available"]:
                # a Ceres installation on this box: the reference's own solver, timed beside the port (never seen so far - untested path)
                leg = ceres_reference_leg(windows, [batches[where[i][0]].parameters(where[i][1]) for i in range(min(8, B))])
                out["cpu_baseline"]["ceres_leg"] = leg
                if leg and leg.get("ran") and leg.get("value"):
                    out["cpu_baseline_port"] = dict(out["cpu_baseline"])
                    out["cpu_baseline"].update({"value": leg["value"], "kind": "reference", "sample": leg["sample"], "cores": 1})
            # trajectory error of the GPU solve against the oracle solve of the same windows
            err = []
            for i, xo in enumerate(outs):
                xg = batches[where[i][0]].parameters(where[i][1])
                nf = 6 * windows[i]["num_free_cameras"]
                err.append(np.linalg.norm(synth.camera_centers(xg[:nf]) - synth.camera_centers(xo[:nf]), axis=1))
            err = np.concatenate(err)
            out["traj_error_vs_oracle"] = {"rms_m": float(np.sqrt((err ** 2).mean())), "mean_m": float(err.mean()),
                                           "keyframes": int(err.size)}
        if world == 1 and not args.no_streamed and ns == 1:
            resident_params = {i: batches[where[i][0]].parameters(where[i][1]).copy() for i in range(B)} if B <= 4096 else {}
            # the resident batches are done: their HBM, graphs and queues go back before the stream of windows starts (an idle batch kept open beside
            # the stream's slots makes the process's HIP streams share hardware queues, and uploads then wait behind solves: tools/streamed_dbg.py)
            for bt in batches:
                bt.close()
            batches = []
            sopt = dict(depth=args.stream_depth, chunks_per_window=args.chunks, lba_elimination=args.elim, lba_keep_jacobian=args.keep_jacobian)
            try:
                # the caller's arrays in page-locked memory: read in place, built on the device, written back in place - ONE host thread
                out["streamed"] = streamed_block(windows, local_rank, out["value"], resident_params, batches_timed=args.stream_batches,
                                                 host_threads=args.host_threads or 1, mode="pinned", **sopt)
            except capi.SlslamError as e:
                out["streamed"] = {"error": str(e)}
            if not args.no_extra_configs:
                try:
                    # the caller's packer narrows the indices (68 instead of 80 bytes per observation over the link)
                    out["streamed_packed_indices"] = streamed_block(windows, local_rank, out["value"], resident_params, batches_timed=args.stream_batches,
                                                                    host_threads=args.host_threads or 1, mode="packed", **sopt)
                    # ordinary (pageable) arrays: the staging copy on TWO host threads, the build on the device
                    out["streamed_pageable"] = streamed_block(windows, local_rank, out["value"], resident_params, batches_timed=max(6, args.stream_batches // 2),
                                                              host_threads=args.host_threads or 2, mode="pageable", **sopt)
                    # the round-5 path for comparison: everything on the host threads (up to 16)
                    out["streamed_host_packer"] = streamed_block(windows, local_rank, out["value"], resident_params, batches_timed=max(6, args.stream_batches // 2),
                                                                 host_threads=args.host_threads, mode="host", **sopt)
                except capi.SlslamError as e:
                    out["streamed_pageable"] = {"error": str(e)}
        if world == 1 and not args.no_extra_configs:
            for bt in batches:           # release HBM before the extra measurements
                bt.close()
            batches = []
            try:
                out["mixed_precision"] = mixed_precision_block(windows, counts, local_rank, out["value"])
                out["mixed_precision"]["sum_final_cost_double_path"] = final_cost
            except capi.SlslamError as e:
                out["mixed_precision"] = {"error": str(e)}
            # BASELINE configs[1]: 10-keyframe / 500-line window
            nb = min(B, 512)
            w500 = [synth.make_window(2_000_000 + i, num_lines=500) for i in range(nb)]
            v500, ms500 = time_batch(w500, local_rank, 5, 2)
            c2 = {"workload": "synthetic 10 free + 10 fixed keyframe window, 500 lines, ~%d observations" % int(np.mean([len(w["camera_index"]) for w in w500])),
                  "batched": {"windows": nb, "value": v500, "unit": "LM iterations/s", "ms_per_step": ms500},
                  "single_window": single_window_latency(500, local_rank)}
            if not args.no_cpu_baseline:
                from oracle import pyoracle          # cpu_baseline leg only
                t0 = time.perf_counter()
                _, so, _ = pyoracle.lba_solve(w500[0], linear_solver=1)
                dt = time.perf_counter() - t0
                c2["cpu_oracle_single_solve_ms"] = 1e3 * dt
                c2["cpu_oracle_lm_iterations_per_s"] = (so["num_successful_steps"] + so["num_unsuccessful_steps"]) / dt
            out["config2_window_500_lines"] = c2
            out["latency_single_window"] = single_window_latency(args.lines, local_rank)
            # the window sizes of the reference's own study (house scene: 74 lines; W free + W fixed keyframes).  W = 40 is
            # beyond the tiled sweeps and takes the global-memory path (lba_big.h, lba_big_solve.h)
            out["latency_study_windows"] = {
                "W=%d" % W: single_window_latency(74, local_rank, num_kf=2 * W, num_free=W, mean_track=mt)
                for W, mt in ((5, 8.4), (10, 16.5), (20, 32.0), (40, 61.0))}
            if not args.no_cpu_baseline:
                out["config5_pose_graph"] = pose_graph_block()
        if world == 1:
            print(json.dumps(order_line(out)))
    if world > 1:
        # ---- BASELINE config 4 as it reads - a STREAM of windows sharded over the GPUs: every rank streams its shard of every set through a
        # stream object of its own (page-locked arrays, the build stage on the device, results in place), host threads = the quota's share
        sb = None
        if not args.no_streamed and ns == 1:
            resident_params = {}
            for bt in batches:
                bt.close()
            batches = []
            quota = host_cpu_info().get("cgroup_cpu_quota") or os.cpu_count() or 1
            ht = args.host_threads or max(1, min(2, int(quota) // world))
            dist.barrier()
            try:
                sb = streamed_block(windows, local_rank, (iters_total / elapsed) / world, resident_params, batches_timed=args.stream_batches,
                                    host_threads=ht, mode="pinned", depth=args.stream_depth, chunks_per_window=args.chunks, lba_elimination=args.elim,
                                    lba_keep_jacobian=args.keep_jacobian)
            except capi.SlslamError as e:
                sb = {"error": str(e)}
            allsb = [None] * world
            dist.all_gather_object(allsb, sb)
            if rank == 0:
                good = [x for x in allsb if x and "error" not in x]
                keep = ("value", "fraction_of_resident", "host_threads", "steady_ms_per_batch", "ms_per_batch", "device_builds", "zero_copy_batches",
                        "windows_handed_to_the_host_path", "ms_per_batch_in_submit")
                out["streamed"] = {"value": sum(x["value"] for x in good), "unit": "LM iterations/s",
                                   "fraction_of_resident": sum(x["value"] for x in good) / out["value"] if good else None,
                                   "host_threads_per_rank": ht, "ranks": len(good), "mode": "pinned",
                                   "per_rank": [({k: x.get(k) for k in keep} if x and "error" not in x else x) for x in allsb],
                                   "timed_region": good[0]["timed_region"] if good else None}
        if rank == 0:
            print(json.dumps(order_line(out)))
    for bt in batches:
        bt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
