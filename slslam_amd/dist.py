"""Fan-out of independent windows over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  Windows are independent, so the data path has NO collective: rank r solves the
contiguous shard `shard_range(n, r, world)` on its own GPU.  Two collectives exist, both once per
batch and latency-bound (KB..MB payloads):

* `allreduce_summary` - the three run-level sums the reference keeps (m_sum_num_iteration,
  m_sum_init_cost, m_sum_final_cost: reference src/slam.cpp:949-952) as ONE all-reduce;
* `allgather_parameters` - every rank's solved parameter vectors to every rank as ONE all-gather
  (ragged shards are padded to the longest).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced split of n windows: returns (begin, end) of this rank's shard."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n), world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_summary(num_iterations, initial_cost, final_cost, device=None):
    """Sum of (iterations, initial cost, final cost) over all ranks -> python floats."""
    t = torch.tensor([float(num_iterations), float(initial_cost), float(final_cost)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    v = t.tolist()
    return int(round(v[0])), v[1], v[2]


def allgather_parameters(local, local_len=None):
    """All-gather of each rank's flat float64 result vector (lengths may differ).
    Returns a list with one tensor per rank (on local.device)."""
    if local.dtype != torch.float64 or local.dim() != 1:
        raise ValueError("expected a flat float64 tensor")
    n = int(local.numel() if local_len is None else local_len)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local[:n]]
    world = dist.get_world_size()
    lens = torch.zeros(world, dtype=torch.int64, device=local.device)
    lens[dist.get_rank()] = n
    dist.all_reduce(lens, op=dist.ReduceOp.SUM)
    lens = lens.tolist()
    width = max(lens)
    buf = torch.zeros(width, dtype=torch.float64, device=local.device)
    buf[:n] = local[:n]
    out = torch.empty(world * width, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return [out[r * width:r * width + lens[r]] for r in range(world)]


def export_parameters_device(batch, device):
    """The solved parameter vectors of a finalized LBABatch as ONE flat float64 tensor on `device`, written by the library
    straight into the tensor's memory (slslam_lba_batch_export_device): no host round trip before a collective."""
    n = batch.total_parameters()
    out = torch.empty(max(n, 1), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream if out.is_cuda else None
    batch.export_device(out.data_ptr(), stream)
    return out[:n]


def solve_shard(make_window, total, rank, world, device_index, steps=1, gather=True, **opt):
    """Config 4 of BASELINE.json in one call: the windows [0, total) are split contiguously over the ranks
    (`shard_range`), rank `rank` builds and solves its shard on GPU `device_index` (`make_window(i)` returns window i),
    then ONE all-reduce of the run summary and - if `gather` - ONE all-gather of the solved parameters.
    Returns dict(range, iterations (all ranks), initial_cost, final_cost, gathered (list of per-rank tensors or None),
    batch (the caller closes it))."""
    from . import capi
    lo, hi = shard_range(total, rank, world)
    dev = torch.device("cuda", device_index)
    bt = capi.LBABatch(device=device_index)
    for i in range(lo, hi):
        bt.add(make_window(i))
    bt.finalize(**opt)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(steps):
        bt.reset(stream)
        bt.solve(stream)
    its = bt.iterations(stream)
    bt.download(stream)
    sums = [bt.summary(i) for i in range(hi - lo)]
    tot = allreduce_summary(its, sum(s["initial_cost"] for s in sums), sum(s["final_cost"] for s in sums), device=dev)
    gathered = None
    if gather:
        local = export_parameters_device(bt, dev)
        torch.cuda.current_stream(dev).synchronize()
        gathered = allgather_parameters(local)
    return {"range": (lo, hi), "iterations": tot[0], "initial_cost": tot[1], "final_cost": tot[2], "gathered": gathered, "batch": bt}
