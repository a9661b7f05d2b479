"""Data-parallel helpers: one process per GPU, batch sharded, weights replicated, no collective
inside the forward.  RCCL ("nccl" backend on ROCm) / gloo (CPU tests) is used only for barriers and for reducing
timing / metric scalars, exactly as the reference's eval loops do (utils.py:48-59, compress_nlvr_dtp.py:128-136)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (world, rank, local_rank)."""
    world, rank, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, init_method="env://")
    return world, rank, local_rank


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device_index):
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function), or None when it cannot be read."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except Exception:
        return None


MIN_CORES_PER_RANK = 6  # bench.py runs four in-flight worker threads next to the main thread


def pin_rank_to_cores(local_rank, local_world, device_index=None, cpus_of_gpu=None):
    """Every rank's host thread spins on a pinned slot 24 times per forward (the k hand-over, csrc/prune.hip) and feeds ~300
    launches: give each rank its own slice of host cores, on the NUMA node of ITS GPU when sysfs tells (ranks that share a
    node split that node's cores among themselves), else an even split of the process's current affinity mask.  Returns the
    CPU set chosen (also when sched_setaffinity is unavailable: empty set = nothing done)."""
    if not hasattr(os, "sched_getaffinity") or local_world <= 1:
        return set()
    allowed = sorted(os.sched_getaffinity(0))
    local = cpus_of_gpu if cpus_of_gpu is not None else (gpu_local_cpus(device_index) if device_index is not None else None)
    pool = sorted(set(allowed) & set(local)) if local else []
    if len(pool) >= 2 and len(pool) < len(allowed):
        # ranks whose GPUs share this node: assume the usual even layout (local_world GPUs over the nodes that exist)
        share = max(1, round(local_world * len(pool) / len(allowed)))
        idx = local_rank % share
        n = len(pool) // share
        mine = pool[idx * n:(idx + 1) * n] if n else pool
    else:
        n = len(allowed) // local_world
        mine = allowed[local_rank * n:(local_rank + 1) * n] if n else allowed
    if len(mine) < MIN_CORES_PER_RANK:
        return set()  # a slice this small would make the rank's own threads (main + in-flight workers) contend: leave it to the OS
    if mine:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            return set()
    return set(mine)


def shard_range(n_samples, rank, world):
    """Contiguous split of `n_samples` NLVR samples; each sample's two images stay on the same rank
    (blip_nlvr.py:67 splits image_embeds by targets.size(0))."""
    base, rem = divmod(n_samples, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_nlvr_batch(images, input_ids, attention_mask, rank, world):
    """images [2B,...] (image0 batch then image1 batch), ids/mask [B,L] -> this rank's shard in the same layout."""
    B = input_ids.shape[0]
    lo, hi = shard_range(B, rank, world)
    img = torch.cat([images[lo:hi], images[B + lo:B + hi]], dim=0)
    return img, input_ids[lo:hi], attention_mask[lo:hi]


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the elapsed time of the timed region)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value, device="cpu"):
    """MIN all-reduce of a python float (the fastest rank's own time in the timed region: max - min shows stragglers)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def sum_over_ranks(values, device="cpu"):
    if not dist.is_initialized():
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def gather_logits(logits):
    """all_gather of per-rank logits [b_r,2] (variable b_r padded to the max) -> [B,2] on every rank."""
    if not dist.is_initialized():
        return logits
    world = dist.get_world_size()
    n = torch.tensor([logits.shape[0]], device=logits.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    m = int(max(int(x.item()) for x in ns))
    pad = torch.zeros((m,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    pad[: logits.shape[0]] = logits
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: int(k.item())] for o, k in zip(outs, ns)], dim=0)


def allreduce_gradients(parameters, bucket_bytes=256 << 20, average=True):
    """The exchange step of data-parallel TRAINING (the reference wraps the model in DistributedDataParallel,
    compress_nlvr_dtp.py:251-253): sums (averages) the .grad of `parameters` over the ranks with bucketed all-reduces - gradients
    are flattened into buckets of up to bucket_bytes in parameter order, one all-reduce per bucket (RCCL over xGMI is per-link
    bound: few large messages), and copied back.  Parameters without a gradient on this rank (a layer that did not run) take part
    with zeros so that every rank issues the same collectives.  Returns the number of buckets.  No-op for world size 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    params = [p for p in parameters if p.requires_grad]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nbytes = p.numel() * p.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    for b in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        o = 0
        for p in b:
            n = p.numel()
            g = flat[o:o + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += n
    return len(buckets)
