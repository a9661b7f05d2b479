"""N>1 data-parallel path on CPU: world_size-2 gloo.  Checks the sharding rule (pairs stay together, shards cover the
batch), the barrier / MAX / SUM reductions bench.py relies on, and the multi-GPU parity rule of SURVEY.md 8(e):
each shard's result equals the oracle run on that shard's samples (k = batch max couples samples only within a
shard, so a sharded global batch is NOT expected to equal the unsharded batch)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from madtp_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(4)
    from madtp_amd import synth
    from oracle import madtp_oracle as O
    w, r, _ = mdist.init("gloo")
    assert (w, r) == (world, rank)
    B, T = 3, 6.0  # odd batch: ranks get 2 and 1 samples
    images = synth.synth_images(2 * B, 224, 7)
    ids = synth.synth_token_ids(B, 12, 7)
    att = torch.ones_like(ids)
    img_s, ids_s, att_s = mdist.shard_nlvr_batch(images, ids, att, rank, world)
    lo, hi = mdist.shard_range(B, rank, world)
    assert img_s.shape[0] == 2 * (hi - lo) and torch.equal(img_s[: hi - lo], images[lo:hi])
    assert torch.equal(img_s[hi - lo:], images[B + lo:B + hi])
    W = torch.load(os.path.join(out_dir, "weights.pt"), mmap=True, weights_only=True)  # generated once by the parent (~15 s)
    with torch.no_grad():
        logits = O.blip_nlvr_forward(W, img_s, ids_s, att_s, T)
    mdist.barrier()
    allv = mdist.gather_logits(logits)
    assert allv.shape == (B, 2)
    t = mdist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    assert mdist.min_over_ranks(1.0 + rank) == 1.0
    assert mdist.sum_over_ranks([hi - lo, 1.0]) == [float(B), float(world)]
    torch.save({"logits": logits, "all": allv, "range": (lo, hi)}, os.path.join(out_dir, f"r{rank}.pt"))
    mdist.barrier()
    torch.distributed.destroy_process_group()


def test_shard_range_covers_batch():
    for n in (1, 2, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(600)
def test_world2_gloo_sharded_forward_matches_per_shard_oracle(tmp_path):
    world = 2
    port = _free_port()
    from madtp_amd import specs, synth
    W = specs.synth_weights(specs.blip_nlvr_shapes(224), 0)
    torch.save(W, os.path.join(tmp_path, "weights.pt"))
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    # every rank saw the same gathered logits, in global sample order
    assert torch.equal(res[0]["all"], res[1]["all"])
    assert torch.equal(res[0]["all"], torch.cat([res[0]["logits"], res[1]["logits"]]))
    # per-shard parity: re-run each shard standalone in this process
    from oracle import madtp_oracle as O
    images = synth.synth_images(6, 224, 7)
    ids = synth.synth_token_ids(3, 12, 7)
    for r in range(world):
        lo, hi = res[r]["range"]
        img = torch.cat([images[lo:hi], images[3 + lo:3 + hi]])
        with torch.no_grad():
            ref = O.blip_nlvr_forward(W, img, ids[lo:hi], torch.ones_like(ids[lo:hi]), 6.0)
        assert (ref - res[r]["logits"]).abs().max().item() < 1e-5


def _retrieval_worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(4)
    from madtp_amd import blip_retrieval as br, harness
    from oracle import madtp_oracle as O
    mdist.init("gloo")
    W = torch.load(os.path.join(out_dir, "weights.pt"), mmap=True, weights_only=True)  # generated once by the parent
    batches, ids, att = harness.retrieval_inputs(3, 2, 4, 224, 35, 0)
    with torch.no_grad():
        i2t, t2i = O.retrieval_evaluate(W, batches, ids, att, 6.0, 2, rank=rank, world=world)
    a, b = br.all_reduce_scores(i2t.numpy(), t2i.numpy())
    c, d = br.all_gather_scores(i2t.numpy(), t2i.numpy())
    torch.save({"i2t": i2t, "t2i": t2i, "sum_i2t": a, "sum_t2i": b, "gat_i2t": c, "gat_t2i": d},
               os.path.join(out_dir, f"retr{rank}.pt"))
    mdist.barrier()
    torch.distributed.destroy_process_group()


def test_retrieval_rank_slices_and_all_reduce(tmp_path):
    """SURVEY 8f rank 1, multi-GPU: queries shard by rank (compress_retrieval_dtp.py:158-162, 181-183), one SUM all-reduce
    of the score matrices (:200-203).  world_size-2 gloo with the CPU oracle as the scorer: the slices are disjoint, cover
    all queries, and the reduced matrices rank candidates exactly like the single-rank evaluation."""
    import numpy as np
    from madtp_amd import harness, specs
    from oracle import madtp_oracle as O
    world = 2
    W = specs.synth_weights(specs.blip_retrieval_shapes(224), 0)
    torch.save(W, os.path.join(str(tmp_path), "weights.pt"))
    mp.spawn(_retrieval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f"retr{r}.pt"), weights_only=False) for r in range(world)]
    batches, ids, att = harness.retrieval_inputs(3, 2, 4, 224, 35, 0)
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(4)  # (the workers' thread count: the same reduction splits as the ranks used)
    try:
        with torch.no_grad():
            full_i2t, full_t2i = O.retrieval_evaluate(W, batches, ids, att, 6.0, 2)
    finally:
        torch.set_num_threads(prev_threads)  # later test modules of this process keep the machine's thread count
    for key, full in (("i2t", full_i2t.numpy()), ("t2i", full_t2i.numpy())):
        own = [(o[key].numpy() != -100.0) for o in outs]
        assert not (own[0] & own[1]).any() and np.array_equal(own[0] | own[1], full != -100.0)
        red = outs[0]["sum_" + key]
        assert np.array_equal(red, outs[1]["sum_" + key])
        done = full != -100.0
        assert np.allclose(red[done], full[done] - 100.0 * (world - 1), atol=1e-4)   # uniform shift: same ranking
        assert (red[~done] == -100.0 * world).all()
        # the row-slice all-gather reproduces the single-rank matrices themselves, on every rank
        # (bit-exactly the rows each rank computed; against the separate single-rank oracle run only to float noise, the CPU
        #  oracle's reductions depend on the thread count)
        assert np.array_equal(outs[0]["gat_" + key], outs[1]["gat_" + key])
        gat = outs[0]["gat_" + key]
        for r, o in enumerate(outs):
            assert np.array_equal(gat[own[r]], o[key].numpy()[own[r]])
        assert np.array_equal(gat != -100.0, done) and np.allclose(gat[done], full[done], atol=1e-4)


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    mdist.init("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    extra = torch.nn.Linear(4, 4)  # runs on rank 0 only: the other rank has no gradient for it
    x, y = torch.randn(8, 16), torch.randn(8, 4)
    lo, hi = mdist.shard_range(8, rank, world)
    out = model(x[lo:hi])
    if rank == 0:
        out = out + 0.0 * extra(out).sum()
    ((out - y[lo:hi]) ** 2).mean().backward()
    params = list(model.parameters()) + list(extra.parameters())
    nb = mdist.allreduce_gradients(params, bucket_bytes=1024)  # small buckets: several collectives
    torch.save({"grads": [p.grad.clone() for p in params], "buckets": nb}, os.path.join(out_dir, f"grad{rank}.pt"))
    mdist.barrier()
    torch.distributed.destroy_process_group()


def test_allreduce_gradients_equals_full_batch(tmp_path):
    """The exchange step of data-parallel training (madtp_amd.dist.allreduce_gradients, the DDP all-reduce of
    compress_nlvr_dtp.py:251-253 as bucketed collectives): with the batch sharded over two gloo ranks and a mean loss per shard, the
    averaged gradients equal the full-batch gradients on every rank, a parameter that has a gradient on one rank only included."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), f"grad{r}.pt"), weights_only=False) for r in range(world)]
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 4))
    extra = torch.nn.Linear(4, 4)
    x, y = torch.randn(8, 16), torch.randn(8, 4)
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in model.parameters()]
    assert outs[0]["buckets"] == outs[1]["buckets"] and outs[0]["buckets"] >= 2
    for r in range(world):
        for a, b in zip(outs[r]["grads"][:len(ref)], ref):
            assert (a - b).abs().max().item() < 1e-6
        for a, b in zip(outs[r]["grads"][len(ref):], outs[0]["grads"][len(ref):]):
            assert torch.equal(a, b)   # the rank-0-only layer: the same (averaged) gradient everywhere


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver starts its 1-GPU run (no launcher, no WORLD_SIZE): bench.py re-executes itself under
    torch.distributed.run with 2 ranks and rank 0 prints ONE JSON line with n_gpus 2 (MADTP_BENCH_DRY=1: gloo and a sleep in place of
    the GPU forward - the launch, rendezvous, barriers, MAX reduction and the line are the real code).  Started by a launcher
    (WORLD_SIZE set) the same command does not launch again."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MADTP_BENCH_DRY="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    # the contract's own command line (a launcher provides the ranks): same line, no second launch
    port = "29631"
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                        env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    l2 = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(l2) == 1 and json.loads(l2[0])["n_gpus"] == 2
    # a rank count that does not match the launcher's is refused with a message, not a hang
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                        capture_output=True, text=True, timeout=120)
    assert r3.returncode != 0 and "WORLD_SIZE=2" in (r3.stderr + r3.stdout)
