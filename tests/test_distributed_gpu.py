"""N>1 PRODUCT path on a real MI355X: two processes (one GPU shared - the test box has one; `bench.py --gpus N` puts one rank per
GPU) each run the HIP BLIP_NLVR forward on their shard of the batch, and every shard is checked against the CPU oracle run
on that shard's samples (k = max_b count couples samples only within a shard).  Also exercises the per-device
k hand-over slots and the library from two processes at once."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, mode, B, T):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    from madtp_amd import dist as mdist, harness, hip, runtime, synth
    torch.cuda.set_device(0)
    w, r, _ = mdist.init("gloo")  # one GPU for both ranks: RCCL needs a device per rank, gloo moves the CPU copies
    assert (w, r) == (world, rank)
    hip.load()
    model = harness.build_nlvr(224, 0, "cuda")
    images = synth.synth_images(2 * B, 224, 7)
    ids = synth.synth_token_ids(B, 20, 7)
    ids[:, 0] = 30523
    att = torch.ones_like(ids)
    img_s, ids_s, att_s = mdist.shard_nlvr_batch(images, ids, att, rank, world)
    lo, hi = mdist.shard_range(B, rank, world)
    text = {"input_ids": ids_s.cuda(), "attention_mask": att_s.cuda()}
    targets = torch.zeros(hi - lo, dtype=torch.long, device="cuda")
    with runtime.precision(mode):
        mdist.barrier()
        logits, trace = harness.run_nlvr(model, img_s.cuda(), text, targets, T)
        logits2, _ = harness.run_nlvr(model, img_s.cuda(), text, targets, T)  # both ranks keep launching concurrently
    assert torch.equal(logits, logits2)
    allv = mdist.gather_logits(logits.cpu())
    assert allv.shape == (B, 2)
    assert mdist.max_over_ranks(1.0 + rank) == float(world)
    sets = {side: harness.compose_ids(trace[side], n0) for side, n0 in (("vit", 196), ("text", 19))}
    torch.save({"logits": logits.cpu(), "all": allv, "range": (lo, hi), "sets": sets}, os.path.join(out_dir, f"g{rank}.pt"))
    mdist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["fp32", "f16x3"])
def test_world2_hip_forward_matches_per_shard_oracle(tmp_path, mode):
    from madtp_amd import build
    build.build(verbose=False)
    world, B, T = 2, 5, 6.0  # odd batch: the ranks get 3 and 2 samples
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode, B, T), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"g{r}.pt"), weights_only=False) for r in range(world)]
    assert torch.equal(res[0]["all"], res[1]["all"])
    assert torch.equal(res[0]["all"], torch.cat([res[0]["logits"], res[1]["logits"]]))
    from madtp_amd import specs, synth
    from oracle import madtp_oracle as O
    images = synth.synth_images(2 * B, 224, 7)
    ids = synth.synth_token_ids(B, 20, 7)
    ids[:, 0] = 30523
    W = specs.synth_weights(specs.blip_nlvr_shapes(224), 0)
    for r in range(world):
        lo, hi = res[r]["range"]
        img = torch.cat([images[lo:hi], images[B + lo:B + hi]])
        tr = {}
        with torch.no_grad():
            ref = O.blip_nlvr_forward(W, img, ids[lo:hi], torch.ones_like(ids[lo:hi]), T, trace=tr)
        assert (ref - res[r]["logits"]).abs().max().item() < 1e-3
        for side, n0 in (("vit", 196), ("text", 19)):
            assert res[r]["sets"][side] == O.compose_ids(tr[side], n0), f"rank {r} {side}: kept sets differ from the shard oracle"


def _nccl_worker(rank, world, port, out_dir):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import numpy as np
    from madtp_amd import dist as mdist
    from madtp_amd.blip_retrieval import all_gather_scores, all_reduce_scores, rank_rows
    torch.cuda.set_device(rank)
    w, r, _ = mdist.init("nccl")  # RCCL: one device per rank
    assert (w, r) == (world, rank)
    assert mdist.max_over_ranks(1.0 + rank, device="cuda") == float(world)
    assert mdist.min_over_ranks(1.0 + rank, device="cuda") == 1.0
    assert mdist.sum_over_ranks([1.0, float(rank)], device="cuda") == [float(world), float(sum(range(world)))]
    # config 3's one exchange (compress_retrieval_dtp.py:202-205): every rank filled its row slice of both score matrices
    n_img, n_txt = 13, 31
    full_i2t = np.arange(n_img * n_txt, dtype=np.float32).reshape(n_img, n_txt)
    full_t2i = -np.arange(n_txt * n_img, dtype=np.float32).reshape(n_txt, n_img)
    mine = []
    for full in (full_i2t, full_t2i):
        m = np.full_like(full, -100.0)
        s, e = rank_rows(full.shape[0], rank, world)
        m[s:e] = full[s:e]
        mine.append(m)
    a, b = all_gather_scores(mine[0], mine[1])
    assert np.array_equal(a, full_i2t) and np.array_equal(b, full_t2i)
    ra, rb = all_reduce_scores(mine[0], mine[1])   # the reference's SUM all-reduce: uniformly shifted by -100 (world - 1)
    assert np.allclose(ra, full_i2t - 100.0 * (world - 1)) and np.allclose(rb, full_t2i - 100.0 * (world - 1))
    mdist.barrier()
    torch.distributed.destroy_process_group()


def _nccl_single_rank_worker(rank, port):
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from madtp_amd import dist as mdist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    assert dist.get_backend() == "nccl"
    assert mdist.max_over_ranks(3.5, device="cuda") == 3.5 and mdist.min_over_ranks(3.5, device="cuda") == 3.5
    assert mdist.sum_over_ranks([1.0, 2.0], device="cuda") == [1.0, 2.0]
    logits = torch.arange(10, dtype=torch.float32, device="cuda").reshape(5, 2)
    assert torch.equal(mdist.gather_logits(logits), logits)
    t = torch.full((1 << 20,), 2.0, device="cuda")   # a 4 MB all-reduce through the RCCL ring code path
    dist.all_reduce(t)
    assert float(t[0]) == 2.0 and float(t[-1]) == 2.0
    mdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_reductions_and_score_gather():
    """RCCL itself (backend "nccl", one device per rank) for the collectives the path uses: the MAX / SUM scalar reductions of
    bench.py and the retrieval score-matrix exchange.  Needs >= 2 GPUs: skipped on the one-GPU test boxes, runs on a node."""
    if torch.cuda.device_count() < 2:
        # one-GPU form: RCCL refuses two ranks on one device, so the communicator has ONE rank - librccl is loaded, the
        # communicator is created and the all_reduce (MAX / MIN / SUM) / all_gather / barrier calls of the path run through it
        mp.spawn(_nccl_single_rank_worker, args=(_free_port(),), nprocs=1, join=True)
        return
    world = 2
    mp.spawn(_nccl_worker, args=(world, _free_port(), ""), nprocs=world, join=True)
