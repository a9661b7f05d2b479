"""Does replaying the sync-free ViT encoder as a hipGraph pay (round-5 review, item 6)?  The twelve pruned blocks of
VisionTransformer.forward enqueued by madtp_vit_encoder_async without a host read (enqueue-only form, dims_host = NULL), (a) eagerly,
(b) captured once into a hipGraph (torch.cuda.CUDAGraph on the capture stream the library's per-stream workspaces were warmed on) and
replayed - next to (c) the sync-free call with its one host read and (d) the default encoder call with the per-layer host read of k.
Checks that a replay leaves the eager call's records and output.   usage: graph_replay_probe.py [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madtp_amd import configs, harness, hip, runtime
hip.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
T = configs.temperature_for("nlvr", 64, 0.5)[0]
model = harness.build_nlvr(224, 0, "cuda")
venc = model.visual_encoder


def timed(fn, reps=60):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"precision {mode}; ms per call of the 12-block encoder loop alone (patch embedding / final norm / att_ft sum not included)")
print(f"{'images':>6s} {'host k':>9s} {'sync-free':>10s} {'enqueue-only eager':>19s} {'graph replay':>13s}   (one graph = 12 x 11 kernel nodes + 2 memsets)")
with runtime.precision(mode), torch.no_grad():
    for n in (1, 2, 4, 8, 16):
        images, _, _ = harness.nlvr_inputs(max(1, (n + 1) // 2), 224, 20, seed=3)
        img = images[:n].contiguous()
        patches, np_ = venc.patch_embed.run(img)
        x = hip.assemble_tokens(patches, venc.cls_token, venc.pos_embed, n, np_)
        weights, qargs, deferred = venc._encoder_call_prep(img, model.space_dict)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            host_k = timed(lambda: hip.vit_encoder(weights, x, qargs, T))
            sync_free = timed(lambda: hip.vit_encoder(weights, x, qargs, T, sync_free=True))
            def eager():
                r = hip.vit_encoder(weights, x, qargs, T, sync_free=True, enqueue_only=True)
                s.synchronize()
                return r
            eager_ms = timed(eager)
            ref = eager()
            ref_dims = ref.dims_dev.clone()
            ref_y = ref.buf.clone()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run = hip.vit_encoder(weights, x, qargs, T, sync_free=True, enqueue_only=True)
            def replay():
                g.replay()
                s.synchronize()
            graph_ms = timed(replay)
            replay()
            assert torch.equal(run.dims_dev[: 13 * 4], ref_dims[: 13 * 4]), "the replayed records differ from the eager call's"
            n_last = int(run.dims_dev[12 * 4].item())
            D = x.shape[-1]
            off = run.ptr(11, "y") - run.buf.data_ptr()   # the last layer's output rows, through the layout both runs share
            nbytes = n * n_last * D * 4
            assert torch.equal(run.buf[off:off + nbytes], ref_y[off:off + nbytes]), "the replayed output differs from the eager call's"
        print(f"{n:6d} {host_k:9.3f} {sync_free:10.3f} {eager_ms:19.3f} {graph_ms:13.3f}", flush=True)
