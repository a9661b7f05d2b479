"""torch.profiler view of one benchmark forward: which ATen ops / memcpys run beside the library's kernels."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import harness, runtime, configs
from torch.profiler import profile, ProfilerActivity

B = 64
model = harness.build_nlvr(seed=0).cuda().eval()
images, text, targets = harness.nlvr_inputs(B, seed=0)
T = 8.612223847001898
with runtime.precision("bf16"), torch.no_grad():
    for _ in range(2):
        model(images, text, targets, temperature=T, train=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(images, text, targets, temperature=T, train=False)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=50))
