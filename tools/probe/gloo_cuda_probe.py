"""Does this torch build's gloo backend all-reduce CUDA (HIP) tensors, two processes sharing one GPU?"""
import os
import sys
import importlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "RANK" not in os.environ:
    launch = importlib.import_module("iclr2025_3d-mom_amd.launch")
    rc, out = launch.spawn_ranks(2, [sys.executable, os.path.abspath(__file__)], timeout=120)
    print("rc", rc, out)
    sys.exit(0)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
t = torch.full((1000,), float(dist.get_rank() + 1), device="cuda")
try:
    w = dist.all_reduce(t, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    print("gloo cuda all_reduce ok:", float(t[0]), flush=True)
except Exception as e:
    print("gloo cuda all_reduce FAILED:", repr(e)[:300], flush=True)
dist.destroy_process_group()
