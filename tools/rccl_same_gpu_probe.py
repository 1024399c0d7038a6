"""Does RCCL on this image let TWO ranks share ONE GPU?  (It would turn the gloo stand-ins of the N > 1 tests into real RCCL runs.)
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29701 tools/rccl_same_gpu_probe.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.full((1024,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank %d: all_reduce over RCCL with both ranks on cuda:0 -> %s" % (rank, x[0].item()), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank %d: %s: %s" % (rank, type(e).__name__, str(e)[:300].replace("\n", " ")), flush=True)
