"""RCCL sanity on a one-GPU box: a world-size-1 `nccl` group, the collectives bench.py and the dictionary update
use (all_reduce MAX, barrier, all_gather).  The multi-rank RCCL path itself needs one GPU per rank (the driver's
8-GPU node); tests/test_gpu_distributed.py rehearses the multi-rank LOGIC with gloo on one GPU."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(4,device="cuda")*3
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
parts=[torch.empty_like(t)]; dist.all_gather(parts,t)
torch.cuda.synchronize(); print("rccl ok", t.tolist(), parts[0].tolist())
dist.destroy_process_group()
