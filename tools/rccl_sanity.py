import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.arange(10, dtype=torch.float64, device="cuda"); dist.all_reduce(t); print("allreduce f64 ok", t.sum().item())
x=torch.ones(4,3,3,device="cuda"); full=torch.empty(4,3,3,device="cuda"); dist.all_gather_into_tensor(full,x); print("all_gather_into_tensor ok")
n=torch.tensor([5],dtype=torch.int64,device="cuda"); l=[torch.zeros_like(n)]; dist.all_gather(l,n); print("all_gather int64 ok", l[0].item())
m=torch.tensor([1],dtype=torch.int32,device="cuda"); dist.all_reduce(m,op=dist.ReduceOp.MIN); print("allreduce MIN int32 ok")
dist.barrier(); print("barrier ok")
g=torch.cuda.CUDAGraph(); s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        y = t * 2
torch.cuda.current_stream().wait_stream(s); g.replay(); torch.cuda.synchronize(); print("graph capture with a live communicator ok")
dist.destroy_process_group(); print("done")
