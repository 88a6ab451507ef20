"""Does torch.distributed.gather on an RCCL group of ONE rank work, and what does a call cost? (round 6)"""
import os, time, socket
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
import torch, torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
g = dist.new_group(backend="nccl")
x = torch.ones(1, device=dev); dist.all_reduce(x, group=g); torch.cuda.synchronize()
print("bring-up s", time.perf_counter() - t0, float(x))
for nbytes in (500_000, 1_400_000, 4_000_000):
    src = [torch.full((nbytes,), k + 1, dtype=torch.uint8, device=dev) for k in range(2)]
    rcv = [[torch.empty(nbytes, dtype=torch.uint8, device=dev)] for _ in range(2)]
    for k in range(4):
        w = dist.gather(src[k & 1], rcv[k & 1], dst=0, group=g, async_op=True); w.wait()
    torch.cuda.synchronize()
    assert int(rcv[0][0][0]) == 1 and int(rcv[1][0][-1]) == 2
    K = 300
    t0 = time.perf_counter()
    for k in range(K):
        w = dist.gather(src[k & 1], rcv[k & 1], dst=0, group=g, async_op=True); w.wait()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{nbytes} B: host enqueue {1e6*(t1-t0)/K:.1f} us/call, wall {1e6*(t2-t0)/K:.1f} us/call")
dist.destroy_process_group()
print("ok")
