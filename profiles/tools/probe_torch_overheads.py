import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
a = torch.zeros(6_000_000, dtype=torch.uint8, device="cuda"); g = torch.zeros_like(a)
big = torch.randn(8192, 8192, device="cuda")
def busy():
    for _ in range(6): (big @ big)
torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); busy(); t1 = time.perf_counter()
    w = dist.all_gather_into_tensor(g, a, async_op=True); t2 = time.perf_counter()
    w.wait(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("issue busy %.3f ms, all_gather call %.3f ms, wait() %.3f ms, sync %.3f ms, work type %s" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, type(w).__name__))
s = torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    with torch.cuda.stream(s):
        pass
t1 = time.perf_counter()
for _ in range(200):
    e = torch.cuda.Event(); e.record(); s.wait_event(e)
t2 = time.perf_counter()
for _ in range(200):
    w = dist.all_gather_into_tensor(g, a, async_op=True); w.wait()
t3 = time.perf_counter(); torch.cuda.synchronize()
print("stream ctx %.1f us, event+record+wait %.1f us, all_gather+wait %.1f us" % ((t1-t0)/200*1e6, (t2-t1)/200*1e6, (t3-t2)/200*1e6))
