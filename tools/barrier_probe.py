import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
dist.init_process_group("gloo")
t = torch.zeros(1, dtype=torch.float64)
for name, fn in (("barrier", dist.barrier), ("all_reduce", lambda: dist.all_reduce(t))):
    fn(); ts = []
    for _ in range(10):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "ms:", " ".join("%.3f" % x for x in ts))
dist.destroy_process_group()
