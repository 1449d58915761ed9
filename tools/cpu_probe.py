import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
a = torch.randn(512, 1024); b = torch.randn(1024, 4096)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    for _ in range(2): (a @ b)
    t = time.time()
    for _ in range(10): (a @ b)
    dt = (time.time() - t) / 10
    print(n, "threads: %.2f ms  %.2f TFLOP/s" % (dt * 1e3, 2 * 512 * 1024 * 4096 / dt / 1e12), flush=True)
