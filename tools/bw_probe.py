"""What does this card stream? Read-only (sum), copy (read+write) and fill (write-only) rates of plain torch kernels over buffers of the
sizes the hot path touches — the practical ceiling the kernels' GB/s figures in profiles/ should be read against (peak 8 TB/s)."""
import torch, json
dev = torch.device("cuda", 0)
out = {}
for mb in (200, 600, 3000):
    n = mb * 1000 * 1000 // 4
    x = torch.randint(0, 1 << 20, (n,), dtype=torch.int32, device=dev)
    y = torch.empty_like(x)
    def timeit(f, reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    t_sum = timeit(lambda: x.sum())
    t_max = timeit(lambda: x.max())
    t_copy = timeit(lambda: y.copy_(x))
    t_fill = timeit(lambda: y.fill_(7))
    t_add = timeit(lambda: torch.add(x, 1, out=y))
    out[f"{mb}MB"] = {"read_sum_GBps": round(mb / t_sum, 1), "read_max_GBps": round(mb / t_max, 1), "copy_GBps_rw": round(2 * mb / t_copy, 1),
                      "fill_GBps": round(mb / t_fill, 1), "add_GBps_rw": round(2 * mb / t_add, 1)}
    del x, y
print(json.dumps(out))
