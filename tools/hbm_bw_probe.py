import torch, time
dev = torch.device("cuda:0")
n = 768 * 1024 * 1024  # 3 GiB of float32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
tw = t(lambda: a.fill_(1.0)); print(f"write-only fill: {n*4/tw/1e12:.2f} TB/s")
tc = t(lambda: b.copy_(a)); print(f"copy (read+write): {2*n*4/tc/1e12:.2f} TB/s total")
tr = t(lambda: a.sum()); print(f"read-only sum: {n*4/tr/1e12:.2f} TB/s")
tm = t(lambda: torch.add(a, 1.0, out=b)); print(f"add out-of-place: {2*n*4/tm/1e12:.2f} TB/s total")
