import torch, time
dev = torch.device("cuda", 0)
hin = torch.empty(98 << 20, dtype=torch.uint8).pin_memory()
hout = torch.empty(268 << 20, dtype=torch.uint8).pin_memory()
din = torch.empty(98 << 20, dtype=torch.uint8, device=dev)
dout = torch.empty(268 << 20, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def serial():
    din.copy_(hin, non_blocking=True); hout.copy_(dout, non_blocking=True)
def duplex():
    with torch.cuda.stream(s1): din.copy_(hin, non_blocking=True)
    with torch.cuda.stream(s2): hout.copy_(dout, non_blocking=True)
def h2d(): din.copy_(hin, non_blocking=True)
def d2h(): hout.copy_(dout, non_blocking=True)
print("h2d 98MB %.2f ms (%.1f GB/s)  d2h 268MB %.2f ms (%.1f GB/s)" % (t(h2d), 98*1.048576/t(h2d), t(d2h), 268*1.048576/t(d2h)))
print("serial %.2f ms   duplex (2 streams) %.2f ms" % (t(serial), t(duplex)))
# pageable
pin = torch.empty(98 << 20, dtype=torch.uint8); pout = torch.empty(268 << 20, dtype=torch.uint8)
def pser(): din.copy_(pin); pout.copy_(dout)
print("pageable serial %.2f ms" % t(pser, 3))
