import time, torch
d = torch.randn(256, 256, device="cuda:0"); h = torch.empty(256, 256); hp = torch.empty(256, 256, pin_memory=True)
big = torch.randn(6000, 784, device="cuda:0"); hb = torch.empty(6000, 784); hbp = torch.empty(6000, 784, pin_memory=True)
def t(fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("threads", torch.get_num_threads())
print("small D2H  pageable copy_ : %.3f ms" % t(lambda: h.copy_(d)))
print("small D2H  .cpu()          : %.3f ms" % t(lambda: d.cpu()))
print("small D2H  pinned copy_    : %.3f ms" % t(lambda: (hp.copy_(d, non_blocking=True), torch.cuda.synchronize())))
print("small H2D  pageable .to    : %.3f ms" % t(lambda: h.to("cuda:0")))
print("19 MB D2H  pageable copy_ : %.3f ms" % t(lambda: hb.copy_(big), 10))
print("19 MB D2H  pinned copy_    : %.3f ms" % t(lambda: (hbp.copy_(big, non_blocking=True), torch.cuda.synchronize()), 10))
print("19 MB H2D  pageable .to    : %.3f ms" % t(lambda: hb.to("cuda:0"), 10))
print("19 MB H2D  pinned .to      : %.3f ms" % t(lambda: hbp.to("cuda:0", non_blocking=True), 10))
p = torch.nn.Parameter(torch.empty(256, 256))
print("small D2H  into Parameter.data copy_: %.3f ms" % t(lambda: p.data.copy_(d)))
