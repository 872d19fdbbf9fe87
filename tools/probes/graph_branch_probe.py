"""Do independent branches of a captured HIP graph run CONCURRENTLY on this ROCm?  Two / four single-block spin kernels
(torch.cuda._sleep) on forked streams inside one capture vs the same kernels serialized on one stream."""
import time
import torch

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
CYC = 200000            # ~100 us at 2.1 GHz


def build(nbranch, fork):
    main = torch.cuda.Stream()
    sides = [torch.cuda.Stream() for _ in range(nbranch - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        torch.cuda._sleep(1000)
        main.synchronize()
        with torch.cuda.graph(g, stream=main):
            if fork:
                ev = torch.cuda.Event()
                ev.record(main)
                for s in sides:
                    s.wait_event(ev)
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(CYC)
                torch.cuda._sleep(CYC)
                for s in sides:
                    e2 = torch.cuda.Event()
                    e2.record(s)
                    main.wait_event(e2)
            else:
                for _ in range(nbranch):
                    torch.cuda._sleep(CYC)
    return g


for nb in (2, 4):
    for fork in (False, True):
        g = build(nb, fork)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print("branches %d %-10s: %.1f us per replay" % (nb, "forked" if fork else "serial", (time.perf_counter() - t0) / 20 * 1e6))
# eager, two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s1):
        torch.cuda._sleep(CYC)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(CYC)
torch.cuda.synchronize()
print("eager two streams: %.1f us per pair" % ((time.perf_counter() - t0) / 20 * 1e6))
