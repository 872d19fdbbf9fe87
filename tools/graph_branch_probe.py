"""Do two independent kernels on forked streams inside ONE captured HIP graph overlap on MI355X / ROCm 7.2?"""
import torch, time
dev = torch.device("cuda", 0)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(1, device=dev)
CY = 40000   # ~20 us of spinning per kernel

def body(fork):
    torch.cuda._sleep(CY)
    if fork:
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(s_side):
            s_side.wait_event(ev)
            torch.cuda._sleep(CY)
            ev2 = torch.cuda.Event(); ev2.record()
        torch.cuda._sleep(CY)
        torch.cuda.current_stream().wait_event(ev2)
    else:
        torch.cuda._sleep(CY)
        torch.cuda._sleep(CY)
    torch.cuda._sleep(CY)

for fork in (False, True):
    with torch.cuda.stream(s_main):
        body(fork); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s_main, capture_error_mode="thread_local"):
            body(fork)
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("fork" if fork else "serial", "graph replay: %.2f us per replay (4 sleeps of ~%d cycles)" % (e0.elapsed_time(e1) * 1e3 / 50, CY))
        # eager streams
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): body(fork)
        torch.cuda.synchronize()
        print("   eager: %.2f us" % ((time.perf_counter() - t) * 1e6 / 50))
