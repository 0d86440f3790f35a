"""Does a forked branch of a captured hipGraph run concurrently with its sibling?  Two independent chains of K small
kernels: serial on one stream; fork A = side chain captured first, main chain second; fork B = main chain captured first
(fork event recorded before it), side chain second.  Prints replay times; real concurrency => about half of serial."""
import torch, time, sys
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048 * 128
a = torch.randn(n, device=dev); b = torch.randn(n, device=dev)
side = torch.cuda.Stream()

def chain(t):
    for _ in range(K):
        t = torch.sin(t) * 1.0001 + 0.1
    return t

def run(mode):
    cur = torch.cuda.current_stream()
    if mode == "A":
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y = chain(b)
        x = chain(a)
        cur.wait_stream(side)
    elif mode == "B":
        e = torch.cuda.Event(); e.record(cur)
        x = chain(a)
        side.wait_event(e)
        with torch.cuda.stream(side):
            y = chain(b)
        cur.wait_stream(side)
    else:
        y = chain(b); x = chain(a)
    return x + y

for mode in ("serial", "A", "B"):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): run(mode)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = run(mode)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        print(f"{mode:6s}: {(time.perf_counter()-t0)/20*1e6:.1f} us per replay ({3*2*K} kernels, n={n})")
