"""Does a replayed HIP graph run parallel branches (fork / join through a side stream during capture) concurrently?
A long kernel (big elementwise pass) on the capture stream, n short dependent kernels on a forked stream; replay time of
the forked graph vs the same work captured on one stream."""
import torch, time
dev = 'cuda:0'
big = torch.randn(64 << 20, device=dev)          # 256 MB: ~100+ us per pass
out = torch.empty_like(big)
small = [torch.randn(4096, device=dev) for _ in range(12)]
side = torch.cuda.Stream()

def work(fork):
    cur = torch.cuda.current_stream()
    if fork:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for s in small:
                s.mul_(1.0001).add_(1e-6)          # 2 tiny launches each, a dependent chain
    torch.add(big, 1.0, out=out)
    torch.mul(out, 0.5, out=big)
    if fork:
        cur.wait_stream(side)
    else:
        for s in small:
            s.mul_(1.0001).add_(1e-6)

def bench(fork):
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        work(fork); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            work(fork)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / 200 * 1e6

def only(which):
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            if which == 'big':
                torch.add(big, 1.0, out=out); torch.mul(out, 0.5, out=big)
            else:
                for s in small:
                    s.mul_(1.0001).add_(1e-6)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / 200 * 1e6

print('big alone %.1f us   small chain alone %.1f us' % (only('big'), only('small')))
print('one stream  %.1f us' % bench(False))
print('forked      %.1f us' % bench(True))
print('one stream  %.1f us' % bench(False))
print('forked      %.1f us' % bench(True))
