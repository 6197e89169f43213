"""Does a replayed HIP graph run parallel branches (fork / join through side streams during capture) concurrently?
Test 1: two (three) chains of n tiny DEPENDENT kernels -- no contention for CUs, so concurrent branches cost one chain.
Test 2: the same chains as separate graphs replayed on separate streams (what the sampler graph does)."""
import os, sys, time, torch
dev = 'cuda:0'
N = 40
bufs = [torch.randn(4096, device=dev) for _ in range(3)]
sides = [torch.cuda.Stream() for _ in range(2)]

def chain(b):
    for _ in range(N):
        b.mul_(1.0001)

def work(nbranch, fork):
    cur = torch.cuda.current_stream()
    if fork:
        for s in sides[:nbranch - 1]:
            s.wait_stream(cur)
        for k in range(1, nbranch):
            with torch.cuda.stream(sides[k - 1]):
                chain(bufs[k])
        chain(bufs[0])
        for s in sides[:nbranch - 1]:
            cur.wait_stream(s)
    else:
        for k in range(nbranch):
            chain(bufs[k])

def timeit(g, n=200):
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        g()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

def captured(nbranch, fork):
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        work(nbranch, fork); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            work(nbranch, fork)
        return timeit(g.replay)

print('GPU_MAX_HW_QUEUES', os.environ.get('GPU_MAX_HW_QUEUES'), ' DEBUG_HIP_GRAPH*', {k: v for k, v in os.environ.items() if 'GRAPH' in k})
print('1 chain of %d                 %.1f us' % (N, captured(1, False)))
for nb in (2, 3):
    print('%d chains, one stream        %.1f us' % (nb, captured(nb, False)))
    print('%d chains, forked in a graph %.1f us' % (nb, captured(nb, True)))
# separate graphs on separate streams
gs, ss = [], [torch.cuda.Stream() for _ in range(3)]
for k in range(3):
    with torch.cuda.stream(ss[k]):
        chain(bufs[k]); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=ss[k]):
            chain(bufs[k])
        gs.append(g)
def multi(nb):
    def f():
        for k in range(nb):
            with torch.cuda.stream(ss[k]):
                gs[k].replay()
    return f
for nb in (1, 2, 3):
    print('%d separate graphs on %d streams %.1f us' % (nb, nb, timeit(multi(nb))))
