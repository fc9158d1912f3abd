"""Does a hipGraph shorten the gaps between DEPENDENT small launches on this stack?  A chain of 8x8-level linears
(2048 x 1280 x 1280, each reading the previous output) run eagerly through the ctypes path and as a captured graph:
    python tools/graph_gap_probe.py [chain length]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v_express_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = torch.Generator().manual_seed(0)
    for m, c in ((2048, 1280), (8192, 1280), (32768, 640)):
        x = (torch.randn(m, c, generator=g)).to("cuda").to(BF)
        w = (torch.randn(c, c, generator=g) * c ** -0.5).to("cuda").to(BF)
        bufs = [torch.empty_like(x), torch.empty_like(x)]

        def chain():
            src = x
            for i in range(n):
                dst = bufs[i & 1]
                ops.gemm(src, w, None, out=dst)
                src = dst
        chain()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            chain()
        e1.record()
        torch.cuda.synchronize()
        eager = 1e3 * e0.elapsed_time(e1) / (3 * n)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            chain()
        graph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        gr = 1e3 * e0.elapsed_time(e1) / (3 * n)
        print(f"{m:6d} x {c} x {c}: chain of {n} dependent launches  eager {eager:7.2f} us / launch   graph {gr:7.2f} us / launch")


if __name__ == "__main__":
    main()
