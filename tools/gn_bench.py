"""GroupNorm micro-benchmark at the clip's shapes (GPU box): the statistics pass + apply pass (`ops.groupnorm`) and the apply
pass alone, rotating over enough buffers that no launch finds its input in the L2s (the Infinity Cache holds 256 MB: the
rotation is 1 GB).  Prints us per launch and algorithmic GB/s (stats+apply: 6 B per element, apply: 4 B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import v_express_amd as vx  # noqa: F401
    from v_express_amd import ops
    dev = torch.device("cuda", 0)
    shapes = [(32, 64, 64, 320, 0), (32, 64, 64, 640, 320), (32, 64, 64, 320, 320), (32, 32, 32, 640, 0),
              (32, 32, 32, 1280, 640), (32, 16, 16, 1280, 0), (32, 16, 16, 1280, 1280), (32, 8, 8, 1280, 0)]
    reps = int(os.environ.get("GN_REPS", "30"))
    for frames, H, W, c1, c2 in shapes:
        hw, c = H * W, c1 + c2
        nbuf = max(2, min(16, int(1e9 // (frames * hw * c * 2))))
        xs = [torch.randn(frames, hw, c1, device=dev).to(ops.BF16) for _ in range(nbuf)]
        x2 = [torch.randn(frames, hw, c2, device=dev).to(ops.BF16) for _ in range(nbuf)] if c2 else None
        g = torch.rand(c, device=dev) + 0.5
        b = torch.randn(c, device=dev)
        res = {}
        for pad in (True,):
            kw = dict(frames=frames, hw=hw, groups=32, eps=1e-5, silu=True, pad_hw=(H, W) if pad else None)
            # stats + apply
            for i in range(3):
                ops.groupnorm(xs[i % nbuf], g, b, x2=x2[i % nbuf] if x2 else None, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                ops.groupnorm(xs[i % nbuf], g, b, x2=x2[i % nbuf] if x2 else None, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            res["stats+apply"] = (us, 6 * frames * hw * c / us / 1e3)
            if not c2:
                # apply alone: statistics attached as a producer would leave them
                ws, slabs = ops.groupnorm_stats(xs[0], frames=frames, hw=hw, groups=32)
                for x in xs:
                    x._vx_gn = ops.GnStats(ws, slabs, 32, frames, hw, c1)
                for i in range(3):
                    ops.groupnorm(xs[i % nbuf], g, b, **kw)
                torch.cuda.synchronize()
                e0.record()
                for i in range(reps):
                    ops.groupnorm(xs[i % nbuf], g, b, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / reps
                res["apply"] = (us, 4 * frames * hw * c / us / 1e3)
                e0.record()
                for i in range(reps):
                    ops.groupnorm_stats(ops_no_gn(xs[i % nbuf]), frames=frames, hw=hw, groups=32)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / reps
                res["stats"] = (us, 2 * frames * hw * c / us / 1e3)
        print(f"{frames}x{H}x{W} c={c1}+{c2}: " + "  ".join(f"{k} {v[0]:7.1f} us {v[1]:6.0f} GB/s" for k, v in res.items()),
              flush=True)
        del xs, x2
        ops.clear_caches()
        torch.cuda.empty_cache()


def ops_no_gn(x):
    y = x.view(x.shape)      # a fresh tensor object: no producer statistics attached
    return y


if __name__ == "__main__":
    main()
