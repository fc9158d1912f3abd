"""Yardstick only (never the product path): torch.matmul in bf16 (hipBLASLt / rocBLAS behind PyTorch-ROCm) on the
heaviest GEMM shapes of one CFG UNet3D forward + decode (profiles/r02j_gemm_by_shape.txt), as plain [M, K] x [N, K]^T
products on uniform random data, next to what vx_gemm reaches on the same box for the same (M, N, K) as a plain linear
(no conv gather, bias only).  Says which shapes have headroom.

    python tools/blaslt_yardstick.py [reps]      -> table on stdout"""
import sys

import torch

sys.path.insert(0, ".")
SHAPES = [  # (M, N, K, note)
    (131072, 2560, 320, "L0 GEGLU proj (ours: GEGLU epilogue)"),
    (131072, 320, 320, "L0 linear"),
    (32768, 5120, 640, "L1 GEGLU proj"),
    (8192, 10240, 1280, "L2 GEGLU proj"),
    (32768, 640, 640, "L1 linear"),
    (131072, 320, 2880, "L0 conv3x3 as GEMM"),
    (8192, 1280, 1280, "L2 linear"),
    (131072, 320, 1280, "L0 FF out"),
    (8192, 1280, 11520, "L2 conv3x3 as GEMM"),
    (131072, 960, 320, "L0 qkv"),
    (32768, 640, 5760, "L1 conv3x3 as GEMM"),
    (8192, 1280, 5120, "L2 FF out"),
    (32768, 640, 2560, "L1 FF out"),
    (2048, 1280, 11520, "L3 conv3x3 as GEMM"),
    (2048, 1280, 1280, "L3 linear"),
    (4096, 1280, 1280, "L2 linear, one CFG half"),
    (1048576, 128, 1152, "VAE 512^2 conv as GEMM"),
    (262144, 256, 2304, "VAE 256^2 conv as GEMM"),
    (65536, 512, 4608, "VAE 128^2 conv as GEMM"),
]


def timed(fn, reps):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    from v_express_amd import ops
    dev = "cuda"
    print(f"{'M':>8} {'N':>6} {'K':>6} {'blaslt us':>10} {'TF/s':>8} {'vx_gemm us':>11} {'TF/s':>8}  note")
    for m, n, k, note in SHAPES:
        a = (torch.rand((m, k), device=dev) * 2 - 1).to(torch.bfloat16)
        w = ((torch.rand((n, k), device=dev) * 2 - 1) * k ** -0.5).to(torch.bfloat16)
        b = torch.zeros(n, device=dev)
        out = torch.empty((m, n), device=dev, dtype=torch.bfloat16)
        t_ref = timed(lambda: torch.matmul(a, w.t(), out=out), reps)
        with ops.frame_rows(m // 32 if m % 32 == 0 else m, items=2):
            with ops.GemmProfile() as prof:
                ops.gemm(a, w, b, out=out)
            t_vx = timed(lambda: ops.gemm(a, w, b, out=out), reps)
        kern = prof.records[0][3].replace("gemm_", "").replace("_kernel", "")
        fl = 2.0 * m * n * k
        print(f"{m:8d} {n:6d} {k:6d} {t_ref:10.1f} {fl / t_ref * 1e-6:8.1f} {t_vx:11.1f} {fl / t_vx * 1e-6:8.1f}  {note}  [{kern}]")
        del a, w, out


if __name__ == "__main__":
    main()
