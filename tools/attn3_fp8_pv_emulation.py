"""attn3 lever (b) of VERDICT r05 item 4, measured before any kernel work: the P.V product of the d = 40 spatial attention with
P and V^T in OCP e4m3 (v_mfma_scale_f32_16x16x128_f8f6f4 would halve the PV matrix cycles) - does it hold the EXISTING bf16
tolerance of the kernel tests (tests/test_gpu_kernels.py::check for attention: max|err| <= 2^-6 max|ref|, rel-L2 <= 1e-2) and
of the model-level goldens?  CPU emulation of exactly what the kernel would compute, on the kernel tests' own input
distributions and on a peaked-softmax case:

    O[q, :] = (sum_k p8[q, k] * v8[k, :]) * (sv[:] / sp) / l[q],   p = exp(s - m) in [0, 1],  l = sum_k p (fp32, unrounded)
    bf16 form (shipped):  p rounded to bf16, V in bf16
    e4m3 form:            p8 = e4m3(p * sp)  (sp = 2^8: p = 1 -> 256, the format's normals reach down to p = 2^-14),
                          v8 = e4m3(V / sv), sv = amax per (head, channel) / 448     (both: round-to-nearest-even, saturating)

Relative error of an e4m3 value: 2^-4 worst case, ~2.6e-2 rms - against bf16's 2^-9 / 1.1e-3.  The products' errors are
independent across keys, so the output's relative L2 error is ~ the per-term rms (it does NOT average out: signal and noise
both grow with sqrt(keys)).  Prints the figures; exit code 0 always (a measurement, not a test).
    python tools/attn3_fp8_pv_emulation.py > profiles/r06a_attn3_e4m3_pv_emulation.txt"""
import torch

torch.manual_seed(0)
E4 = torch.float8_e4m3fn


def to_e4m3(x):
    return x.clamp(-448.0, 448.0).to(E4).float()


def run(name, q, k, v):
    """q, k, v: [B, H, N, d] float32 holding bf16-representable values"""
    d = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * d ** -0.5
    m = s.amax(dim=-1, keepdim=True)
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    ref = (p.double() @ v.double() / l.double()).float()
    out_bf = (p.bfloat16().float() @ v) / l
    sp = 256.0
    sv = v.abs().amax(dim=-2, keepdim=True) / 448.0                      # per (batch, head, channel)
    out_f8 = (to_e4m3(p * sp) @ to_e4m3(v / sv)) * (sv / sp) / l
    out_f8p = (to_e4m3(p * sp) @ v) / sp / l                              # only P in e4m3 (no such MFMA: shows the split)
    out_f8v = (p.bfloat16().float() @ to_e4m3(v / sv)) * sv / l           # only V in e4m3
    scale = ref.abs().max().item()

    def fig(o):
        e = (o - ref).abs()
        return e.max().item() / scale, (e.pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()
    print(f"{name}")
    for tag, o in (("bf16 P, bf16 V (shipped)", out_bf), ("e4m3 P, e4m3 V", out_f8), ("e4m3 P only", out_f8p), ("e4m3 V only", out_f8v)):
        mx, rl = fig(o)
        ok = mx <= 2 ** -6 and rl <= 1e-2
        print(f"    {tag:26s} max|err| / max|ref| = {mx:9.3e} (bound 1.56e-2)   rel-L2 = {rl:9.3e} (bound 1e-2)   {'within' if ok else 'OUTSIDE'} the kernel tolerance")


def bf(x):
    return x.bfloat16().float()


def main():
    print(__doc__.split("Prints")[0])
    B, H, N, d = 1, 8, 4096, 40
    g = torch.Generator().manual_seed(1)
    # (a) the kernel tests' distribution: unit-normal q, k, v (scores ~ N(0, 1): a flat softmax over 4096 keys)
    q, k, v = (bf(torch.randn(B, H, N, d, generator=g)) for _ in range(3))
    run(f"(a) kernel-test inputs: q, k, v ~ N(0, 1), {H} heads x {N} tokens, d = {d}", q[:, :, :512], k, v)
    # (b) a peaked softmax (scores ~ N(0, 4^2)): a few keys carry the row, as in trained attention maps
    run("(b) peaked softmax: q scaled x4 (a handful of keys per row carry the weight)", 4 * q[:, :, :512], k, v)
    # (c) values with per-channel structure (means, outlier channels), as residual-stream activations have
    v2 = bf(v * (1 + 3 * torch.rand(1, H, 1, d, generator=g)) + torch.randn(1, H, 1, d, generator=g))
    run("(c) values with per-channel means / scales (activations), flat softmax", q[:, :, :512], k, v2)
    print("\nReading: on zero-mean values - the kernel tests' own inputs (a), and what LN(h) W_v of a random-init model produces "
          "- the e4m3 form has rel-L2 2.9 - 3.7e-2:\n3 - 4x OUTSIDE the attention kernel's tolerance (1e-2), and the whole budget of a "
          "CFG forward (3e-2) from ONE of its 32 attention sites; each operand alone\nis already outside (P 1.1 - 2.5e-2, V 2.6e-2).  "
          "It is inside (3e-3) only when per-channel MEANS dominate the values (c): the error is relative to the\nfluctuating part of V, "
          "and scales (per row, per channel) cannot help - it is the 3-bit mantissa, not the range.  Lever (b) is closed on parity\n"
          "grounds before any kernel work: it would need its own, 4x looser tolerance, which north_star's \"stated fp16 tolerance\" does "
          "not allow.")


if __name__ == "__main__":
    main()
