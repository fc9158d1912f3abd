"""Lane-level emulation of tools/conv3/vx_conv3.hip (the 3x3 convolution with GroupNorm + SiLU applied in its A-staging path):
the SAME index formulas as the kernel - plane copies (copy slot -> halo row / columns, slot swizzle on the source side),
the in-place normalisation of the units a thread copied (validity of the halo row, table index), the weight copies and
their XOR swizzle, the K order (32-channel chunk major, taps innermost, two tap-steps per K-tile), the period of nine
K-tiles with its two plane buffers, the fragment addresses of every (phase, fragment, tap) and the accumulator -> output
mapping - executed on a byte image of the LDS in numpy and compared with a float64 convolution of the normalised,
zero-padded input.  Timing (counted waits, barriers) is NOT modelled here: tools/conv3_schedule_check.py proves that.

    python tools/conv3_emulate.py            # W = 32 and W = 64 cases, concat source, top / bottom tiles

Run by tests/test_host_logic.py (CPU).  `permute_weight` is what ops.conv3_weight uses."""
import numpy as np

BM, BN = 256, 320
PIECE, BBUF, PLANE = 8192, 5 * 8192, 25344
PL_OFF, B_OFF = 0, 2 * PLANE
TAB, TAB_OFF = 8192, B_OFF + 2 * BBUF
DUMP_OFF = TAB_OFF + 2 * TAB
LDS_BYTES = DUMP_OFF + 1024


def bf16_round(x):
    """float32 array -> nearest-even bfloat16, returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def permute_weight(w):
    """[n][ky][kx][c] (as [n, 9 c]) -> [n][(chunk, tap, 32)]: w_perm[n][(chunk * 9 + tap) * 32 + i] = w[n][tap][chunk * 32 + i]."""
    n, k = w.shape
    c = k // 9
    return np.ascontiguousarray(w.reshape(n, 9, c // 32, 32).transpose(0, 2, 1, 3).reshape(n, k))


def silu(x):
    return x / (1.0 + np.exp(-x))


class Emu:
    def __init__(self, W, H, frames, c1, c2, n, seed=0, silu_on=True):
        assert W in (64, 32) and (H * W) % BM == 0 and (c1 + c2) % 64 == 0 and c1 % 32 == 0 and n % BN == 0
        rng = np.random.default_rng(seed)
        self.W, self.H, self.frames, self.c1, self.c2, self.n = W, H, frames, c1, c2, n
        self.R, self.WP = BM // W, W + 2
        self.cin = c1 + c2
        self.silu_on = silu_on
        self.x1 = bf16_round(rng.standard_normal((frames, H, W, c1)).astype(np.float32))
        self.x2 = bf16_round(rng.standard_normal((frames, H, W, c2)).astype(np.float32)) if c2 else None
        self.w = bf16_round((rng.standard_normal((n, 9 * self.cin)) * (9 * self.cin) ** -0.5).astype(np.float32))
        self.wp = permute_weight(self.w)
        self.ab = np.zeros((frames, 1024, 2), np.float32)
        self.ab[:, :self.cin, 0] = rng.uniform(0.5, 1.5, (frames, self.cin))
        self.ab[:, :self.cin, 1] = rng.uniform(-0.5, 0.5, (frames, self.cin))
        self.lds = np.zeros(LDS_BYTES // 2, np.float32)           # one float32 per bf16 slot (values are bf16-exact)
        self.out = np.zeros((frames * H * W, n), np.float64)

    # ---- the kernel's formulas -------------------------------------------------------------------------------------
    def hy_of(self, q, wave):
        return (8 * q + wave) // (self.W // 16)

    def hslots(self):
        return (self.R + 2) * (self.W // 16)

    def copy_plane(self, fr, oy0, ci, buf):
        """issue_h for q = 0, 1, 2 and every wave / lane: 16 bytes = 8 channels per lane."""
        W, H, WP = self.W, self.H, self.WP
        first = ci < self.c1
        src = self.x1 if first else self.x2
        cl = ci if first else ci - self.c1
        for q in range(3):
            for wave in range(8):
                hx0 = 1 + 16 * (wave % (W // 16))
                hy = self.hy_of(q, wave)
                iy = min(max(oy0 + hy - 1, 0), H - 1)
                dst = PL_OFF + buf * PLANE + (hy * WP + hx0) * 64 if 8 * q + wave < self.hslots() else DUMP_OFF
                for lane in range(64):
                    hpx = hx0 + (lane >> 2)
                    hkg = (lane & 3) ^ (((hpx >> 2) & 1) << 1)
                    vals = src[fr, iy, hpx - 1, cl + hkg * 8: cl + hkg * 8 + 8]
                    a = (dst + lane * 16) // 2
                    self.lds[a:a + 8] = vals

    def transform_plane(self, fr, oy0, ci, buf):
        W, H, WP = self.W, self.H, self.WP
        for q in range(3):
            for wave in range(8):
                if 8 * q + wave >= self.hslots():
                    continue
                hx0 = 1 + 16 * (wave % (W // 16))
                hy = self.hy_of(q, wave)
                iy = oy0 + hy - 1
                for lane in range(64):
                    hpx = hx0 + (lane >> 2)
                    hkg = (lane & 3) ^ (((hpx >> 2) & 1) << 1)
                    a = (PL_OFF + buf * PLANE + (hy * WP + hx0) * 64 + lane * 16) // 2
                    if iy < 0 or iy >= H:
                        self.lds[a:a + 8] = 0.0
                        continue
                    t = self.ab[fr, ci + hkg * 8: ci + hkg * 8 + 8]
                    # one fused multiply-add in float32 (the kernel's v_pk_fma_f32): exact product and sum, one rounding
                    v = (self.lds[a:a + 8].astype(np.float64) * t[:, 0].astype(np.float64) + t[:, 1].astype(np.float64)).astype(np.float32)
                    if self.silu_on:
                        v = silu(v.astype(np.float64)).astype(np.float32)
                    self.lds[a:a + 8] = bf16_round(v.astype(np.float32))

    def copy_b(self, tile_n, kt, bbuf):
        """the five weight pieces of K-tile kt: thread (r0, slot) copies K chunk slot ^ ((r0 >> 1) & 7) of row 64 q + r0."""
        kbytes = 9 * self.cin * 2
        for q in range(5):
            for tid in range(512):
                r0 = tid >> 3
                cc = (tid & 7) ^ ((r0 >> 1) & 7)
                row = tile_n * BN + 64 * q + r0
                k0 = (kt * 128 + cc * 16) // 2
                a = (B_OFF + bbuf + q * PIECE + (tid >> 6) * 1024 + (tid & 63) * 16) // 2
                self.lds[a:a + 8] = self.wp[row, k0:k0 + 8]
                assert kbytes >= kt * 128 + 128

    def a_const(self, hf, s):
        ml = 64 * hf + 16 * s
        return ((ml // self.W) * self.WP + (ml % self.W)) * 64

    def run(self):
        W, H, WP, R = self.W, self.H, self.WP, self.R
        hw = H * W
        n_tiles = self.n // BN
        total_tiles = (self.frames * hw // BM) * n_tiles
        NP = self.cin // 64
        lanes = np.arange(64)
        frow, fgrp = lanes & 15, lanes >> 4
        sw = (frow >> 1) & 7
        ck = [((fgrp ^ sw) << 4), (((4 + fgrp) ^ sw) << 4)]
        a_slot = [(fgrp ^ ((((frow + kx) >> 2) & 1) << 1)) << 4 for kx in range(3)]
        for lid in range(total_tiles):
            tile_m, tile_n = divmod(lid, n_tiles)
            m0 = tile_m * BM
            fr = m0 // hw
            oy0 = (m0 - fr * hw) // W
            acc = np.zeros((8, 8, 5, 64, 4), np.float64)            # [wave][i][j][lane][r]
            u = 0
            # plane of chunk 0 (the prologue / the previous tile's last period copies and normalises it)
            self.copy_plane(fr, oy0, 0, 0)
            self.transform_plane(fr, oy0, 0, 0)
            for pp in range(NP):
                for v in range(9):
                    if v == 0:      # the odd chunk of this period (kernel: copies at v = 0, normalised at v = 2, 2, 3)
                        self.copy_plane(fr, oy0, (2 * pp + 1) * 32, 1)
                        self.transform_plane(fr, oy0, (2 * pp + 1) * 32, 1)
                    bbuf = BBUF if (u & 1) else 0
                    self.copy_b(tile_n, 9 * pp + v, bbuf)
                    for ph in range(4):
                        kk, hf = ph >> 1, ph & 1
                        ts = 2 * v + kk
                        pl = 1 if ts >= 9 else 0
                        tap = ts - 9 * pl
                        ky = (1 if tap >= 3 else 0) + (1 if tap >= 6 else 0)
                        kx = tap - 3 * ky
                        aoff = pl * PLANE + (ky * WP + kx) * 64
                        for wave in range(8):
                            grp, wc = wave >> 2, wave & 3
                            b_rd = B_OFF + bbuf + (wc * 80 + frow) * 128
                            a_lane = PL_OFF + (grp * (R // 2) * WP + frow) * 64
                            bfr = []
                            for j in range(5):
                                addr = (b_rd + j * 2048 + ck[kk]) // 2
                                bfr.append(self.lds[addr[:, None] + np.arange(8)])        # [lane][8]
                            for s in range(4):
                                addr = (a_lane + a_slot[kx] + aoff + self.a_const(hf, s)) // 2
                                af = self.lds[addr[:, None] + np.arange(8)]               # [lane][8]
                                # MFMA 16x16x32 (C^T form): first operand rows = weight rows (lane & 15), second operand
                                # columns = pixels (lane & 15), both hold k = 8 (lane >> 4) .. + 7
                                B = af.reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32).astype(np.float64)
                                for j in range(5):
                                    A = bfr[j].reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32).astype(np.float64)
                                    D = B @ A.T                                           # [pixel][weight row]
                                    # lane l owns D[l & 15][4 (l >> 4) .. + 3]
                                    acc[wave, 4 * hf + s, j] += D.reshape(16, 4, 4).transpose(1, 0, 2).reshape(64, 4)
                    if v == 5 and pp + 1 < NP:      # the even chunk of the next period (copies at v = 5, normalised at 7, 7, 8)
                        self.copy_plane(fr, oy0, (2 * pp + 2) * 32, 0)
                        self.transform_plane(fr, oy0, (2 * pp + 2) * 32, 0)
                    u += 1
            for wave in range(8):
                grp, wc = wave >> 2, wave & 3
                for i in range(8):
                    for j in range(5):
                        blk = acc[wave, i, j].reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)     # [lrow][4 lq + r]
                        r0 = tile_m * BM + 128 * grp + 16 * i
                        c0 = tile_n * BN + 80 * wc + 16 * j
                        self.out[r0:r0 + 16, c0:c0 + 16] = blk
        return self.out

    def reference(self):
        """float64 convolution of the normalised (bf16-rounded), zero-padded input with the un-permuted weights."""
        x = self.x1 if self.x2 is None else np.concatenate([self.x1, self.x2], axis=-1)
        a = x.astype(np.float64) * self.ab[:, None, None, :self.cin, 0] + self.ab[:, None, None, :self.cin, 1]
        # the kernel computes x * scale + shift as ONE fused multiply-add in float32, then SiLU in float32
        a32 = (x.astype(np.float64) * self.ab[:, None, None, :self.cin, 0].astype(np.float64) +
               self.ab[:, None, None, :self.cin, 1].astype(np.float64)).astype(np.float32)
        if self.silu_on:
            a32 = silu(a32.astype(np.float64)).astype(np.float32)
        a = bf16_round(a32).astype(np.float64)
        pad = np.zeros((self.frames, self.H + 2, self.W + 2, self.cin))
        pad[:, 1:-1, 1:-1] = a
        w = self.w.astype(np.float64).reshape(self.n, 3, 3, self.cin)
        out = np.zeros((self.frames, self.H, self.W, self.n))
        for ky in range(3):
            for kx in range(3):
                out += pad[:, ky:ky + self.H, kx:kx + self.W] @ w[:, ky, kx].T
        return out.reshape(-1, self.n)


# ds_read_b128 is serviced in four NON-contiguous groups of 16 lanes (MI355X_MICROARCH.md, LDS); bank of byte address a =
# (a / 4) % 64; each extra distinct address on a busy bank within a group costs one more LDS cycle
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def b128_conflicts(addrs):
    """extra LDS cycles of one ds_read_b128 given the 64 lanes' byte addresses"""
    extra = 0
    for g in B128_GROUPS:
        banks = {}
        for lane in g:
            banks.setdefault((int(addrs[lane]) // 16) % 16, set()).add(int(addrs[lane]))
        extra += sum(len(v) - 1 for v in banks.values())
    return extra


def fragment_read_conflicts(W, swizzle="pair"):
    """extra LDS cycles over every activation-fragment read of a period (all taps, row halves, fragments, wave rows) and
    every weight-fragment read, with the kernel's address formulas: must be 0.  swizzle="quad" = the first version's
    (hx >> 2) & 3, kept to show what it cost."""
    R, WP = BM // W, W + 2
    lanes = np.arange(64)
    frow, fgrp = lanes & 15, lanes >> 4
    if swizzle == "pair":
        a_slot = [(fgrp ^ ((((frow + kx) >> 2) & 1) << 1)) << 4 for kx in range(3)]
    else:
        a_slot = [(fgrp ^ (((frow + kx) >> 2) & 3)) << 4 for kx in range(3)]
    extra = 0
    for grp in range(2):
        a_lane = PL_OFF + (grp * (R // 2) * WP + frow) * 64
        for tap in range(9):
            ky, kx = divmod(tap, 3)
            for hf in range(2):
                for s in range(4):
                    ml = 64 * hf + 16 * s
                    extra += b128_conflicts(a_lane + a_slot[kx] + (ky * WP + kx) * 64 + ((ml // W) * WP + (ml % W)) * 64)
    sw = (frow >> 1) & 7
    for wc in range(4):
        for j in range(5):
            for ck in (((fgrp ^ sw) << 4), (((4 + fgrp) ^ sw) << 4)):
                extra += b128_conflicts(B_OFF + (wc * 80 + frow) * 128 + j * 2048 + ck)
    return extra


def check(W, H, frames, c1, c2, n, seed=0):
    e = Emu(W, H, frames, c1, c2, n, seed)
    got, want = e.run(), e.reference()
    err = np.abs(got - want).max() / np.abs(want).max()
    return err


def main():
    for W, H, frames, c1, c2, n in ((32, 8, 2, 64, 0, 320), (32, 16, 1, 32, 96, 320), (64, 8, 1, 64, 0, 320)):
        err = check(W, H, frames, c1, c2, n)
        print(f"W={W} H={H} frames={frames} c1={c1} c2={c2} n={n}: max rel err {err:.3e}")
        assert err < 1e-6, err
    for W in (64, 32):
        c, c_old = fragment_read_conflicts(W), fragment_read_conflicts(W, "quad")
        print(f"W={W}: extra LDS cycles from bank conflicts over all fragment reads: {c} (first version's swizzle: {c_old} over 184 reads)")
        assert c == 0
    print("conv3 emulation ok")


if __name__ == "__main__":
    main()
