"""Lane-level emulation of csrc/vx_tblock.hip (the fused temporal attention block) in numpy: the pack layouts, the
fragment addresses, the operand roles of every MFMA and the LDS hand-over of O^T, written with the SAME index formulas as
the kernel, run for one tile against a plain float64 statement of the block.  No GPU needed: this is how the index
arithmetic of the kernel was checked before it ever ran (tests/test_host_logic.py runs it).

MFMA lane layouts (gfx950, as used by every kernel of this library):
  16x16x32:  A lane l holds A[m = l & 15][k = 8 (l >> 4) + 0..7],  B lane l holds B[k = 8 (l >> 4) + 0..7][n = l & 15],
             D lane l holds D[m = 4 (l >> 4) + 0..3][n = l & 15]
  16x16x16:  the same with 4 k per lane: k = 4 (l >> 4) + 0..3
"""
import numpy as np

C, HEADS, D, BLOCKS = 320, 8, 40, 8
# window length F = 16: one 16-row block per pixel, 8 pixels per tile; F = 24 (round 5): two blocks per pixel (frames 0-15,
# frames 16-23 + 8 padding rows that repeat frame 23), 4 pixels per tile


def geo(F):
    HB = 2 if F > 16 else 1
    return HB, BLOCKS // HB


def blk_pix(b, HB):
    return b if HB == 1 else b >> 1


def blk_frames(b, HB, F):
    """frame of each of the 16 rows of block b (padding rows re-read the last frame)"""
    fr = np.arange(16) if HB == 1 else 16 * (b & 1) + np.arange(16)
    return np.minimum(fr, F - 1)
KS = C // 32
PCOLS = HEADS * 8 * 16
NW, NPX = 8, 2
SNJ = 20 // (NW // 2)
LANES = np.arange(64)
LROW, LQ = LANES & 15, LANES >> 4


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def mfma(a, b, c, kper):
    """a, b: [64][kper] lane registers, c: [64][4] -> d [64][4]"""
    A = np.zeros((16, 4 * kper), np.float64)
    B = np.zeros((4 * kper, 16), np.float64)
    for e in range(kper):
        A[LROW, kper * LQ + e] = a[:, e]
        B[kper * LQ + e, LROW] = b[:, e]
    Dm = A @ B
    d = c.astype(np.float64).copy()
    for r in range(4):
        d[:, r] += Dm[4 * LQ + r, LROW]
    return d.astype(np.float32)


def src_col(head, blk, r):
    q, k, v = head * D, C + head * D, 2 * C + head * D
    if blk == 0: return q + r
    if blk == 1: return q + 16 + r
    if blk == 2: return k + r
    if blk == 3: return k + 16 + r
    if blk == 4: return q + 32 + r if r < 8 else k + 32 + (r - 8)
    if blk == 5: return v + r
    if blk == 6: return v + 16 + r
    return v + 32 + r if r < 8 else -1


def pack(wqkv, bias, colsum, pe, wo, F=16):
    """tblock_pack_kernel: returns wqkv_t [32 chunks][10][2][64][8] (the weights of a chunk), tab [32 chunks][2 HB][256]
    (the fp32 tables behind them, one per (head of the pair, block half): bias + positional row of frame 16 half + row,
    [row][column] for blocks 0..4, [column][row] for the V blocks; frames >= F: 0), wo_t [4][25600] in elements,
    colsum_p [1024]"""
    HB, _ = geo(F)
    wqkv_t = np.zeros((32, KS, 2, 64, 8), np.float32)
    tab = np.zeros((32, 2 * HB, 256), np.float32)
    for ch in range(32):
        hp, blk = ch >> 3, ch & 7
        for wn in range(2):
            head = 2 * hp + wn
            for lane in range(64):
                col = src_col(head, blk, lane & 15)
                if col < 0:
                    continue
                for ks in range(KS):
                    wqkv_t[ch, ks, wn, lane] = wqkv[col, 32 * ks + 8 * (lane >> 4): 32 * ks + 8 * (lane >> 4) + 8]
            for hh in range(HB):
                for e in range(256):
                    fr, r = (e >> 4, e & 15) if blk < 5 else (e & 15, e >> 4)
                    fr += 16 * hh
                    col = src_col(head, blk, r)
                    if col >= 0 and fr < F:
                        tab[ch, wn * HB + hh, e] = bias[col] + pe[fr, col]
    wo_t = np.zeros((4, 25600), np.float32)          # per head pair: part0 10240 | part1 10240 | part2 5120 elements
    for hp in range(4):
        for part in range(2):
            head = 2 * hp + part
            for j in range(20):
                for lane in range(64):
                    lq = lane >> 4
                    src = wo[16 * j + (lane & 15), head * D:]
                    dst = part * 10240 + (j * 64 + lane) * 8
                    wo_t[hp, dst:dst + 4] = src[4 * lq:4 * lq + 4]
                    wo_t[hp, dst + 4:dst + 8] = src[16 + 4 * lq:16 + 4 * lq + 4]
        for j in range(20):
            for lane in range(64):
                lq = lane >> 4
                head = 2 * hp + (lq >> 1)
                base = head * D + 32 + 4 * (lq & 1)
                dst = 20480 + (j * 64 + lane) * 4
                wo_t[hp, dst:dst + 4] = wo[16 * j + (lane & 15), base:base + 4]
    colsum_p = np.zeros(PCOLS, np.float32)
    for pc in range(PCOLS):
        col = src_col(pc >> 7, (pc >> 4) & 7, pc & 15)
        if col >= 0:
            colsum_p[pc] = colsum[col]
    return wqkv_t, tab, wo_t, colsum_p


def run_tile(x_tile, packed, scale_log2e, eps):
    """x_tile: [PIX pixels][F frames][320] (bf16 values as float32) -> the tile's output rows, same shape"""
    wqkv_t, tab, wo_t, colsum_p = packed
    PIX, F = x_tile.shape[0], x_tile.shape[1]
    HB, pix_ = geo(F)
    assert pix_ == PIX
    o_lds = np.zeros((4, BLOCKS, 1280), np.float32)    # per head pair, block: kb0 512 elements | kb1 512 | kb2 256
    st_lds = np.zeros((BLOCKS, 16, 2), np.float32)
    for wave in range(NW):
        wm, wn = wave >> 1, wave & 1
        xa = np.zeros((NPX, KS, 64, 8), np.float32)
        rs, rm = np.zeros((NPX, 64), np.float32), np.zeros((NPX, 64), np.float32)
        for i in range(NPX):
            b_ = NPX * wm + i
            rows = x_tile[blk_pix(b_, HB)][blk_frames(b_, HB, F)]      # [row of the block][320]
            for ks in range(KS):
                for e in range(8):
                    xa[i, ks, :, e] = rows[LROW, 32 * ks + 8 * LQ + e]
            mean = rows.astype(np.float64).mean(axis=1)
            var = ((rows.astype(np.float64) - mean[:, None]) ** 2).mean(axis=1)
            rstd = 1.0 / np.sqrt(var + eps)
            rs[i] = rstd[LROW]
            rm[i] = (-mean * rstd)[LROW]
            st_lds[NPX * wm + i, :, 0] = rstd
            st_lds[NPX * wm + i, :, 1] = -mean * rstd
        for hp in range(4):
            head = 2 * hp + wn
            Qp = np.zeros((NPX, 2, 64, 4), np.float32)
            Kp = np.zeros((NPX, 2, 64, 4), np.float32)
            Mp = np.zeros((NPX, 64, 4), np.float32)
            Vp = np.zeros((NPX, 3, 64, 4), np.float32)
            for blk in range(8):
                c = 8 * hp + blk
                pc = (head * 8 + blk) * 16
                plain = blk >= 5
                tqh = [np.stack([tab[c, wn * HB + hh, LROW * 16 + 4 * LQ + r] for r in range(4)], axis=1) for hh in range(HB)]
                P = np.zeros((NPX, 64, 4), np.float32)
                for ks in range(KS):
                    wf = wqkv_t[c, ks, wn]               # smem + slot + ks * 2048 + wn * 1024 + lane * 16
                    for i in range(NPX):
                        P[i] = mfma(xa[i, ks], wf, P[i], 8) if plain else mfma(wf, xa[i, ks], P[i], 8)
                if blk < 5:
                    s4 = np.stack([colsum_p[pc + 4 * LQ + r] for r in range(4)], axis=1)
                    for i in range(NPX):
                        tq = tqh[0 if HB == 1 else i]
                        v = rs[i][:, None] * P[i] + (rm[i][:, None] * s4 + tq)
                        v = bf16(v)
                        if blk < 2: Qp[i, blk] = v
                        elif blk < 4: Kp[i, blk - 2] = v
                        else: Mp[i] = v
                else:
                    s1 = colsum_p[pc + LROW]
                    for i in range(NPX):
                        tq = tqh[0 if HB == 1 else i]
                        a = np.stack([st_lds[NPX * wm + i, 4 * LQ + r, 0] for r in range(4)], axis=1)
                        b = np.stack([st_lds[NPX * wm + i, 4 * LQ + r, 1] for r in range(4)], axis=1)
                        Vp[i, blk - 5] = bf16(a * P[i] + (b * s1[:, None] + tq))
            # permlane32_swap(M, 0): [0] = (M lanes 0-31 | 0) = q 32..39, [1] = (M lanes 32-63 moved to 0-31 | 0) = k 32..39
            mq, mk = [], []
            for i in range(NPX):
                mq.append(np.where((LANES < 32)[:, None], Mp[i], 0.0).astype(np.float32))
                km = np.zeros((64, 4), np.float32)
                km[:32] = Mp[i][32:]
                mk.append(km)
            for i in range(NPX):                       # query block i against HB key blocks
                qa = np.concatenate([Qp[i, 0], Qp[i, 1]], axis=1)
                scs = []
                for kbi in range(HB):
                    kb = i if HB == 1 else kbi
                    ka = np.concatenate([Kp[kb, 0], Kp[kb, 1]], axis=1)
                    sc = mfma(ka, qa, np.zeros((64, 4), np.float32), 8)
                    sc = sc + mfma(mk[kb], mq[i], np.zeros((64, 4), np.float32), 4)
                    if HB == 2 and kbi == 1:
                        sc = np.where((LQ >= 2)[:, None], -np.inf, sc).astype(np.float32)   # frames 24 .. 31 do not exist
                    scs.append(sc)
                mx = np.max(np.stack([sc.max(axis=1) for sc in scs]), axis=0)
                mx = np.maximum(mx, mx[LANES ^ 16])
                mx = np.maximum(mx, mx[LANES ^ 32])
                prs = [np.exp2(sc * scale_log2e - (mx * scale_log2e)[:, None]).astype(np.float32) for sc in scs]
                sm = sum(pr.sum(axis=1) for pr in prs)
                sm = sm + sm[LANES ^ 16]
                sm = sm + sm[LANES ^ 32]
                inv_l = (1.0 / sm).astype(np.float32)
                o = []
                for vb in range(3):
                    a = np.zeros((64, 4), np.float32)
                    for kbi in range(HB):
                        a = mfma(Vp[i if HB == 1 else kbi, vb], bf16(prs[kbi]), a, 4)
                    o.append(bf16(a * inv_l[:, None]))
                pix = NPX * wm + i
                for lane in range(64):
                    base = wn * 512 + lane * 8
                    o_lds[hp, pix, base:base + 4] = o[0][lane]
                    o_lds[hp, pix, base + 4:base + 8] = o[1][lane]
                    lq, lrow = lane >> 4, lane & 15
                    if lq < 2:
                        b2 = 1024 + ((2 * wn + lq) * 16 + lrow) * 4
                        o_lds[hp, pix, b2:b2 + 4] = o[2][lane]
    # ---- phase 2
    out = np.zeros_like(x_tile)
    for wave in range(NW):
        s2_pix0, s2_cg = 4 * (wave // (NW // 2)), wave % (NW // 2)
        s2_col0 = 16 * SNJ * s2_cg
        Y = np.zeros((4, SNJ, 64, 4), np.float32)
        for hp2 in range(4):
            for part in range(3):
                for i in range(4):
                    pix = s2_pix0 + i
                    if part < 2:
                        oa = o_lds[hp2, pix, part * 512:(part + 1) * 512].reshape(64, 8)
                    else:
                        oa = o_lds[hp2, pix, 1024:1280].reshape(64, 4)
                    for j in range(SNJ):
                        jj = SNJ * s2_cg + j
                        if part < 2:
                            bw = wo_t[hp2, part * 10240 + jj * 512: part * 10240 + (jj + 1) * 512].reshape(64, 8)
                            Y[i, j] = mfma(bw, oa, Y[i, j], 8)
                        else:
                            bw = wo_t[hp2, 20480 + jj * 256: 20480 + (jj + 1) * 256].reshape(64, 4)
                            Y[i, j] = mfma(bw, oa, Y[i, j], 4)
        for i in range(4):
            b_ = s2_pix0 + i
            fr = (np.arange(16) if HB == 1 else 16 * (b_ & 1) + np.arange(16))[LROW]
            keep = fr < F                                  # padding rows are never stored
            for j in range(SNJ):
                for r in range(4):
                    col = s2_col0 + 16 * j + 4 * LQ + r
                    out[blk_pix(b_, HB), fr[keep], col[keep]] = Y[i, j][keep, r]
    return out          # the out-projection WITHOUT bias / residual


def reference(x_tile, wqkv, bias, colsum, pe, wo, scale_log2e, eps):
    """The same block in float64 with the same rounding points: [PIX][F][320] -> out-projection without bias / residual"""
    PIX, F = x_tile.shape[0], x_tile.shape[1]
    x = x_tile.astype(np.float64)
    mean = x.mean(axis=2, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=2, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    acc = x @ wqkv.astype(np.float64).T
    qkv = bf16(rstd * acc + (-mean * rstd) * colsum[None, None, :] + (bias[None, None, :] + pe[None, :, :])).astype(np.float64)
    out = np.zeros_like(x)
    o_all = np.zeros((PIX, F, C))
    for h in range(HEADS):
        q, k, v = (qkv[:, :, o + h * D: o + (h + 1) * D] for o in (0, C, 2 * C))
        s = np.einsum("pqd,pkd->pqk", q, k)
        m = s.max(axis=2, keepdims=True)
        p_ = np.exp2((s - m) * scale_log2e)
        l = p_.sum(axis=2, keepdims=True)
        o = np.einsum("pqk,pkd->pqd", bf16(p_).astype(np.float64), v) / l
        o_all[:, :, h * D:(h + 1) * D] = bf16(o)
    out = o_all @ wo.astype(np.float64).T
    return out


def self_check(seed=0, F=16):
    rng = np.random.default_rng(seed)
    PIX = geo(F)[1]
    x = bf16(rng.standard_normal((PIX, F, C)) * 1.5 + 0.3 * rng.standard_normal((PIX, F, 1)))
    wqkv = bf16(rng.standard_normal((3 * C, C)) * C ** -0.5)
    wo = bf16(rng.standard_normal((C, C)) * C ** -0.5)
    bias = (0.2 * rng.standard_normal(3 * C)).astype(np.float32)
    colsum = wqkv.astype(np.float64).sum(axis=1).astype(np.float32)
    pe = (0.5 * rng.standard_normal((F, 3 * C))).astype(np.float32)
    scale_log2e, eps = np.float32(D ** -0.5 * 1.4426950408889634), 1e-5
    got = run_tile(x, pack(wqkv, bias, colsum, pe, wo, F), scale_log2e, eps)
    ref = reference(x, wqkv, bias, colsum, pe, wo, scale_log2e, eps)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    return float(err)


if __name__ == "__main__":
    for F_ in (16, 24):
        print(f"F = {F_}: max |emulated - reference| / max |reference| =", self_check(F=F_))
