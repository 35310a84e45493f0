"""Second, independent restatement of the reference's calc_cost_sgm semantics in plain
Python/numpy (slow: tiny shapes only).  Written in the reference's own raster order -- two
passes, four path buffers per pass -- i.e. structured differently from oracle/fsgm_oracle_epi.cpp
(one independent recurrence per direction), so that agreement between the two is evidence that
both read the reference the same way.  Citations: calc_cost_sgm.cpp.  Test infrastructure only.
"""
import math
import numpy as np


def _u8(v):
    return v & 0xFF


def sgm_step(Lpre, C, D, P1, P2):                       # :33-66
    out = [0] * (D + 1)
    lpm = Lpre[D]
    mn = 255
    for d in range(D):
        min1 = Lpre[d]
        min2 = _u8(lpm + P2)
        if d > 0:
            min2 = min(min2, _u8(Lpre[d - 1] + P1))
        if d < D - 1:
            min2 = min(min2, _u8(Lpre[d + 1] + P1))
        best = min(_u8(lpm + P2), min1, min2)
        out[d] = _u8(int(C[d]) + best - lpm)
        mn = min(mn, out[d])
    out[D] = mn
    return out


def sgm_raster(C, P1, P2, diag):                        # :86-257
    H, W, D = C.shape
    Sp = np.zeros((H, W, D), np.int64)
    for ps in range(2):
        if ps == 0:
            ystart, yend, ystep, xstart, xend, xstep = 0, H, 1, 0, W, 1
        else:
            ystart, yend, ystep, xstart, xend, xstep = H - 1, -1, -1, W - 1, -1, -1
        L1pre = None
        L3pre, L2pre, L4pre = {}, {}, {}
        y = ystart
        while y != yend:
            L3cur, L2cur, L4cur = {}, {}, {}
            x = xstart
            while x != xend:
                c = C[y, x]
                startL = [int(v) for v in c] + [0]
                L1 = startL if x == xstart else sgm_step(L1pre, c, D, P1, P2)
                L3 = startL if y == ystart else sgm_step(L3pre[x], c, D, P1, P2)
                tot = np.array(L1[:D]) + np.array(L3[:D])
                if diag:
                    L2 = startL if (x == xstart or y == ystart) else sgm_step(L2pre[x - xstep], c, D, P1, P2)
                    L4 = startL if (y == ystart or x == xend - xstep) else sgm_step(L4pre[x + xstep], c, D, P1, P2)
                    tot = tot + np.array(L2[:D]) + np.array(L4[:D])
                    L2cur[x], L4cur[x] = L2, L4
                Sp[y, x] += tot
                L1pre = L1
                L3cur[x] = L3
                x += xstep
            L3pre, L2pre, L4pre = L3cur, L2cur, L4cur
            y += ystep
    return Sp.astype(np.uint32)


def census(img):                                        # common.cpp:3-27
    H, W = img.shape
    out = np.zeros((H, W), np.uint32)
    for y in range(H):
        for x in range(W):
            code = 0
            for oy in range(-2, 3):
                for ox in range(-2, 3):
                    y2 = min(max(y + oy, 0), H - 1)
                    x2 = min(max(x + ox, 0), W - 1)
                    if img[y2, x2] >= img[y, x]:
                        code += 1
                    code = (code << 1) & 0xFFFFFFFF
            out[y, x] = code
    return out


def _c_round(v):                                        # C round(): half away from zero
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def calc_cost(I1, I2, D, vMax, pd0, nd, off):           # :319-412
    H, W = I1.shape
    c1, c2 = census(I1), census(I2)
    n = float(D + 1)
    raw = np.zeros((H, W, D), np.int64)
    for y in range(H):
        for x in range(W):
            bx, by = pd0[0, y, x] - 1, pd0[1, y, x] - 1
            for d in range(D):
                vzRatio = 1.0 * d / n * vMax
                vzInd = vzRatio / (1 - vzRatio)
                ox = off[y, x] * vzInd * nd[0, y, x]
                oy = off[y, x] * vzInd * nd[1, y, x]
                x2 = min(max(int(_c_round(bx + ox)), 0), W - 1)
                y2 = min(max(int(_c_round(by + oy)), 0), H - 1)
                raw[y, x, d] = bin(int(c1[y, x]) ^ int(c2[y2, x2])).count("1")
    C = np.zeros((H, W, D), np.uint8)
    for y in range(H):
        for x in range(W):
            ys = [min(max(y + k, 0), H - 1) for k in range(-2, 3)]
            xs = [min(max(x + k, 0), W - 1) for k in range(-2, 3)]
            s = raw[np.ix_(ys, xs)].sum(axis=(0, 1))
            C[y, x] = [int(1.0 * int(v) / 25 + 0.5) for v in s]
    return C


def wta(Sp, subpixel=True):                             # :259-308
    H, W, D = Sp.shape
    flat = np.concatenate([Sp.reshape(-1).astype(np.int64), [0]])
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    for y in range(H):
        for x in range(W):
            s = Sp[y, x]
            idx = 0
            for d in range(1, D):
                if s[d] < s[idx]:
                    idx = d
            minC[y, x] = s[idx]
            if not subpixel:
                bestD[y, x] = idx
            elif 1 < idx < D:
                base = (y * W + x) * D
                c_1, c, c1 = float(flat[base + idx - 1]), float(flat[base + idx]), float(flat[base + idx + 1])
                sub = float(idx)
                sub += (c1 - c_1) / (c - c_1) / 2.0 if c1 < c_1 else (c1 - c_1) / (c - c1) / 2.0
                bestD[y, x] = int(sub * 256)
            else:
                bestD[y, x] = idx * 256
    return bestD, minC


def vz_to_disp(bestD, off, vMax, n):                    # :414-426
    out = np.zeros_like(bestD)
    H, W = bestD.shape
    for y in range(H):
        for x in range(W):
            d = float(bestD[y, x]) / 256
            r = d / n * vMax
            out[y, x] = int((off[y, x] * (r / (1 - r))) * 256)
    return out
