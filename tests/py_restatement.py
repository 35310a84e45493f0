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


# =====================================================================================
# calc_pyd_cost_sgm.cpp -- second restatement, raster order like the reference
# =====================================================================================
def _trunc(v):                                          # C (int) conversion of a double
    return int(v)


def pyd_cost(cen1, cen2, mv, rAgg, rX, rY):             # :374-437
    H, W = cen1.shape
    Sy = 2 * rY + 1
    D = (2 * rX + 1) * Sy
    C = np.zeros((H, W, D), np.uint8)
    win = (2 * rAgg + 1) ** 2
    for y in range(H):
        for x in range(W):
            mvx, mvy = mv[0, y, x], mv[1, y, x]
            d = 0
            for offx in range(-rX, rX + 1):
                for offy in range(-rY, rY + 1):
                    s = 0
                    for ay in range(-rAgg, rAgg + 1):
                        for ax in range(-rAgg, rAgg + 1):
                            y1, x1 = y + ay, x + ax
                            if y1 < 0 or y1 > H - 1 or x1 < 0 or x1 > W - 1:
                                s += 5
                                continue
                            y2 = _trunc(1.0 * (offy + y1) + mvy + 0.5)
                            x2 = _trunc(1.0 * (offx + x1) + mvx + 0.5)
                            if y2 < 0 or y2 > H - 1 or x2 < 0 or x2 > W - 1:
                                s += 5
                                continue
                            s += bin(int(cen1[y1, x1]) ^ int(cen2[y2, x2])).count("1")
                    C[y, x, d] = _trunc((1.0 * s / win) + 0.5) & 0xFF
                    d += 1
    return C


def pyd_step(Lpre, C, dx, dy, Sx, Sy, P1, P2):          # :34-89
    D = Sx * Sy
    out = [0] * (D + 1)
    lpm = Lpre[D]
    mn = 255
    for sx in range(Sx):
        for sy in range(Sy):
            ypre = _trunc(sy + dy + 0.5)
            xpre = _trunc(sx + dx + 0.5)
            min1 = min2 = _u8(lpm + P2)
            if 0 <= xpre < Sx and 0 <= ypre < Sy:
                min1 = Lpre[xpre * Sy + ypre]
            for k in range(-2, 3):
                for m in range(-2, 3):
                    if m == 0 and k == 0:
                        continue
                    ty, tx = ypre + k, xpre + m
                    if 0 <= tx < Sx and 0 <= ty < Sy:
                        min2 = min(min2, _u8(Lpre[tx * Sy + ty] + P1))
            best = min(_u8(lpm + P2), min1, min2)
            d = sx * Sy + sy
            out[d] = _u8(int(C[d]) + best - lpm)
            mn = min(mn, out[d])
    out[D] = mn
    return out


def pyd_sgm2d(I1, C, mv, Sx, Sy, P1, P2, diag, totalPass, adaptive):      # :114-296
    H, W, D = C.shape
    Sp = np.zeros((H, W, D), np.int64)

    def p2(cur, pre):
        if not adaptive:
            return P2
        return P2 // 8 if abs(int(I1[cur]) - int(I1[pre])) > 50 else P2

    ystart, yend, ystep, xstart, xend, xstep = 0, H, 1, 0, W, 1
    for ps in range(totalPass):
        if ps == 1:
            ystart, yend, ystep, xstart, xend, xstep = H - 1, -1, -1, W - 1, -1, -1
        L1pre = None
        L3pre, L2pre, L4pre = {}, {}, {}
        y = ystart
        while y != yend:
            L3cur, L2cur, L4cur = {}, {}, {}
            x = xstart
            while x != xend:
                c = C[y, x]
                st = [int(v) for v in c] + [0]

                def step(Lp, py, px):
                    return pyd_step(Lp, c, mv[0, y, x] - mv[0, py, px], mv[1, y, x] - mv[1, py, px], Sx, Sy, P1,
                                    p2((y, x), (py, px)))
                L1 = st if x == xstart else step(L1pre, y, x - xstep)
                L3 = st if y == ystart else step(L3pre[x], y - ystep, x)
                tot = np.array(L1[:D]) + np.array(L3[:D])
                if diag:
                    L2 = st if (x == xstart or y == ystart) else step(L2pre[x - xstep], y - ystep, x - xstep)
                    L4 = st if (y == ystart or x == xend - xstep) else step(L4pre[x + xstep], y - ystep, x + xstep)
                    tot = tot + np.array(L2[:D]) + np.array(L4[:D])
                    L2cur[x], L4cur[x] = L2, L4
                Sp[y, x] += tot
                L1pre = L1
                L3cur[x] = L3
                x += xstep
            L3pre, L2pre, L4pre = L3cur, L2cur, L4cur
            y += ystep
    return Sp.astype(np.uint32)


def pyd_wta(Sp, Sx, Sy, subpixel):                      # :298-364
    H, W, D = Sp.shape
    bestD = np.zeros((H, W), np.uint32)
    minC = np.zeros((H, W), np.uint32)
    mvSub = np.zeros((2, H, W))

    def par(cl, c0, cr):
        return (cr - cl) / (c0 - cl) / 2.0 if cr < cl else (cr - cl) / (c0 - cr) / 2.0
    for y in range(H):
        for x in range(W):
            s = [float(v) for v in Sp[y, x]]
            idx = 0
            for d in range(1, D):
                if s[d] < s[idx]:
                    idx = d
            bestD[y, x], minC[y, x] = idx, int(s[idx])
            if not subpixel:
                continue
            dx, dy = idx // Sy, idx % Sy
            if 0 < dy < Sy - 1:
                mvSub[1, y, x] = par(s[idx - 1], s[idx], s[idx + 1])
            if 0 < dx < Sx - 1:
                mvSub[0, y, x] = par(s[idx - Sy], s[idx], s[idx + Sy])
    return bestD, minC, mvSub


# =====================================================================================
# calc_pyd_cost_sgm_ng.cpp -- second restatement
# =====================================================================================
def ng_cost(cen1, cen2, mv, rAgg, r):                   # :370-446
    H, W = cen1.shape
    mvH, mvW = mv.shape[1:]
    cph = (2 * r + 1) ** 2
    D = 9 * cph
    out = np.zeros((H, W, D, 3), np.int64)              # mvx, mvy, cost
    win = (2 * rAgg + 1) ** 2
    for y in range(H):
        for x in range(W):
            d = 0
            for dy in (-8, 0, 8):
                for dx in (-8, 0, 8):
                    yn, xn = min(max(y + dy, 0), mvH - 1), min(max(x + dx, 0), mvW - 1)
                    mvx, mvy = mv[0, yn, xn], mv[1, yn, xn]
                    for offx in range(-r, r + 1):
                        for offy in range(-r, r + 1):
                            s = 0
                            for ay in range(-rAgg, rAgg + 1):
                                for ax in range(-rAgg, rAgg + 1):
                                    y1, x1 = y + ay, x + ax
                                    if y1 < 0 or y1 > H - 1 or x1 < 0 or x1 > W - 1:
                                        s += 5
                                        continue
                                    y2, x2 = _trunc((offy + y1) + mvy), _trunc((offx + x1) + mvx)
                                    if y2 < 0 or y2 > H - 1 or x2 < 0 or x2 > W - 1:
                                        s += 5
                                        continue
                                    s += bin(int(cen1[y1, x1]) ^ int(cen2[y2, x2])).count("1")
                            out[y, x, d] = (_trunc(mvx + offx), _trunc(mvy + offy), _trunc((1.0 * s / win) + 0.5))
                            d += 1
    return out


def ng_step(Lpre, lpm_entry, C, P1, P2):                # :39-78; entries are (mvx, mvy, cost)
    D = len(C)
    lpm = _u8(lpm_entry)
    out = []
    mn = 255
    for d in range(D):
        mvx, mvy, cc = C[d]
        min1 = min2 = _u8(lpm + P2)
        for d2 in range(D):
            px, py, pc = Lpre[d2]
            if mvx == px and mvy == py:
                min1 = _u8(pc)
            elif abs(mvx - px) <= 2 and abs(mvy - py) <= 2:
                min2 = min(min2, _u8(pc + P1))
        best = min(_u8(lpm + P2), min1, min2)
        cost = (cc + best) - lpm
        out.append((mvx, mvy, cost))
        mn = min(mn, _u8(cost))
    return out, mn


def ng_sgm2d(Cc, P1, P2):                               # :101-299 (2 passes x 2 paths)
    H, W, D, _ = Cc.shape
    Sp = np.zeros((H, W, D), np.int64)
    for ps in range(2):
        if ps == 0:
            ys, ye, yst, xs, xe, xst = 0, H, 1, 0, W, 1
        else:
            ys, ye, yst, xs, xe, xst = H - 1, -1, -1, W - 1, -1, -1
        L1pre = None
        L3pre = {}
        y = ys
        while y != ye:
            L3cur = {}
            x = xs
            while x != xe:
                c = [tuple(int(v) for v in e) for e in Cc[y, x]]
                L1 = (c, 0) if x == xs else ng_step(L1pre[0], L1pre[1], c, P1, P2)
                L3 = (c, 0) if y == ys else ng_step(L3pre[x][0], L3pre[x][1], c, P1, P2)
                Sp[y, x] += np.array([e[2] for e in L1[0]]) + np.array([e[2] for e in L3[0]])
                L1pre = L1
                L3cur[x] = L3
                x += xst
            L3pre = L3cur
            y += yst
    S = (Sp & 0xFFFFFFFF).astype(np.uint32)
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W))
    for y in range(H):
        for x in range(W):
            idx = 0
            for d in range(1, D):
                if S[y, x, d] < S[y, x, idx]:
                    idx = d
            minC[y, x] = S[y, x, idx]
            flow[0, y, x], flow[1, y, x] = Cc[y, x, idx, 0], Cc[y, x, idx, 1]
    return S, minC, flow


# =====================================================================================
# calc_cost_sgm_ng.cpp -- second restatement (literal double buffers, raster order)
# =====================================================================================
def otf(I1, I2, P1, P2, rnd):
    H, W = I1.shape
    c1, c2 = census(I1), census(I2)
    N, M, D = 2, 1, 108
    E = D + N
    zero = lambda n: [[0, 0, 0] for _ in range(n)]       # [mvx, mvy, cost]
    L1 = [zero(E), zero(E)]
    L2 = [[zero(E) for _ in range(W)] for _ in range(2)]
    L3 = [[zero(E) for _ in range(W)] for _ in range(2)]
    L4 = [[zero(E) for _ in range(W)] for _ in range(2)]
    Sp = np.zeros((H, W, D), np.int64)
    Cvol = np.zeros((H, W, D, 3), np.int64)
    ri = [0]

    def step(L, Lpre, C, p2):                            # :46-98, writes into list L in place
        lpm = _u8(Lpre[D][2])
        for i in range(N):
            L[D + i][2] = 255
        for d in range(D):
            mvx, mvy, cc = C[d]
            min1 = min2 = _u8(lpm + p2)
            for d2 in range(D):
                px, py, pc = Lpre[d2]
                if mvx == px and mvy == py:
                    min1 = _u8(pc)
                elif abs(mvx - px) <= 2 and abs(mvy - py) <= 2:
                    min2 = min(min2, _u8(pc + P1))
            best = min(_u8(lpm + p2), min1, min2)
            L[d] = [mvx, mvy, (cc + best) - lpm]
            j = 0
            while j < N and not (L[d][2] < L[D + j][2]):
                j += 1
            if j < N:
                for i in range(N - 1, j, -1):
                    L[D + i] = list(L[D + i - 1])
                L[D + j] = list(L[d])

    def start(L, C):
        for d in range(D):
            L[d] = list(C[d])
        L[D][2] = 0

    l1pre, l1cur, rpre, rcur = 0, 1, 0, 1
    for y in range(H):
        for x in range(W):
            pL1c, pL2c, pL3c, pL4c = L1[l1cur], L2[rcur][x], L3[rcur][x], L4[rcur][x]
            C = []
            for hint in (pL1c, pL2c, pL3c, pL4c):        # :122-186
                for i in range(N + M):
                    if i < N:
                        mvx, mvy = hint[D + i][0], hint[D + i][1]
                    else:
                        mvx = int(rnd[ri[0]]) % 256 - 128
                        mvy = int(rnd[ri[0] + 1]) % 128 - 64
                        ri[0] += 2
                    for offy in (-1, 0, 1):
                        for offx in (-1, 0, 1):
                            s = 0
                            for ay in range(-2, 3):
                                for ax in range(-2, 3):
                                    y1 = min(max(y + ay, 0), H - 1)
                                    x1 = min(max(x + ax, 0), W - 1)
                                    y2 = min(max((offy + y1) + mvy, 0), H - 1)
                                    x2 = min(max((offx + x1) + mvx, 0), W - 1)
                                    s += bin(int(c1[y1, x1]) ^ int(c2[y2, x2])).count("1")
                            C.append([mvx + offx, mvy + offy, int(1.0 * s / 25 + 0.5)])
            Cvol[y, x] = C
            if x == 0:
                start(pL1c, C)
                start(pL2c, C)
            if y == 0:
                start(pL3c, C)
                start(pL2c, C)
                start(pL4c, C)
            if x == W - 1:
                start(pL4c, C)
            pc = int(I1[y, x])

            def ap2(pp):
                return P2 // 8 if abs(pc - int(pp)) > 50 else P2
            if x != 0:
                step(pL1c, L1[l1pre], C, ap2(I1[y, x - 1]))
            if y != 0:
                step(pL3c, L3[rpre][x], C, ap2(I1[y - 1, x]))
            if x != 0 and y != 0:
                step(pL2c, L2[rpre][x - 1], C, ap2(I1[y - 1, x - 1]))
            if x != W - 1 and y != 0:
                step(pL4c, L4[rpre][x + 1], C, ap2(I1[y - 1, x + 1]))
            for d in range(D):
                Sp[y, x, d] += pL1c[d][2] + pL3c[d][2]
                Sp[y, x, d] += pL2c[d][2] + pL4c[d][2]
            l1pre, l1cur = l1cur, l1pre
        rpre, rcur = rcur, rpre
    S = (Sp & 0xFFFFFFFF).astype(np.uint32)
    minC = np.zeros((H, W), np.uint32)
    flow = np.zeros((2, H, W))
    for y in range(H):
        for x in range(W):
            idx = 0
            for d in range(1, D):
                if S[y, x, d] < S[y, x, idx]:
                    idx = d
            minC[y, x] = S[y, x, idx]
            flow[0, y, x], flow[1, y, x] = Cvol[y, x, idx, 0], Cvol[y, x, idx, 1]
    return minC, flow


def fb_check(D1, pd0, nd, off, vMax, n, thr=2):         # calc_cost_sgm.cpp:429-536 (USE_VZIND), raster order
    H, W = D1.shape
    INVALID = 512 << 8
    D2 = np.full((H, W), INVALID, np.int64)

    def disp(y, x):
        d = float(D1[y, x]) / 256
        r = d / n * vMax
        return off[y, x] * (r / (1 - r))
    for y in range(H):
        for x in range(W):
            d = disp(y, x)
            p2x = int((pd0[0, y, x] - 1) + d * nd[0, y, x])
            p2y = int((pd0[1, y, x] - 1) + d * nd[1, y, x])
            for dy in (0, 1):
                for dx in (0, 1):
                    tx, ty = dx + p2x, dy + p2y
                    if 0 <= tx < W and 0 <= ty < H:
                        if D2[ty, tx] == INVALID or D2[ty, tx] < D1[y, x]:
                            D2[ty, tx] = D1[y, x]
    conf = np.ones((H, W), np.uint8)
    for y in range(H):
        for x in range(W):
            d = disp(y, x)
            p2x = int(_c_round((pd0[0, y, x] - 1) + d * nd[0, y, x]))
            p2y = int(_c_round((pd0[1, y, x] - 1) + d * nd[1, y, x]))
            if p2x < 0 or p2x > W - 1 or p2y < 0 or p2y > H - 1 or D2[p2y, p2x] == INVALID:
                conf[y, x] = 0
            elif abs(int(D1[y, x]) - int(D2[p2y, p2x])) > thr:
                conf[y, x] = 0
    return conf, D2.astype(np.uint32)


# =====================================================================================
# pyramidal_sgm.m -- second restatement of the driver.  The toolbox functions it calls are restated
# in the form of imresize's own algorithm (contribution tables in floating point, one dimension at
# a time), not in the integer form the oracle uses.
# =====================================================================================
def _pyr_kernel(x):                                     # impyramid's piecewise-constant 'reduce' kernel
    for brk, val in ((3.5, 0.0), (2.5, 0.0625), (1.5, 0.25), (0.5, 0.375), (-0.5, 0.25), (-1.5, 0.0625)):
        if x >= brk:
            return val
    return 0.0


def _contributions(in_len, out_len, scale, kernel, kernel_width):
    """imresize's contributions(): per output sample, the input indices (0-based, mirrored) and weights."""
    table = []
    P = int(np.ceil(kernel_width)) + 2
    for x in range(1, out_len + 1):
        u = x / scale + 0.5 * (1 - 1 / scale)
        left = int(np.floor(u - kernel_width / 2))
        idx = [left + k for k in range(P)]
        wts = [kernel(u - i) for i in idx]
        tot = sum(wts)
        wts = [w / tot for w in wts]
        aux = list(range(1, in_len + 1)) + list(range(in_len, 0, -1))
        idx = [aux[(i - 1) % len(aux)] - 1 for i in idx]
        table.append([(i, w) for i, w in zip(idx, wts) if w != 0.0])
    return table


def _resize_along(img, axis, table):
    out_shape = list(img.shape)
    out_shape[axis] = len(table)
    out = np.zeros(out_shape, np.uint8)
    src = np.moveaxis(img, axis, 0).astype(np.float64)
    dst = np.moveaxis(out, axis, 0)
    for o, taps in enumerate(table):
        acc = np.zeros(src.shape[1:])
        for i, w in taps:
            acc = acc + w * src[i]
        dst[o] = np.clip(np.floor(acc + 0.5), 0, 255).astype(np.uint8)      # uint8(): round, saturate
    return out


def impyramid_reduce(img):                              # impyramid(A, 'reduce'), one channel, (H, W)
    H, W = img.shape
    t = _resize_along(img, 0, _contributions(H, (H + 1) // 2, 0.5, _pyr_kernel, 5))      # dimension 1 first
    return _resize_along(t, 1, _contributions(W, (W + 1) // 2, 0.5, _pyr_kernel, 5))


def rgb2gray(rgb):                                      # (3, H, W) uint8
    T = np.linalg.inv(np.array([[1.0, 0.956, 0.621], [1.0, -0.272, -0.647], [1.0, -1.106, 1.703]]))
    c = T[0]
    v = c[0] * rgb[0].astype(np.float64) + c[1] * rgb[1].astype(np.float64) + c[2] * rgb[2].astype(np.float64)
    return np.floor(v + 0.5).astype(np.uint8)


def resize2_nearest(a):                                 # imresize(a, 2, 'nearest') on (H, W)
    box = lambda x: 1.0 if -0.5 <= x < 0.5 else 0.0
    H, W = a.shape
    ty = [t[0][0] for t in _contributions(H, 2 * H, 2.0, box, 1.0)]
    tx = [t[0][0] for t in _contributions(W, 2 * W, 2.0, box, 1.0)]
    return a[np.ix_(ty, tx)]


def pyramidal_sgm(I0, I1, numPyd, P1=6, P2=32, agg=2, ver=5, hor=5, diag=1, totalPass=2, adaptive=0):   # :1-77
    chans = (lambda a: [a] if a.ndim == 2 else list(a))
    p0, p1 = [chans(I0)], [chans(I1)]
    for _ in range(1, numPyd):                                                       # :28-31
        p0.append([impyramid_reduce(c) for c in p0[-1]])
        p1.append([impyramid_reduce(c) for c in p1[-1]])
    gray = lambda cs: cs[0] if len(cs) == 1 else rgb2gray(np.stack(cs))
    Hc, Wc = p0[-1][0].shape
    mvPre = np.zeros((2, Hc, Wc))                                                    # :34
    Sx, Sy = 2 * hor + 1, 2 * ver + 1
    levels = [None] * numPyd
    minC = None
    for l in range(numPyd - 1, -1, -1):                                              # :37
        g0, g1 = gray(p0[l]), gray(p1[l])
        h, w = g0.shape
        C = pyd_cost(census(g0), census(g1), mvPre, agg, hor, ver)                   # :50 (the MEX)
        Sp = pyd_sgm2d(g0, C, mvPre, Sx, Sy, P1, P2, diag, totalPass, adaptive)
        bestD, minC, mvSub = pyd_wta(Sp, Sx, Sy, l == 0)
        idx = bestD.astype(np.int64)
        mvx, mvy = idx // Sy - hor, idx % Sy - ver                                   # :57-60
        cur = np.stack([mvx, mvy]).astype(np.float64) + mvPre[:, :h, :w] + mvSub     # :62-64
        levels[l] = cur
        if l > 0:
            mvPre = 2 * np.stack([resize2_nearest(cur[0]), resize2_nearest(cur[1])])  # :72
    return levels[0], minC, levels


# =====================================================================================
# post-processing (test.m:45-50) -- second restatements.  speckle_filter: NOT a flood fill like the
# original and the oracle but label propagation to a fixed point (regions = connected components of
# the symmetric "valid and |a-b| < maxDiff" relation); the rest literal loops in MATLAB's 1-based terms.
# =====================================================================================
def speckle_filter(image, maxDiff, maxSpeckleSize):     # speckle_filter.m
    H, W = image.shape
    valid = ~np.isnan(image)
    lab = np.where(valid, np.arange(H * W).reshape(H, W), -1)
    right = np.zeros((H, W), bool)
    down = np.zeros((H, W), bool)
    with np.errstate(invalid="ignore"):
        right[:, :-1] = valid[:, :-1] & valid[:, 1:] & (np.abs(image[:, :-1] - image[:, 1:]) < maxDiff)
        down[:-1, :] = valid[:-1, :] & valid[1:, :] & (np.abs(image[:-1, :] - image[1:, :]) < maxDiff)
    while True:                                         # every pixel takes the smallest label among itself and its joined neighbours
        new = lab.copy()
        r, d = right[:, :-1], down[:-1, :]
        new[:, :-1][r] = np.minimum(new[:, :-1][r], lab[:, 1:][r])
        new[:, 1:][r] = np.minimum(new[:, 1:][r], lab[:, :-1][r])
        new[:-1, :][d] = np.minimum(new[:-1, :][d], lab[1:, :][d])
        new[1:, :][d] = np.minimum(new[1:, :][d], lab[:-1, :][d])
        if np.array_equal(new, lab):
            break
        lab = new
    out = image.copy()
    labels = np.zeros((H, W), np.int32)
    roots = np.unique(lab[valid])                       # ascending = raster order of each region's first pixel
    for k, r in enumerate(roots):
        region = lab == r
        labels[region] = k + 1
        if region.sum() < maxSpeckleSize:
            out[region] = np.nan
    return out, labels


def vzInd2Disp(w, O, vMax, n):                          # vzInd2Disp.m
    vzRatio = w / n * vMax
    return O * (vzRatio / (1 - vzRatio))


def calc_disp_from_first(D1, Pd0, nd, O, vMax, n):      # calc_disp_from_first.m
    rows, cols = D1.shape
    D2 = -np.ones((rows, cols))
    for j in range(1, rows + 1):
        for i in range(1, cols + 1):
            v = D1[j - 1, i - 1]
            disp = vzInd2Disp(v, O[j - 1, i - 1], vMax, n)
            p2 = (Pd0[0, j - 1, i - 1] + disp * nd[0, j - 1, i - 1], Pd0[1, j - 1, i - 1] + disp * nd[1, j - 1, i - 1])
            if np.isnan(p2[0]) or np.isnan(p2[1]):
                continue
            sx0, sy0 = np.floor(p2[0]), np.floor(p2[1])
            for sx, sy in ((sx0, sy0), (sx0 + 1, sy0), (sx0, sy0 + 1), (sx0 + 1, sy0 + 1)):
                if 1 <= sx <= cols and 1 <= sy <= rows:
                    t = D2[int(sy) - 1, int(sx) - 1]
                    if t == 0 or t < v:
                        D2[int(sy) - 1, int(sx) - 1] = v
    return D2


def forward_backward_check(D1, D2, Pd0, nd, O, vMax, n):   # forward_backward_check.m
    rows, cols = D1.shape
    out = D1.copy()
    rnd = lambda v: np.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)       # MATLAB round: half away from zero
    for j in range(rows):
        for i in range(cols):
            v = out[j, i]
            if np.isnan(v):
                continue
            disp = vzInd2Disp(v, O[j, i], vMax, n)
            px, py = rnd(Pd0[0, j, i] + disp * nd[0, j, i]), rnd(Pd0[1, j, i] + disp * nd[1, j, i])
            if px < 1 or px > cols or py < 1 or py > rows:
                out[j, i] = np.nan
                continue
            d2 = D2[int(py) - 1, int(px) - 1]
            if d2 == -1 or abs(v - d2) > 2.0:
                out[j, i] = np.nan
    return out


def scanline_in_fill(a):                                # scanline_in_fill.m, literal
    a = a.copy()
    H, W = a.shape
    for v in range(H):
        count = 0
        for u in range(1, W + 1):
            if not np.isnan(a[v, u - 1]):
                if count >= 1:
                    u1, u2 = u - count, u - 1
                    if u1 > 1 and u2 < W:
                        a[v, u1 - 1:u2] = min(a[v, u1 - 2], a[v, u2])
                count = 0
            else:
                count += 1
        for u in range(W):
            if not np.isnan(a[v, u]):
                a[v, :u] = a[v, u]
                break
        for u in range(W - 1, -1, -1):
            if not np.isnan(a[v, u]):
                a[v, u + 1:] = a[v, u]
                break
    for u in range(W):
        for v in range(H):
            if not np.isnan(a[v, u]):
                a[:v, u] = a[v, u]
                break
        for v in range(H - 1, -1, -1):
            if not np.isnan(a[v, u]):
                a[v + 1:, u] = a[v, u]
                break
    return a


# =====================================================================================
# dense half of the epipolar driver -- second restatement, whole-array numpy in the originals' terms
# (rotation_motion.m, epipolar_geometry.m:99-115); products written out element by element (no BLAS)
# =====================================================================================
def epipolar_maps(F, Hm, epi, direction, W, H):
    xx, yy = np.meshgrid(np.arange(1, W + 1, dtype=np.float64), np.arange(1, H + 1, dtype=np.float64))
    x0, y0 = xx - 1, yy - 1                                                      # rotation_motion.m:11-13
    mul = lambda M, i: (M[i, 0] * x0 + M[i, 1] * y0) + M[i, 2]
    l = [mul(F, i) for i in range(3)]                                            # :49
    nf = np.sqrt(l[0] * l[0] + l[1] * l[1])
    nf[nf < 1e-6] = 1.0                                                          # :51
    l = [v / nf for v in l]
    q = [mul(Hm, i) for i in range(3)]                                           # :21
    p1 = [q[0] / q[2], q[1] / q[2], q[2] / q[2]]                                 # :22
    off = [p1[0] - x0, p1[1] - y0]                                               # :23
    coef = -((l[0] * p1[0] + l[1] * p1[1]) + l[2] * p1[2])                       # :27
    rflow = np.stack([off[0] + coef * l[0], off[1] + coef * l[1]])               # :28
    Pd0 = np.stack([xx, yy]) + rflow                                             # epipolar_geometry.m:106
    direct = Pd0 - np.array([epi[0], epi[1]]).reshape(2, 1, 1)                   # :107
    if direction:
        direct = -direct
    offset = np.sqrt(direct[0] * direct[0] + direct[1] * direct[1])              # :112
    return Pd0, direct / offset, offset, rflow
