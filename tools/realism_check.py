#!/usr/bin/env python3
"""Realism check of the UNPINNED layers: the CPU oracle's pyramidal_sgm -- pyramidal_sgm.m restated around the
calc_pyd_cost_sgm oracle, with impyramid / rgb2gray / imresize restated from their published behaviour -- on the one KITTI pair
the reference ships (proj/example/000000_10.png, 000000_11.png) against its ground-truth flow (000000_10_gtFlow.png), scored
the way test_psgm.m:49-50 scores it (flow_error with thr = [3; 0.05]: outlier = end-point error > 3 px AND > 5 % of the
ground-truth magnitude, over the valid pixels).

Build container only: the images are the reference's (KITTI licence) and are neither committed nor shipped to the GPU box;
this is a plausibility bound on the restated toolbox functions, NOT a parity pin.
usage: tools/realism_check.py [numPyd ...]        (default: 3 as test_psgm.m:33, and 5 = pyramidal_sgm.m:12's default)"""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

EX = "/root/reference/proj/example"


def read_png(path):
    """Minimal PNG reader (8 / 16 bit, gray / RGB, non-interlaced): PIL truncates 16-bit RGB, which is what KITTI flow maps are."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, [], None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif typ == b"IDAT":
            idat.append(data)
        pos += 12 + n
    W, H, depth, ctype, _, _, interlace = hdr
    assert interlace == 0 and depth in (8, 16) and ctype in (0, 2), hdr
    ch, bpp = (1 if ctype == 0 else 3), (1 if ctype == 0 else 3) * depth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(H, 1 + W * bpp)
    out = np.zeros((H, W * bpp), np.uint8)
    prev = np.zeros(W * bpp, np.int32)
    for y in range(H):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:                                   # Sub / Average / Paeth depend on the left neighbour: bpp interleaved chains
            cur = np.zeros_like(line)
            for i in range(0, W * bpp, bpp):
                a = cur[i - bpp:i] if i else np.zeros(bpp, np.int32)
                bb = prev[i:i + bpp]
                c = prev[i - bpp:i] if i else np.zeros(bpp, np.int32)
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + bb) >> 1
                else:
                    pa, pb, pc = np.abs(bb - c), np.abs(a - c), np.abs(a + bb - 2 * c)
                    p = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, bb, c))
                cur[i:i + bpp] = (line[i:i + bpp] + p) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        out = out.reshape(H, W, ch, 2)
        img = (out[..., 0].astype(np.uint16) << 8) | out[..., 1]
    else:
        img = out.reshape(H, W, ch)
    return img[..., 0] if ch == 1 else img


def flow_error(gt_u, gt_v, valid, u, v, tau=(3.0, 0.05)):
    """KITTI devkit flow_error: share of valid pixels with E > tau[0] and E / |gt| > tau[1]; mean E over the valid pixels."""
    E = np.sqrt((u - gt_u) ** 2 + (v - gt_v) ** 2)
    mag = np.sqrt(gt_u ** 2 + gt_v ** 2)
    bad = valid & (E > tau[0]) & (E / np.maximum(mag, 1e-30) > tau[1])
    return bad.sum() / valid.sum(), E[valid].mean()


def main():
    from oracle import pyoracle
    I0 = read_png(f"{EX}/000000_10.png")
    I1 = read_png(f"{EX}/000000_11.png")
    gt = read_png(f"{EX}/000000_10_gtFlow.png").astype(np.float64)
    gu, gv, valid = (gt[..., 0] - 2 ** 15) / 64.0, (gt[..., 1] - 2 ** 15) / 64.0, gt[..., 2] > 0        # flow_read_kitti
    H, W = I0.shape[:2]
    print(f"pair {W}x{H}, {I0.shape[2] if I0.ndim == 3 else 1} channels; ground truth valid on {valid.mean() * 100:.1f} % of the pixels, "
          f"|flow| up to {np.sqrt(gu ** 2 + gv ** 2)[valid].max():.1f} px")
    P0 = np.ascontiguousarray(np.moveaxis(I0, 2, 0)) if I0.ndim == 3 else I0                            # (3, H, W) planes, x fastest
    P1 = np.ascontiguousarray(np.moveaxis(I1, 2, 0)) if I1.ndim == 3 else I1
    for n in [int(a) for a in sys.argv[1:]] or [3, 5]:
        t0 = time.time()
        mv, minC, lv = pyoracle.pyramidal_sgm(P0, P1, n)                                                 # pyramidal_sgm.m defaults: P1=6, P2=32, 11x11, 8 paths, 2 passes
        out, aepe = flow_error(gu, gv, valid, mv[0], mv[1])
        zo, za = flow_error(gu, gv, valid, np.zeros_like(gu), np.zeros_like(gu))
        reach = 5 * (2 ** n - 1)                                                                         # +-5 per level, doubled from level to level
        near = valid & (np.maximum(np.abs(gu), np.abs(gv)) <= reach)
        no, na = flow_error(gu, gv, near, mv[0], mv[1])
        print(f"oracle pyramidal_sgm(I0, I1, {n}): outlier {out:.4f}, AEPE {aepe:.3f} px   (zero flow: {zo:.4f} / {za:.3f}); "
              f"pixels whose true flow is within the pyramid's reach of +-{reach} px ({near.sum() / valid.sum() * 100:.0f} % of the valid ones): "
              f"outlier {no:.4f}, AEPE {na:.3f}   [{time.time() - t0:.0f} s on one core]", flush=True)


if __name__ == "__main__":
    main()
