"""Deterministic synthetic inputs of KITTI shape: the same arrays on every
machine, no dataset needed.  Counter-based generator (splitmix64 of seed+index) so values do not
depend on numpy's RNG version.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """n pseudo-random uint64, element i = splitmix64 finaliser of (seed * 2^32 + offset + i)."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(offset) + (np.uint64(seed) << np.uint64(32)))
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform_u8(seed, shape, hi=255):
    """uint8 uniform in [0, hi]."""
    n = int(np.prod(shape))
    return ((splitmix64(seed, n) >> np.uint64(33)) % np.uint64(hi + 1)).astype(np.uint8).reshape(shape)


def uniform_f64(seed, shape):
    """float64 uniform in [0, 1) with 53 random bits."""
    n = int(np.prod(shape))
    return ((splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)).reshape(shape)


def image_pair(W, H, D, seed=1):
    """I1 = 3x3 box-smoothed u8 noise, I2[y][x] = I1[y][min(W-1, x + s(x,y))], smooth shift s in [0, D/2)."""
    noise = uniform_u8(seed, (H, W)).astype(np.int32)
    pad = np.pad(noise, 1, mode="edge")
    acc = np.zeros((H, W), np.int32)
    for dy in range(3):
        for dx in range(3):
            acc += pad[dy:dy + H, dx:dx + W]
    I1 = (acc // 9).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    s = ((np.sin(xx / 97.0) * np.cos(yy / 61.0) * 0.5 + 0.5) * (D / 2 - 1)).astype(np.int64)
    xs = np.minimum(W - 1, xx + s)
    I2 = I1[yy, xs]
    return np.ascontiguousarray(I1), np.ascontiguousarray(I2)


def epi_maps(W, H, kind="axis", seed=7):
    """pixelPosD0 (2,H,W), normlizeDirection (2,H,W), offsetFromPosD0 (H,W).

    'axis'   : the survey's timing inputs -- Pd0=(x+1,y+1), direction (-1,0), offset 200.
    'general': fractional start positions, a rotating unit-direction field and a varying offset,
               so that both coordinates, round-half cases and clamping are exercised.
    'radial' : a forward-moving camera's field -- directions away from an epipole inside the image, offset = distance to it.
    """
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    pd0 = np.stack([xx + 1.0, yy + 1.0])
    if kind == "axis":
        nd = np.stack([-np.ones((H, W)), np.zeros((H, W))])
        off = np.full((H, W), 200.0)
    elif kind == "general":
        jit = uniform_f64(seed, (2, H, W)) - 0.5
        jit[0, ::3, ::5] = 0.5                      # exact .5 cases for round-half-away-from-zero
        jit[1, ::4, ::7] = -0.5
        pd0 = pd0 + jit
        ang = 0.9 * np.sin(xx / 53.0) + 1.3 * np.cos(yy / 37.0) + 2.0 * uniform_f64(seed + 1, (H, W))
        nd = np.stack([np.cos(ang), np.sin(ang)])
        off = 40.0 + 400.0 * uniform_f64(seed + 2, (H, W))
    elif kind == "radial":
        # what epipolar_geometry.m:99-115 produces for a forward-moving camera: start positions = pixel + a small smooth
        # rotation flow, unit directions pointing away from the epipole, offset = distance to it (both coordinates walk, the
        # field is smooth: neighbouring pixels sample neighbouring positions)
        ex, ey = 0.47 * W + 3.3, 0.55 * H - 1.7
        pd0 = pd0 + np.stack([0.8 * np.sin(yy / 91.0) + 0.002 * (xx - W / 2), 0.6 * np.cos(xx / 123.0) - 0.001 * (yy - H / 2)])
        dxy = np.stack([pd0[0] - ex, pd0[1] - ey])
        off = np.sqrt((dxy ** 2).sum(0))
        nd = dxy / np.maximum(off, 1e-9)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(pd0), np.ascontiguousarray(nd), np.ascontiguousarray(off)


def cost_volume(W, H, D, seed=3, cmax=24):
    """u8 uniform in [0, cmax] of shape (H, W, D): aggregation-only input."""
    return uniform_u8(seed, (H, W, D), hi=cmax)


def hint_map(mvW, mvH, kind="zero", seed=21, amp=3.0):
    """preMv (2, mvH, mvW) for the pyramidal / neighbour-guided variants.

    'zero'   : coarsest pyramid level (pyramidal_sgm.m:34).
    'even'   : 2*integer values in 2x2 blocks, what 2*imresize(mv,2,'nearest') hands the next
               level when sub-pixel refinement is off (pyramidal_sgm.m:72).
    'int'    : integer hints that change from pixel to pixel (negative deltas: the truncating
               conversion rounds those up, calc_pyd_cost_sgm.cpp:46-47).
    'general': fractional hints (exercise the +0.5 / truncation rules and the (-1,0) -> 0 case).
    """
    if kind == "zero":
        return np.zeros((2, mvH, mvW))
    if kind == "even":
        h2, w2 = (mvH + 1) // 2, (mvW + 1) // 2
        coarse = np.floor(uniform_f64(seed, (2, h2, w2)) * (2 * amp + 1)) - amp
        return np.ascontiguousarray(2.0 * np.repeat(np.repeat(coarse, 2, axis=1), 2, axis=2)[:, :mvH, :mvW])
    if kind == "int":                                                  # whole numbers of either sign, a new one per pixel
        return np.ascontiguousarray(np.floor(uniform_f64(seed, (2, mvH, mvW)) * (2 * amp + 1)) - amp)
    if kind == "general":
        mv = (uniform_f64(seed, (2, mvH, mvW)) - 0.5) * 2 * amp
        mv[:, ::3, ::4] = np.round(mv[:, ::3, ::4] * 2) / 2          # exact halves
        mv[0, 1::5, 1::6] = -0.75                                     # lands in (-1, 0) near the left border
        return np.ascontiguousarray(mv)
    raise ValueError(kind)


def vz_index_map(W, H, D, seed=11, invalid=0.08, speckles=0.04):
    """A vz-index map like MATLAB sgm's D1 (test.m:36): smooth sub-pixel indices in [0, D-1], a share of
    isolated outliers (speckles), NaN holes (single pixels, runs, whole rows/columns at the borders)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    base = (np.sin(xx / 23.0) * np.cos(yy / 17.0) * 0.5 + 0.5) * (D - 1) * 0.6
    base = np.round(base * 4) / 4                                   # quarter steps: exact ties in |a-b| < maxDiff
    r = uniform_f64(seed, (H, W))
    out = np.where(r < speckles, np.floor(uniform_f64(seed + 1, (H, W)) * (D - 1)), base)
    hole = uniform_f64(seed + 2, (H, W)) < invalid
    hole[:, : max(1, W // 16)] |= uniform_f64(seed + 3, (H, max(1, W // 16))) < 0.6
    if H > 6:
        hole[H // 3, W // 4: W // 2] = True                         # a long run inside a row
        hole[0, :] = True                                           # a whole row: only the column pass reaches it
    out[hole] = np.nan
    return np.ascontiguousarray(out)


def epi_geometry(W, H, kind="forward"):
    """A plausible sparse geometry for the epipolar driver: (F, H, epipole, direction).  'forward': camera
    moving forward with a small rotation -- epipole near the image centre, expansion; 'contract': the
    reverse.  H = K R K^-1 for a rotation of about half a degree per axis, F consistent with the epipole
    (F = [e']_x H, so that every rotated point's epipolar line passes through e')."""
    f = 0.58 * W
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    ax, ay, az = 0.004, -0.007, 0.003
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    Hm = K @ (Rz @ Ry @ Rx) @ np.linalg.inv(K)
    e = np.array([W * 0.47 + 3.3, H * 0.55 - 1.7, 1.0])
    ex = np.array([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
    F = ex @ Hm
    F = F / np.abs(F).max()
    return F, Hm, (float(e[0]), float(e[1])), (1 if kind == "contract" else 0)
