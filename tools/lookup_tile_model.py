"""Address-path model of the displaced CorrBlock lookup: 128-byte lines one wave's load instruction touches when the 64 lanes are
64 consecutive pixels of an image row (the layout in use: E_l[b][p / 128][dy][dx][p % 128]) against an 8 x 8 query tile per wave
(VERDICT r04 #5's proposal: E_l[b][tile][dy][dx][64]).  Level 0, one window cell per load (the kernel issues the 10 x 10 cells of a
lane's window as separate loads; every lane reads the cell at ITS integer displacement).  No GPU needed.

    python tools/lookup_tile_model.py  ->  profiles/r05_lookup_tile_model.txt (stdout)
"""
import numpy as np

H, W = 60, 128


def lines_per_load(dy, dx, tiles):
    """dy, dx: (H, W) integer displacements of the lanes; tiles: list of (ys, xs) index arrays of 64 lanes each, in lane order.
    A lane's 4-byte element sits at slot (lane index inside the wave's 64-pixel block) of the 256-byte row of ITS (dy, dx):
    lanes with equal displacement share that row; a row's 256 bytes are two 128-byte lines (lanes 0-31 / 32-63)."""
    tot = 0
    for ys, xs in tiles:
        d = dy[ys, xs].astype(np.int64) * 4096 + dx[ys, xs]
        half = np.arange(64) // 32
        tot += len(set(zip(d.tolist(), half.tolist())))
    return tot / len(tiles)


def row_tiles():
    return [(np.full(64, y), np.arange(x0, x0 + 64)) for y in range(H) for x0 in (0, 64)]


def square_tiles():
    out = []
    for y0 in range(0, H - H % 8, 8):
        for x0 in range(0, W, 8):
            ys, xs = np.meshgrid(np.arange(y0, y0 + 8), np.arange(x0, x0 + 8), indexing="ij")
            out.append((ys.reshape(-1), xs.reshape(-1)))
    return out


def main():
    rng = np.random.default_rng(0)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rt, st = row_tiles(), square_tiles()
    print("lines (128 B) per wave-load, 60 x 128 queries; 2.0 = every lane at the same displacement")
    print("%-44s %10s %10s" % ("flow", "1 x 64 row", "8 x 8 tile"))
    cases = [("constant", 0.0, 0.0, 0.0)]
    cases += [("i.i.d. noise sigma = %.2f px" % s, s, 0.0, 0.0) for s in (0.1, 0.25, 0.5, 1.0, 2.0)]
    cases += [("zoom %.0f %% (gradient %.2f px/px)" % (100 * g, g), 0.0, g, 0.0) for g in (0.01, 0.02, 0.05, 0.1, 0.2)]
    cases += [("rotation %.2f rad" % r, 0.0, 0.0, r) for r in (0.01, 0.02, 0.05, 0.1)]
    cases += [("zoom 2 %% + noise sigma = %.2f" % s, s, 0.02, 0.0) for s in (0.1, 0.5)]
    for name, sigma, zoom, rot in cases:
        cy, cx = (H - 1) / 2.0, (W - 1) / 2.0
        fy = zoom * (ys - cy) + rot * (xs - cx) + 0.3 + sigma * rng.standard_normal((H, W))
        fx = zoom * (xs - cx) - rot * (ys - cy) + 0.6 + sigma * rng.standard_normal((H, W))
        dy, dx = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
        print("%-44s %10.2f %10.2f" % (name, lines_per_load(dy, dx, rt), lines_per_load(dy, dx, st)))


if __name__ == "__main__":
    main()
