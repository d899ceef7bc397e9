"""Synthetic MOM6-shaped grids for bench.py and the full-size tests.

Not a reference function: momlevel ships no large test data.  The time-invariant
grid (vertical levels, bathymetry, land mask, areacello, reference volcello) is
built on the host in numpy -- it is (nz,ny,nx) at most -- and the streamed
(time,z,y,x) fields theta/S are generated ON DEVICE by ``mlx_synth_field`` from
a counter-based hash (splitmix64 of the global cell index), so that

* any slab can be replayed bit-for-bit in numpy (``field_numpy`` below) for
  parity checks without ever moving the 100+ GB fields through the host, and
* a rank that owns a horizontal tile generates exactly its part of the global field.
"""

import numpy as np

SEED = 20251114
FIELD_THETAO, FIELD_SO, FIELD_DEPTH = 1, 2, 3
THETA_LO, THETA_SCALE = -2.0, 34.0  # theta in [-2, 32) degC
SO_LO, SO_SCALE = 30.0, 10.0        # S in [30, 40) psu
OCEAN_AREA = 3.6111092e14           # m2, util.validate_areacello's reference value

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 on uint64 arrays (wrap-around arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, field_id, gidx):
    h = splitmix64(np.uint64(seed) ^ (np.uint64(field_id) << np.uint64(60)) ^ gidx)
    return (h >> np.uint64(11)).astype(np.float64) * 2.0**-53


def field_numpy(shape, *, seed, field_id, lo, scale, mask3d=None, t0=0, global_hw=None,
                origin=(0, 0), dtype=np.float64):
    """numpy replay of mlx_synth_field (same hash, same lo + scale*u rounding)."""
    nt, nz, ny, nx = shape
    NY, NX = global_hw if global_hw is not None else (ny, nx)
    t = (np.arange(nt, dtype=np.uint64) + np.uint64(t0))[:, None, None, None]
    z = np.arange(nz, dtype=np.uint64)[None, :, None, None]
    y = (np.arange(ny, dtype=np.uint64) + np.uint64(origin[0]))[None, None, :, None]
    x = (np.arange(nx, dtype=np.uint64) + np.uint64(origin[1]))[None, None, None, :]
    gidx = ((t * np.uint64(nz) + z) * np.uint64(NY) + y) * np.uint64(NX) + x
    v = lo + scale * uniform01(seed, field_id, gidx)
    if mask3d is not None:
        v = np.where(np.isnan(mask3d)[None], np.nan, v)
    return v.astype(dtype)


def vertical_grid(nz=75):
    """dz_k = 2*1.075**k metres (75 levels -> ~6 km); returns (z_l, z_i)."""
    dz = 2.0 * 1.075 ** np.arange(nz)
    z_i = np.concatenate([[0.0], np.cumsum(dz)])
    z_l = 0.5 * (z_i[1:] + z_i[:-1])
    return z_l, z_i


def _calc_dz_host(z_i, depth):
    """calc_dz with default arguments (derived.py:295-318) -- grid set-up only."""
    depth = np.where(np.isnan(depth), 0.0, depth)[None]
    ztop, zbot = z_i[:-1][:, None, None], z_i[1:][:, None, None]
    part = np.maximum(depth - ztop, 0.0)
    return np.minimum(np.maximum(zbot, 0.0), np.minimum(part, zbot - ztop))


def make_grid(ny, nx, nz=75, seed=SEED, tile=None):
    """Time-invariant synthetic grid.

    Returns a dict of numpy arrays: z_l, z_i, deptho (ny,nx; NaN on land),
    areacello (ny,nx; NaN on land, GLOBAL sum = 3.6111092e14), volcello
    (nz,ny,nx; NaN on land and below the bottom).  ``tile=(y0, y1, x0, x1)``
    cuts the horizontal arrays to a rank's tile of the (ny,nx) global grid;
    areacello keeps its global normalisation.
    """
    z_l, z_i = vertical_grid(nz)
    yy = np.arange(ny, dtype=np.float64)[:, None]
    xx = np.arange(nx, dtype=np.float64)[None, :]
    land = np.sin(3.0 * np.pi * xx / nx) * np.cos(2.0 * np.pi * yy / ny) > 0.45
    gidx = (np.arange(ny, dtype=np.uint64)[:, None] * np.uint64(nx)
            + np.arange(nx, dtype=np.uint64)[None, :])
    # at least the first level is wet on every ocean column
    deptho = z_i[1] + (z_i[-1] - z_i[1]) * uniform01(seed, FIELD_DEPTH, gidx)
    deptho = np.where(land, np.nan, deptho)
    lat = (yy + 0.5) / ny * np.pi - 0.5 * np.pi
    area = np.cos(lat) * np.ones((1, nx))
    area = np.where(land, 0.0, area)
    area = area / area.sum() * OCEAN_AREA
    areacello = np.where(land, np.nan, area)
    dz = _calc_dz_host(z_i, deptho)
    volcello = areacello[None] * dz
    volcello = np.where((dz == 0.0) | land[None], np.nan, volcello)
    g = {"z_l": z_l, "z_i": z_i, "deptho": deptho, "areacello": areacello,
         "volcello": volcello, "global_hw": (ny, nx), "origin": (0, 0)}
    if tile is not None:
        y0, y1, x0, x1 = tile
        for k in ("deptho", "areacello"):
            g[k] = np.ascontiguousarray(g[k][y0:y1, x0:x1])
        g["volcello"] = np.ascontiguousarray(g["volcello"][:, y0:y1, x0:x1])
        g["origin"] = (y0, x0)
    return g


def tile_bounds(ny, nx, rank, world):
    """Horizontal decomposition used by the multi-GPU path: 1x1, 1x2, 2x2, 2x4 (y by x)."""
    layouts = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}
    if world not in layouts:
        py = 1
        px = world
    else:
        py, px = layouts[world]
    if ny % py or nx % px:
        raise ValueError(f"grid {ny}x{nx} does not tile {py}x{px}")
    ry, rx = divmod(rank, px)
    th, tw = ny // py, nx // px
    return (ry * th, (ry + 1) * th, rx * tw, (rx + 1) * tw)
