"""Occupancy-grid datasets and signed-distance-field generation (host side, numpy/scipy).

Restates the data tables of the reference's toolbox (no code is shared with it):
  matlab/+gpmp2/generate2Ddataset.m:21-76, matlab/+gpmp2/generate3Ddataset.m:21-88,
  matlab/+gpmp2/signedDistanceField2D.m:14-34, matlab/+gpmp2/signedDistanceField3D.m:14-34
  (scipy.ndimage.distance_transform_edt is the equivalent the reference's own Python port uses,
  gpmp2_python/gpmp2_python/utils/signedDistanceField3D.py:32-36).

This is the step *before* the hot path; it only produces inputs.

Index conventions (they matter for parity):
  * 2-D maps are `map[row = y, col = x]` (generate2Ddataset.m:10-11).
  * 3-D maps are `map[i, j, k]` with the FIRST index along x, the second along y
    (generate3Ddataset.m:10-12 "rows (x), cols (y)"); the MATLAB example transposes every slice
    before `initFieldData` (matlab/WAMFactorGraphExample.m:63-67) so the SDF matrix is
    (row = y, col = x).  `sdf3_zyx()` returns the array in the engine's [z][y][x] order.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy import ndimage


@dataclass
class Dataset2D:
    map: np.ndarray          # [rows(y), cols(x)] occupancy 0/1
    rows: int
    cols: int
    origin_x: float
    origin_y: float
    cell_size: float


@dataclass
class Dataset3D:
    map: np.ndarray          # [x, y, z] occupancy 0/1
    rows: int                # x count
    cols: int                # y count
    z: int
    origin_x: float
    origin_y: float
    origin_z: float
    cell_size: float
    corner_idx: list = field(default_factory=list)


def _add_obstacle_2d(position, size, m):
    # 1-based MATLAB indices: position-half : position+half (generate2Ddataset.m:84-92)
    hr = int(np.floor((size[0] - 1) / 2))
    hc = int(np.floor((size[1] - 1) / 2))
    r0, c0 = int(round(position[0])), int(round(position[1]))
    m[r0 - hr - 1:r0 + hr, c0 - hc - 1:c0 + hc] = 1.0
    return m


def generate2Ddataset(name: str) -> Dataset2D:
    def mk(cols, rows, ox, oy, cell):
        return Dataset2D(np.zeros((rows, cols)), rows, cols, ox, oy, cell)

    if name == "OneObstacleDataset":
        d = mk(300, 300, -1, -1, 0.01)
        _add_obstacle_2d([190, 160], [60, 80], d.map)
    elif name == "TwoObstaclesDataset":
        d = mk(300, 300, -1, -1, 0.01)
        _add_obstacle_2d([200, 200], [80, 100], d.map)
        _add_obstacle_2d([160, 80], [30, 80], d.map)
    elif name == "MultiObstacleDataset":
        d = mk(400, 300, -20, -10, 0.1)
        ctr = lambda x, y: [(y - d.origin_y) / d.cell_size, (x - d.origin_x) / d.cell_size]
        dim = lambda w, h: [h / d.cell_size, w / d.cell_size]
        _add_obstacle_2d(ctr(12, 10), dim(5, 7), d.map)
        _add_obstacle_2d(ctr(-7, 10), dim(10, 7), d.map)
        _add_obstacle_2d(ctr(0, -5), dim(10, 5), d.map)
    elif name == "MobileMap1":
        d = mk(500, 500, -5, -5, 0.01)
        ctr = lambda x, y: [(y - d.origin_y) / d.cell_size, (x - d.origin_x) / d.cell_size]
        dim = lambda w, h: [h / d.cell_size, w / d.cell_size]
        _add_obstacle_2d(ctr(0, 0), dim(1, 5), d.map)
        _add_obstacle_2d(ctr(0, 4.5), dim(10, 1), d.map)
        _add_obstacle_2d(ctr(0, -4.5), dim(10, 1), d.map)
        _add_obstacle_2d(ctr(4.5, 0), dim(1, 10), d.map)
        _add_obstacle_2d(ctr(-4.5, 0), dim(1, 10), d.map)
    else:
        raise ValueError("No such dataset exist")
    return d


def _add_obstacle_3d(position, size, m, corners, shift=0):
    h = [int(np.floor((s - 1) / 2)) for s in size]
    lo = [position[k] - h[k] - shift for k in range(3)]  # 1-based inclusive
    hi = [position[k] + h[k] - shift for k in range(3)]
    n = m.shape
    sl = tuple(slice(max(lo[k], 1) - 1, min(hi[k], n[k])) for k in range(3))
    m[sl] = 1.0
    corners.append([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]])


_WAM_DESK = [([170, 220, 130], [140, 60, 5]), ([105, 195, 90], [10, 10, 80]),
             ([235, 195, 90], [10, 10, 80]), ([105, 245, 90], [10, 10, 80]),
             ([235, 245, 90], [10, 10, 80]), ([250, 190, 145], [60, 5, 190]),
             ([250, 90, 145], [60, 5, 190]), ([200, 190, 145], [40, 5, 190]),
             ([250, 140, 240], [60, 100, 5]), ([250, 140, 190], [60, 100, 5]),
             ([250, 140, 140], [60, 100, 5]), ([250, 140, 90], [60, 100, 5])]


def generate3Ddataset(name: str) -> Dataset3D:
    if name == "SmallDemo":
        d = Dataset3D(np.zeros((200, 200, 200)), 200, 200, 200, -1, -1, -1, 0.01)
        _add_obstacle_3d([150, 150, 150], [20, 20, 20], d.map, d.corner_idx)
    elif name == "WAMDeskDataset":
        d = Dataset3D(np.zeros((300, 300, 300)), 300, 300, 300, -1.5, -1.5, -1.5, 0.01)
        for pos, size in _WAM_DESK:
            _add_obstacle_3d(pos, size, d.map, d.corner_idx)
    elif name == "Synth200":
        # BASELINE config 3/4: central crop [51:250]^3 (1-based) of
        # WAMDeskDataset: every obstacle index shifted by -50 and clipped to 1..200.
        d = Dataset3D(np.zeros((200, 200, 200)), 200, 200, 200, -1.0, -1.0, -1.0, 0.01)
        for pos, size in _WAM_DESK:
            _add_obstacle_3d(pos, size, d.map, d.corner_idx, shift=50)
    else:
        raise ValueError("No such dataset exist")
    return d


def _signed_distance(ground_truth_map: np.ndarray, cell_size: float) -> np.ndarray:
    cur = ground_truth_map > 0.75
    if not cur.any():
        return np.ones(ground_truth_map.shape) * 1000.0
    map_dist = ndimage.distance_transform_edt(~cur)   # bwdist(map)
    inv_map_dist = ndimage.distance_transform_edt(cur)  # bwdist(1 - map)
    return ((map_dist - inv_map_dist) * cell_size).astype(np.float64)


def signedDistanceField2D(ground_truth_map, cell_size):
    """field[row = y, col = x]  (signedDistanceField2D.m:14-34)."""
    return _signed_distance(np.asarray(ground_truth_map), cell_size)


def signedDistanceField3D(ground_truth_map, cell_size):
    """field[x, y, z] in the dataset's own index order (signedDistanceField3D.m:14-34)."""
    return _signed_distance(np.asarray(ground_truth_map), cell_size)


def sdf3_zyx(field_xyz: np.ndarray) -> np.ndarray:
    """[x, y, z] field -> contiguous [z][y][x] array (what `field(:,:,z)'` + initFieldData
    builds, matlab/WAMFactorGraphExample.m:63-67)."""
    return np.ascontiguousarray(np.transpose(field_xyz, (2, 1, 0)))
