"""Wall-clock of the device SDF construction against scipy on the WAMDeskDataset map (300^3)."""
import sys
import time

sys.path.insert(0, ".")

import numpy as np

import gpmp2_amd as g
from gpmp2_amd.engine import Engine

eng = Engine()
d3 = g.generate3Ddataset("WAMDeskDataset")
occ = g.sdf3_zyx(d3.map)
org = [d3.origin_x, d3.origin_y, d3.origin_z]
for _ in range(3):
    t0 = time.perf_counter()
    h = eng.sdf_from_occupancy(org, d3.cell_size, occ)
    t1 = time.perf_counter()
    print(f"gpu create_from_occupancy {occ.shape}: {1e3 * (t1 - t0):.1f} ms (incl. 216 MB upload)", flush=True)
t0 = time.perf_counter()
want = g.datasets.signedDistanceField3D(d3.map, d3.cell_size)
t1 = time.perf_counter()
print(f"scipy distance_transform_edt x2: {1e3 * (t1 - t0):.1f} ms", flush=True)
np.testing.assert_array_equal(eng.sdf_field(h)["data"], g.sdf3_zyx(want))
print("bit-exact", flush=True)
