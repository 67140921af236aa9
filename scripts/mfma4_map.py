import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ransac_slam_amd import default_config
from ransac_slam_amd.api import RslamHip
c = RslamHip(default_config())
z = np.zeros(64)
# which (a-lane, b-lane) pairs feed which d-lane: one-hot a, all-ones b with distinct values
def contrib(cbsz, abid):
    amap = {}
    for la in range(64):
        a = np.zeros(64); a[la] = 1.0
        b = np.arange(1, 65, dtype=float)          # b value identifies the b lane
        d = c.mfma4_raw(a, b, z, cbsz, abid)
        for ld in np.flatnonzero(d):
            amap.setdefault(int(ld), []).append((la, int(round(d[ld])) - 1))
    return amap
for cbsz, abid in ((0, 0), (2, 0), (2, 1), (2, 3), (1, 0), (1, 1)):
    m = contrib(cbsz, abid)
    print("cbsz", cbsz, "abid", abid)
    for ld in (0, 1, 2, 3, 4, 5, 16, 17, 20, 33, 63):
        print("  d lane", ld, "<- (a lane, b lane):", m.get(ld))
