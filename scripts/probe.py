import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import default_config
from ransac_slam_amd.api import RslamHip
c = RslamHip(default_config())
for w in (1, 2, 3, 4, 6, 8):
    print(w, c.mfma_f64_probe(w))
for b in (1 << 28, 1 << 30, 1 << 32):
    print("copy", b, c.hbm_copy_peak(b))
