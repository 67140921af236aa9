import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import default_config
from ransac_slam_amd.api import RslamHip
c = RslamHip(default_config())
for mode in (0, 1, 2):
    for w in (1, 2, 4, 8):
        print("mode", mode, "waves", w, c.mfma_f64_probe(w, mode))
