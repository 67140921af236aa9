"""MI355X-native 1-point-RANSAC EKF update (hot path of plumewind/ransac_slam).

The product is the C-ABI library ``librslam_hip.so`` (include/rslam.h) built from
``ransac_slam_amd/csrc``; this package holds its ctypes binding, the synthetic
frame generator and the multi-GPU hypothesis sharding driver.
"""
from .ctypes_defs import (Camera, Config, Layout, StageTimes, default_camera,  # noqa: F401
                          default_config, make_layout)
