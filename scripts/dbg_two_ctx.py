"""debug: second context in one process, back-to-back frames"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ransac_slam_amd import api, default_config
from ransac_slam_amd.synth import make_frame

def mk(compat, seed=2, H=1000):
    fr = make_frame(L=300, H=H, seed=seed)
    c = api.RslamHip(default_config(compat=compat, adaptive=0))
    c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    c.step_predict(); c.sync()
    ic = fr.ic & c.fetch_prediction()[1]
    c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    return c

def run(c, tag, n=40, chunk=1, graph=True):
    out = []
    if not graph:
        c.enable_timing(True)
    dev = []
    for i in range(n):
        t0 = time.perf_counter()
        for _ in range(chunk):
            c.step_frame(graph)
        try:
            c.sync()
            st = 0
        except api.RslamError as e:
            st = c.last_raw_status()
        out.append(((time.perf_counter() - t0) / chunk * 1e3, st))
        if not graph:
            dev.append(c.timings()["total_us"] * 1e-3)
    bad = [(i, a, s) for i, (a, s) in enumerate(out) if s != 0 or a > 1.0]
    print(tag, "median %.3f" % np.median([a for a, _ in out]), "outliers (frame, ms, raw status):", bad,
          ("device-side max %.3f median %.3f ms" % (max(dev), np.median(dev))) if dev else "")
    if not graph:
        c.enable_timing(False)

import gc
a = mk(1)
b = mk(0)
for rep in range(3):
    run(a, "A graph %d" % rep, n=1000, graph=True)
    run(b, "B graph %d" % rep, n=1000, graph=True)
gc.disable()
for rep in range(3):
    run(a, "A graph gc off %d" % rep, n=1000, graph=True)
    run(b, "B graph gc off %d" % rep, n=1000, graph=True)
