"""N > 1 path on CPU: world_size-2 gloo run of the hypothesis-sharding driver with an
oracle-backed engine (the HIP engine needs a GPU; the sharding logic does not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ransac_slam_amd import default_config
from ransac_slam_amd.sharded import ShardedFrame, slice_bounds
from ransac_slam_amd.synth import make_frame


def test_slice_bounds_cover_and_order():
    for H in (0, 1, 7, 1000, 1001, 4000):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, e, chunk = slice_bounds(H, r, world)
                assert 0 <= b <= e <= H and e - b <= chunk
                assert b == min(H, r * chunk)          # global index = rank * chunk + local index
                got += list(range(b, e))
            assert got == list(range(H))


class OracleEngine:
    """Test double with the Engine protocol: scores a slice with the CPU oracle."""

    def __init__(self, frame, cfg):
        from oracle import pyoracle as po
        self.po, self.frame, self.cfg = po, frame, cfg
        self.H = len(frame.draws)
        self.device = torch.device("cpu")
        self.result = None

    def step_predict(self):
        self.o = self.po.Oracle(default_config(compat=self.cfg.compat, adaptive=0), structure=1)
        self.o.predict(self.frame.types, self.frame.x_pred, self.frame.P_pred)

    def step_score(self, b, e, local):
        if e > b:
            self.o.ransac_only(self.frame.z, self.frame.ic, self.frame.draws[b:e])
            sup, _, _ = self.o.supports()
            local[:e - b] = torch.from_numpy(sup.copy())

    def step_update(self, supports_all):
        # replay of Tracking.cpp:403,507-537 on the gathered list (what K5 does on every rank)
        sup = supports_all[:self.H].numpy()
        n_ic = int(self.frame.ic.sum())
        n_hyp, best, besti, ev, i = self.cfg.n_hyp_init, 0, -1, 0, 0
        while i < n_hyp and i < self.H:
            ev = i + 1
            if sup[i] > best:
                best, besti = int(sup[i]), i
                n_hyp = self.po.adaptive_n_hyp(self.cfg.p_success, best, n_ic)
                if n_hyp == 0:
                    break
            if i > n_hyp:
                break
            i += 1
        self.result = (besti, best, ev)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for H in (64, 101):
            fr = make_frame(L=20, H=H, seed=77)
            cfg = default_config(compat=0, adaptive=1)
            eng = OracleEngine(fr, cfg)
            sf = ShardedFrame(eng)
            sf.step()
            out.append((H, eng.result, sf.all[:H].tolist() if world > 1 else sf.local[:H].tolist()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(oracle_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process oracle with the sequential adaptive loop
    for idx, H in enumerate((64, 101)):
        fr = make_frame(L=20, H=H, seed=77)
        o = oracle_lib.Oracle(default_config(compat=0, adaptive=0), structure=1)
        o.predict(fr.types, fr.x_pred, fr.P_pred)
        o.ransac_only(fr.z, fr.ic, fr.draws)
        sup, _, _ = o.supports()
        oa = oracle_lib.Oracle(default_config(compat=0, adaptive=1), structure=1)
        oa.predict(fr.types, fr.x_pred, fr.P_pred)
        ra = oa.ransac_only(fr.z, fr.ic, fr.draws)
        want = (ra["best_hyp"], ra["best_support"], ra["hyps_evaluated"])
        for rank in (0, 1):
            Hh, result, gathered = got[rank][idx]
            assert Hh == H
            assert gathered == sup.tolist()        # same list on every rank, hypothesis i at index i
            assert result == want                  # identical consensus on every rank
