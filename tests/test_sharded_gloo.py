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
                if self.cfg.adaptive:
                    n_hyp = self.po.adaptive_n_hyp(self.cfg.p_success, best, n_ic)
                    if n_hyp == 0:
                        break
            if i > n_hyp:
                break
            i += 1
        self.result = (besti, best, ev)


class ListEngine:
    """Test double whose supports are a given list: ties inside a slice and across the slice boundary."""

    def __init__(self, supports):
        self.sup = list(supports)
        self.H = len(self.sup)
        self.device = torch.device("cpu")
        self.result = None

    def step_predict(self):
        pass

    def step_score(self, b, e, local):
        if e > b:
            local[:e - b] = torch.tensor(self.sup[b:e], dtype=torch.int32)

    def step_update(self, supports_all):
        # Tracking.cpp:507-537 without the adaptive stop: the earliest strict maximum of the whole list
        sup = supports_all[:self.H].tolist()
        best, besti = 0, -1
        for i, v in enumerate(sup):
            if v > best:
                best, besti = int(v), i
        self.result = (besti, best, self.H)


TIE_LISTS = [[0, 3, 7, 2, 7, 7, 1, 0],          # the maximum twice in the first slice's successor: index 2 wins
             [1, 1, 1, 1, 9, 2, 9, 9],          # only the second slice holds it: its FIRST 9
             [4, 4, 4, 4, 4, 4, 4],             # all equal (odd length: the second slice is one short): index 0
             [0, 0, 0, 0, 0, 0],                # nobody has an inlier: no winner at all (support 0 never beats 0)
             [0, 0, 0, 5]]                      # the very last hypothesis


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
        # the exchange as ONE 8-byte MAX all-reduce (mode="allreduce", adaptive = 0) beside the all-gather of the same frame
        pairs = []
        for H in (64, 101):
            fr = make_frame(L=20, H=H, seed=77)
            cfg = default_config(compat=0, adaptive=0)
            res = {}
            for mode in ("allgather", "allreduce"):
                eng = OracleEngine(fr, cfg)
                sf = ShardedFrame(eng, mode=mode)
                sf.step()
                res[mode] = (eng.result, sf.all[:H].tolist())
            pairs.append((H, res))
        ties = []
        for lst in TIE_LISTS:
            res = {}
            for mode in ("allgather", "allreduce"):
                eng = ListEngine(lst)
                ShardedFrame(eng, mode=mode).step()
                res[mode] = eng.result
            ties.append(res)
        q.put((rank, (out, pairs, ties)))
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
            Hh, result, gathered = got[rank][0][idx]
            assert Hh == H
            assert gathered == sup.tolist()        # same list on every rank, hypothesis i at index i
            assert result == want                  # identical consensus on every rank
        # ... and without the adaptive stop: the all-reduce form (8 bytes on the wire) decides what the all-gather form decides,
        # which is the oracle's earliest strict maximum (Tracking.cpp:507-537)
        o0 = oracle_lib.Oracle(default_config(compat=0, adaptive=0), structure=1)
        o0.predict(fr.types, fr.x_pred, fr.P_pred)
        r0 = o0.ransac_only(fr.z, fr.ic, fr.draws)
        want0 = (r0["best_hyp"], r0["best_support"], r0["hyps_evaluated"])
        for rank in (0, 1):
            Hh, res = got[rank][1][idx]
            assert Hh == H
            assert res["allgather"][0] == want0 and res["allreduce"][0] == want0
            assert res["allgather"][1] == sup.tolist()
            onehot = [0] * H
            if want0[1] > 0:
                onehot[want0[0]] = want0[1]
            assert res["allreduce"][1] == onehot    # what crossed the wire was one key: the list is the winner's support, nothing else
    # ties: inside a slice, across the slice boundary, everywhere, nowhere
    for rank in (0, 1):
        for lst, res in zip(TIE_LISTS, got[rank][2]):
            best = max(lst)
            want_t = (lst.index(best) if best > 0 else -1, best, len(lst))
            assert res["allgather"] == want_t and res["allreduce"] == want_t, (lst, res)
