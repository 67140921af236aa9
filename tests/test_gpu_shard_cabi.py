"""The C-ABI route to the hypothesis-sharded frame: rslam_shard_frame with the RCCL all-gather inside.
One GPU here, so the communicator has one rank (a self-gather through the real ncclAllGather on the
context's stream); the slicing / consensus logic for world_size 2 is covered on CPU by
tests/test_sharded_gloo.py and on one device over gloo by tests/test_gpu_configs.py."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from ransac_slam_amd import default_config
from ransac_slam_amd.synth import make_frame

pytestmark = pytest.mark.gpu


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rccl():
    for name in ("librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.fail("RCCL is part of the image: librccl.so.1 must load")


def _reference(hip, fr, cfg):
    ref = hip.RslamHip(cfg)
    ref.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    ref.step_frame(False); ref.sync()
    full = ref.fetch_results()
    ref.close()
    return full


def test_shard_frame_self_gather():
    from ransac_slam_amd import api as hip
    fr = make_frame(L=60, H=257, seed=302)
    cfg = default_config(compat=0, adaptive=1)
    full = _reference(hip, fr, cfg)
    c = hip.RslamHip(cfg)                               # selects the device before RCCL is initialised
    c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    # world = 1 without a communicator: no collective
    for _ in range(2):
        c.shard_frame(None, 0, 1, True)
    c.sync()
    part = c.fetch_results()
    assert np.array_equal(part["x_new"], full["x_new"]) and np.array_equal(part["P_new"], full["P_new"])
    # a communicator of one rank: the supports go through ncclAllGather on the context's stream
    rccl = _rccl()
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        for use_graph in (False, True, True):
            c.shard_frame(comm.value, 0, 1, use_graph)
        c.sync()
        part = c.fetch_results()
        for k in ("best_hyp", "best_support", "hyps_evaluated", "n_li", "n_hi"):
            assert part[k] == full[k]
        assert np.array_equal(part["li"], full["li"]) and np.array_equal(part["hi"], full["hi"])
        assert np.array_equal(part["x_new"], full["x_new"]) and np.array_equal(part["P_new"], full["P_new"])
        # argument errors
        with pytest.raises(hip.RslamError) as e:
            c.shard_frame(None, 0, 2, True)             # two ranks need a communicator
        assert e.value.code == -1
        with pytest.raises(hip.RslamError) as e:
            c.shard_frame(comm.value, 3, 2, True)
        assert e.value.code == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
        c.close()


@pytest.mark.parametrize("compat", [1, 0])
def test_shard_frame_allreduce_self_reduce(oracle_lib, compat):
    """north_star's literal collective, rslam_shard_frame_allreduce: the slice's best hypothesis as ONE 8-byte key
    (support << 32 | ~index), ncclAllReduce(ncclUint64, ncclMax) on the context's stream, the winner's mask recomputed locally,
    then phase 1.  One GPU: no communicator (world 1) and a communicator of one rank (a real self-reduce through RCCL).  The
    posterior must be the all-gather form's and the whole frame's BIT for bit, the consensus the oracle's earliest strict
    maximum (Tracking.cpp:507-537; compat = 1 makes nearly every support 0 or 1: ties everywhere), and a context with the
    adaptive stop must be refused (RSLAM_ERR_ARG): its evaluated count depends on the whole list."""
    from ransac_slam_amd import api as hip
    fr = make_frame(L=60, H=257, seed=302)
    cfg = default_config(compat=compat, adaptive=0)
    o = oracle_lib.Oracle(cfg, structure=1)
    _, v0, _ = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    fr.ic = ic
    full = _reference(hip, fr, cfg)
    for k in ("best_hyp", "best_support", "hyps_evaluated"):
        assert full[k] == r0[k], k
    c = hip.RslamHip(cfg)
    c.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
    for _ in range(2):
        c.shard_frame_allreduce(None, 0, 1, True)
    c.sync()
    part = c.fetch_results()
    assert np.array_equal(part["x_new"], full["x_new"]) and np.array_equal(part["P_new"], full["P_new"])
    rccl = _rccl()
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        for use_graph in (False, True, True):
            c.shard_frame_allreduce(comm.value, 0, 1, use_graph)
        c.sync()
        red = c.fetch_results()
        c.shard_frame(comm.value, 0, 1, True)
        c.sync()
        gat = c.fetch_results()
        for part in (red, gat):
            for k in ("best_hyp", "best_support", "hyps_evaluated", "n_li", "n_hi"):
                assert part[k] == full[k], k
            assert np.array_equal(part["li"], r0["li"]) and np.array_equal(part["hi"], r0["hi"])
            assert np.array_equal(part["x_new"], full["x_new"]) and np.array_equal(part["P_new"], full["P_new"])
        with pytest.raises(hip.RslamError) as e:
            c.shard_frame_allreduce(comm.value, 0, 2, True)          # a communicator of another shape: as for the all-gather form
        assert e.value.code == -8
        # the adaptive stop and the one-key exchange do not go together
        ca = hip.RslamHip(default_config(compat=compat, adaptive=1))
        ca.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, ic, fr.draws)
        with pytest.raises(hip.RslamError) as e:
            ca.shard_frame_allreduce(comm.value, 0, 1, True)
        assert e.value.code == -1
        ca.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
        c.close()


def test_shard_frame_refuses_a_communicator_of_another_shape():
    """A 1-rank communicator offered as rank 0 (or 1) of a world of 2 must come back as RSLAM_ERR_COMM (-8) before anything
    is enqueued -- on a fresh context, and again on a context that has already validated the SAME communicator for
    (rank 0, world 1): the check is keyed on (communicator, rank, world), not on the pointer alone."""
    from ransac_slam_amd import api as hip
    fr = make_frame(L=40, H=64, seed=303)
    cfg = default_config(compat=0, adaptive=1)
    full = _reference(hip, fr, cfg)
    rccl = _rccl()
    fresh = hip.RslamHip(cfg)
    fresh.load_frame(fr.types, fr.x_pred, fr.P_pred, fr.z, fr.ic, fr.draws)
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        for rank in (0, 1):
            with pytest.raises(hip.RslamError) as e:
                fresh.shard_frame(comm.value, rank, 2, False)
            assert e.value.code == -8
        assert fresh.counters()["graph_captures"] == 0      # refused before anything of the frame was captured or enqueued
        fresh.shard_frame(comm.value, 0, 1, False)          # the right shape passes ...
        fresh.sync()
        with pytest.raises(hip.RslamError) as e:            # ... and does not whitewash the wrong one afterwards
            fresh.shard_frame(comm.value, 0, 2, False)
        assert e.value.code == -8
        fresh.shard_frame(comm.value, 0, 1, True)
        fresh.sync()
        part = fresh.fetch_results()
        assert np.array_equal(part["x_new"], full["x_new"]) and np.array_equal(part["P_new"], full["P_new"])
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
        fresh.close()


@pytest.mark.parametrize("compat,mode", [(1, "allgather"), (0, "allgather"), (0, "allreduce"), (1, "allreduce")])
def test_cpp_shard_frame_example(oracle_lib, tmp_path, compat, mode):
    """host/shard_frame_example.cpp (C++, owns the ncclComm_t) as rank 0 of 1, against the oracle: the all-gather form
    (adaptive stop replayed on the gathered list) and the one-key all-reduce form (every draw evaluated)."""
    from ransac_slam_amd import build
    build.build()
    exe = build.build_shard_example()
    fr = make_frame(L=60, H=1100, seed=501, frac_ic=0.9)
    cfg = default_config(compat=compat, adaptive=0 if mode == "allreduce" else 1)
    o = oracle_lib.Oracle(cfg, structure=1)
    h0, v0, S0 = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    fin, fout = tmp_path / "frame.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("4i", fr.n, fr.L, len(fr.draws), compat))
        f.write(fr.types.tobytes()); f.write(fr.ic.astype(np.uint8).tobytes())
        f.write(fr.x_pred.tobytes()); f.write(np.asfortranarray(fr.P_pred).tobytes(order="F"))
        f.write(np.ascontiguousarray(fr.z).tobytes()); f.write(fr.draws.tobytes())
    argv = [exe, str(fin), str(fout)] + (["0", "1", str(tmp_path / "nccl_id"), "0", "allreduce"] if mode == "allreduce" else [])
    subprocess.check_call(argv, timeout=300)
    raw = open(fout, "rb").read()
    n, L = fr.n, fr.L
    sc = np.frombuffer(raw, np.int32, 3); p = 12
    li = np.frombuffer(raw, np.uint8, L, p); p += L
    hi = np.frombuffer(raw, np.uint8, L, p); p += L
    vis = np.frombuffer(raw, np.uint8, L, p); p += L
    p += 16 * L + 32 * L
    x = np.frombuffer(raw, np.float64, n, p); p += 8 * n
    P = np.frombuffer(raw, np.float64, n * n, p).reshape(n, n, order="F")
    assert np.array_equal(vis, v0)
    assert list(sc) == [r0["best_hyp"], r0["best_support"], r0["hyps_evaluated"]]
    assert np.array_equal(li, r0["li"]) and np.array_equal(hi, r0["hi"])
    assert np.max(np.abs(x - r0["x_new"])) <= 1e-9 * max(1.0, np.abs(r0["x_new"]).max())
    assert np.max(np.abs(P - r0["P_new"])) <= 1e-9 * np.abs(r0["P_new"]).max()


@pytest.mark.timeout(600)
def test_cpp_shard_frame_two_ranks_one_device(oracle_lib, tmp_path):
    """rslam_shard_frame with world = 2: two processes (the C++ example as rank 0 and rank 1), each with its own context and
    ONE RCCL communicator of two ranks -- both on device 0, the only device of the test box.  If this RCCL refuses two
    ranks on one device (NCCL's "Duplicate GPU detected": ncclCommInitRank fails on both ranks) the test says so and is
    skipped: the two-rank slicing / tail padding / consensus replay is then covered over gloo (tests/test_gpu_configs.py,
    two ranks on one device) and on CPU (tests/test_sharded_gloo.py), the RCCL call itself by the one-rank self-gather."""
    from ransac_slam_amd import build
    build.build()
    exe = build.build_shard_example()
    fr = make_frame(L=60, H=1001, seed=502, frac_ic=0.9)          # 1001 draws: the second slice is one short (tail padding)
    cfg = default_config(compat=0, adaptive=1)
    o = oracle_lib.Oracle(cfg, structure=1)
    _, v0, _ = o.predict(fr.types, fr.x_pred, fr.P_pred)
    ic = (fr.ic & v0).astype(np.uint8)
    r0 = o.ransac_update(fr.z, ic, fr.draws)
    fin = tmp_path / "frame.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("4i", fr.n, fr.L, len(fr.draws), 0))
        f.write(fr.types.tobytes()); f.write(fr.ic.astype(np.uint8).tobytes())
        f.write(fr.x_pred.tobytes()); f.write(np.asfortranarray(fr.P_pred).tobytes(order="F"))
        f.write(np.ascontiguousarray(fr.z).tobytes()); f.write(fr.draws.tobytes())
    idfile = tmp_path / "nccl_id"
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(fin), str(tmp_path / f"out{r}.bin"), str(r), "2", str(idfile), "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL run on one device hung")
    if any(p.returncode != 0 for p in procs):
        text = "\n".join(outs)
        if "uplicate GPU" in text or "invalid usage" in text.lower() or "ncclCommInitRank" in text:
            why = [ln for ln in text.splitlines() if "uplicate" in ln or "ncclCommInitRank" in ln or "invalid usage" in ln.lower()]
            pytest.skip("this RCCL refuses two ranks on one device: " + (why[0].strip()[-200:] if why else "ncclCommInitRank failed"))
        pytest.fail(text[-2000:])
    n, L = fr.n, fr.L
    for r in range(2):
        raw = open(tmp_path / f"out{r}.bin", "rb").read()
        sc = np.frombuffer(raw, np.int32, 3); p = 12
        li = np.frombuffer(raw, np.uint8, L, p); p += L
        hi = np.frombuffer(raw, np.uint8, L, p); p += L
        p += L + 16 * L + 32 * L
        x = np.frombuffer(raw, np.float64, n, p); p += 8 * n
        P = np.frombuffer(raw, np.float64, n * n, p).reshape(n, n, order="F")
        assert list(sc) == [r0["best_hyp"], r0["best_support"], r0["hyps_evaluated"]], r
        assert np.array_equal(li, r0["li"]) and np.array_equal(hi, r0["hi"])
        assert np.max(np.abs(x - r0["x_new"])) <= 1e-9 * max(1.0, np.abs(r0["x_new"]).max())
        assert np.max(np.abs(P - r0["P_new"])) <= 1e-9 * np.abs(r0["P_new"]).max()
