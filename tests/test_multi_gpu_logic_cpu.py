"""N>1 path on CPU (gloo, world_size 2): calibration-set sharding (reference calibration.py:269-282,
fixture F10), the SUM all-reduce of a unit's flattened gradient buffer and the all-average of the
activation deltas (linklink shim), checked against a single-process emulation of the same shards."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_for_rank_matches_reference_indices(golden):
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    from quant.calibration import shard_for_rank
    g = golden("f10_shards")
    for I, W in ((256, 8), (512, 8), (256, 4), (16, 2)):
        data = torch.arange(I * 3)
        for r in range(W):
            (sh,) = shard_for_rank((data,), I, W, r)
            assert sh.tolist() == list(g[f"I{I}_W{W}_r{r}"])
        # the shards of all ranks partition every timestep group
        allidx = torch.cat([shard_for_rank((data,), I, W, r)[0] for r in range(W)])
        assert sorted(allidx.tolist()) == list(range(I * 3))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    import linklink as link
    import linklink.dist_helper as dh
    from quant.calibration import shard_for_rank
    link.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    assert link.get_world_size() == world and link.get_rank() == rank
    g = torch.Generator().manual_seed(0)
    data = torch.randn(3 * 16, 5, generator=g)            # 3 timestep groups of 16 samples
    (mine,) = shard_for_rank((data,), 16, world, rank)
    # per-rank "gradient" of two tensors, flattened into one buffer, ONE all-reduce (K16)
    w1, w2 = torch.randn(5, 4, generator=g), torch.randn(5, 3, generator=g)
    g1, g2 = mine.t() @ (mine @ w1), mine.t() @ (mine @ w2)
    flat = torch.cat([g1.reshape(-1), g2.reshape(-1)])
    link.allreduce(flat)
    # activation delta all-average
    delta = torch.tensor([float(rank + 1)])
    dh.allaverage(delta)
    torch.save({"flat": flat, "delta": delta, "n": mine.shape[0]}, os.path.join(out_dir, f"r{rank}.pt"))
    link.barrier()


def test_gloo_world2_allreduce_matches_single_process_emulation(tmp_path):
    world, port = 2, 29000 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    from quant.calibration import shard_for_rank
    g = torch.Generator().manual_seed(0)
    data = torch.randn(3 * 16, 5, generator=g)
    w1, w2 = torch.randn(5, 4, generator=g), torch.randn(5, 3, generator=g)
    ref = torch.zeros(5 * 4 + 5 * 3)
    for r in range(world):   # single-process emulation: loop over the same shards, sum the gradients
        (sh,) = shard_for_rank((data,), 16, world, r)
        ref += torch.cat([(sh.t() @ (sh @ w1)).reshape(-1), (sh.t() @ (sh @ w2)).reshape(-1)])
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert res["n"] == 24
        np.testing.assert_allclose(res["flat"].numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
        assert abs(float(res["delta"]) - 1.5) < 1e-6       # (1 + 2) / 2


def _lazy_comm_worker(rank, world, port, out_dir, same_device=False):
    """What a spawned calibration rank does in the reference's order (quant/calibration.py:241-245): rendezvous FIRST, device
    chosen afterwards.  The RCCL communicator of the C ABI must not be bound at the rendezvous (every rank would still be on
    device 0: 'Duplicate GPU detected' / a hang) but at the first device all-reduce, to that tensor's device, with the id
    handed round through the store.  No GPU here: the handle / library calls are recorded by stand-ins."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    import types
    import linklink as link
    from tfmq_dm_amd import _lib
    calls = []

    class FakeHandle:
        comm_world = 0

        def __init__(self, dev):
            self.dev = dev

        def call(self, name, *a):
            calls.append((name, self.dev) + tuple(a[1:3] if name == "comm_init" else ()))
            if name == "comm_init":
                self.ident = bytes(a[0])

    class FakeLib:
        @staticmethod
        def tfmq_comm_unique_id(buf):
            for i in range(128):
                buf[i] = (i * 7 + 3) % 251
            return 0

    _lib._handles.clear()

    def fake_handle(dev=0):
        if dev not in _lib._handles:
            _lib._handles[dev] = FakeHandle(dev)
        return _lib._handles[dev]
    _lib.handle, _lib.load = fake_handle, (lambda: FakeLib)
    torch.cuda.is_available = lambda: True
    link._device_identity = lambda d: f"fake-host|fake-gpu-{d}"      # (no GPU here: the identity the ranks publish before ncclCommInitRank)
    real_init = link.dist.init_process_group
    link.dist.init_process_group = lambda backend=None, **kw: real_init(backend="gloo", **kw)    # "nccl" asked for, gloo underneath
    link.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    assert calls == [] and link.comm_device() is None          # nothing bound at the rendezvous
    dev = 2 if same_device else rank + 2                       # the device this rank picks afterwards

    class FakeDevTensor:
        is_cuda, dtype = True, torch.float32
        device = types.SimpleNamespace(index=dev)

        def is_contiguous(self):
            return True

        def data_ptr(self):
            return 4096

        def numel(self):
            return 8
    torch.cuda.current_stream = lambda d=None: types.SimpleNamespace(cuda_stream=0)
    if same_device:
        # two ranks on ONE GPU: refused on every rank before any RCCL call (ncclCommInitRank would sit in its bootstrap)
        from tfmq_dm_amd._lib import TfmqError
        try:
            link.allreduce(FakeDevTensor())
            err = None
        except TfmqError as e:
            err = str(e)
        torch.save({"calls": calls, "err": err}, os.path.join(out_dir, f"dup{rank}.pt"))
        link.barrier()
        return
    link.allreduce(FakeDevTensor())
    link.allreduce(FakeDevTensor())
    h = _lib._handles[dev]
    torch.save({"calls": calls, "ident": h.ident, "comm_dev": link.comm_device()}, os.path.join(out_dir, f"lazy{rank}.pt"))
    link.barrier()


def test_rccl_communicator_binds_lazily_to_the_tensors_device(tmp_path):
    world, port = 2, 31000 + (os.getpid() % 2000)
    mp.start_processes(_lazy_comm_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    res = [torch.load(os.path.join(str(tmp_path), f"lazy{r}.pt")) for r in range(world)]
    for r in range(world):
        assert res[r]["comm_dev"] == r + 2
        names = [c[0] for c in res[r]["calls"]]
        assert names == ["comm_init", "allreduce_sum_f32", "allreduce_sum_f32"]          # created once, on first use
        assert res[r]["calls"][0][1:] == (r + 2, r, world)                              # device, rank, world
    assert res[0]["ident"] == res[1]["ident"] == bytes((i * 7 + 3) % 251 for i in range(128))   # rank 0's id reached rank 1 via the store


def test_exchange_chunk_cuts_partition_the_gradient_buffer():
    """engine.recon._chunk_cuts: contiguous pieces at layer boundaries, every layer in exactly one piece, never more pieces than asked."""
    sys.path.insert(0, ROOT)
    from tfmq_dm_amd.engine.recon import _chunk_cuts
    rng = np.random.default_rng(0)
    for _ in range(200):
        sizes = [int(s) for s in rng.integers(1, 5000, size=int(rng.integers(1, 12)))]
        n = int(rng.integers(1, 6))
        cuts = _chunk_cuts(sizes, n)
        assert 1 <= len(cuts) <= min(n, len(sizes))
        assert cuts[0][0] == 0 and cuts[0][2] == 0 and cuts[-1][1] == len(sizes) and cuts[-1][3] == sum(sizes)
        for (l0, l1, e0, e1), nxt in zip(cuts, cuts[1:] + [None]):
            assert l1 > l0 and e1 - e0 == sum(sizes[l0:l1])
            if nxt is not None:
                assert nxt[0] == l1 and nxt[2] == e1
    assert _chunk_cuts([10, 10, 10, 10], 2) == [(0, 2, 0, 20), (2, 4, 20, 40)]


def test_rccl_communicator_refuses_two_ranks_on_one_gpu(tmp_path):
    """tfmq_comm_init at world > 1 with a duplicate device fails loudly on every rank instead of hanging -- the ranks
    compare device identities through the rendezvous store before librccl is entered (linklink.init_comm)."""
    world, port = 2, 33000 + (os.getpid() % 2000)
    mp.start_processes(_lazy_comm_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        res = torch.load(os.path.join(str(tmp_path), f"dup{r}.pt"))
        assert res["err"] is not None and "share a GPU" in res["err"] and "(0, 1)" in res["err"], res
        assert not any(c[0] == "comm_init" for c in res["calls"])           # librccl was never entered


def test_identity_match_contradicted_by_the_local_index_is_not_a_refusal():
    """Two ranks of one host that see the same visible-device mask and chose different indices cannot share a GPU: if the driver reports the
    same identity for both (an all-zero uuid on a build without PCI ids), the match is dropped instead of refusing a real 8-GPU node."""
    import tfmq_dm_amd.linklink as link
    assert link._contradicted((0, 1), ["||| 0".replace(" ", ""), "|||1"])             # same mask, indices 0 / 1
    assert not link._contradicted((0, 1), ["|||0", "|||0"])                            # same mask, same index: a real duplicate
    assert not link._contradicted((0, 1), ["0|||0", "1|||0"])                          # per-rank masks (index 0 on both): the identity decides
    assert link._device_hint(3).endswith("|3")


def test_duplicate_devices_is_a_pure_function():
    sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
    import linklink as link
    assert link.duplicate_devices(["a", "b", "c"]) == []
    assert link.duplicate_devices(["a", "b", "a", "b"]) == [(0, 2), (1, 3)]
