"""`cali_model_multi` end to end with world size 2 (reference quant/calibration.py:228-389): two processes on cuda:0,
real collectives (gloo moves the device buffers), the tiny DDPM UNet of fixture F8, 3 timestep groups x 32 samples.

Checked against a single-process emulation of the same shards:
  * checkpoint schema == the reference's own (single-GPU) run of F8, rank 0 is the only writer;
  * every reconstruction unit's all-reduced gradient buffer == local(rank 0) + local(rank 1) -- the flatten / ONE SUM
    all-reduce / un-flatten path of engine.recon._Unit.iterate with a real collective;
  * the replicas stay identical: final alphas (hence AdaRound masks) of rank 1 == those rank 0 saved;
  * activation deltas in the checkpoint == the all-average of the deltas each shard yields on its own (emulated in this
    process from the saved weights), zero-points == rank 0's (not synchronised, quant_model.py:127-132)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "f8_cali_tiny.npz")
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def _paths():
    for p in (ROOT, os.path.join(ROOT, "tfmq-dm_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model(g):
    import tfmq_dm_amd.ddim.models as M
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    return m


def _cali_set(g):
    """3 timestep groups x 32 samples (every rank needs >= 16 per group for the 16-sample quantizer initialisation,
    quant/calibration.py:125): F8's three timesteps, fresh noise-like inputs."""
    gen = torch.Generator().manual_seed(11)
    t3 = T(g["cali_t"])[::16][:3].float()
    return torch.randn(96, 3, 16, 16, generator=gen), t3.repeat_interleave(32)


INTERVAL = 32


def _params():
    from quant.quant_layer import QMODE, Scaler
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return wq, aq, [QMODE.NORMAL.value, QMODE.QDIFF.value]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    _paths()
    real_set_device = torch.cuda.set_device
    torch.cuda.set_device = lambda d: real_set_device(0)      # both ranks share the one GPU of the test box
    from quant.calibration import cali_model_multi
    from quant.reconstruction_util import RLOSS
    from tfmq_dm_amd.engine import recon as R
    R._Unit.trace = []
    g = np.load(GOLDEN, allow_pickle=False)
    xs, ts = _cali_set(g)
    wq, aq, mode = _params()
    torch.manual_seed(5)
    np.random.seed(5)
    qnn = cali_model_multi(rank, "gloo", world, f"tcp://127.0.0.1:{port}", 0, world, _model(g), True,
                           os.path.join(out_dir, "multi.pth"), (xs, ts), (xs, ts), INTERVAL, True,
                           dict(wq_params=wq, aq_params=aq, softmax_a_bit=8, aq_mode=mode, iters=10, batch_size=8, w=0.01,
                                asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=True))
    alphas = {k: v.detach().cpu() for k, v in qnn.state_dict().items() if k.endswith("alpha")}
    torch.save({"trace": R._Unit.trace, "alphas": alphas}, os.path.join(out_dir, f"rank{rank}.pt"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_cali_model_multi_world2_matches_emulation(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _paths()
    out = tempfile.mkdtemp()
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    g = golden("f8_cali_tiny")
    ck = torch.load(os.path.join(out, "multi.pth"), map_location="cpu")
    r0, r1 = (torch.load(os.path.join(out, f"rank{r}.pt"), map_location="cpu") for r in range(world))
    # ---- schema: identical to the reference's single-GPU checkpoint of the same model
    assert set(ck["weight"].keys()) == set(str(k) for k in g["weight_keys"])
    assert [k for k in ck if k.startswith("act_")] == ["act_0", "act_1", "act_2"]
    assert sorted(ck["act_0"].keys()) == [str(k) for k in g["act_keys"]]
    # ---- the exchange step: reduced buffer == sum of the two ranks' local buffers, for every unit of the walk
    assert len(r0["trace"]) == len(r1["trace"]) >= 5          # TIB + layer units + block units
    kinds = set()
    for (k0, l0, red0), (k1, l1, red1) in zip(r0["trace"], r1["trace"]):
        assert k0 == k1
        kinds.add(k0)
        assert torch.equal(red0, red1)
        np.testing.assert_allclose(red0.numpy(), (l0 + l1).numpy(), rtol=1e-6, atol=1e-7 * float(red0.abs().max()))
        if k0 != "TibUnit":      # (the TIB sees only the timesteps, which both shards share)
            assert float((l0 - l1).abs().max()) > 0            # the shards really differ
    assert {"TibUnit", "ResnetUnit", "AttnUnit", "LayerUnit"} <= kinds, kinds
    # ---- replicas in lock-step: rank 1's final alphas are what rank 0 wrote
    n_alpha = 0
    for k, a1 in r1["alphas"].items():
        assert torch.equal(a1, ck["weight"][k].reshape(a1.shape)), k
        assert torch.equal(a1, r0["alphas"][k])
        n_alpha += a1.numel()
    assert n_alpha > 0
    # ---- activation deltas: all-average over the shards; zero-points: rank 0's
    from quant.calibration import _calibrate_activations, load_cali_model, shard_for_rank
    from quant.quant_model import QuantModel
    xs, ts = _cali_set(g)
    wq, aq, mode = _params()
    per_rank = []
    for r in range(world):
        qnn = QuantModel(_model(g).to(DEV).eval(), wq, aq, cali=False, aq_mode=mode).to(DEV).eval()
        load_cali_model(qnn, (torch.randn(1, 3, 16, 16), torch.zeros(1)), use_aq=True, path=os.path.join(out, "multi.pth"))
        np.random.seed(5)
        md = {}
        _calibrate_activations(qnn, shard_for_rank((xs, ts), INTERVAL, world, r), INTERVAL // world, True, md)
        per_rank.append(md)
    for gi in range(3):
        act = ck[f"act_{gi}"]
        for k in act:
            v0, v1 = float(per_rank[0][f"act_{gi}"][k]), float(per_rank[1][f"act_{gi}"][k])
            if k.endswith("delta"):
                assert abs(float(act[k]) - 0.5 * (v0 + v1)) <= 1e-6 * abs(v0), (gi, k, float(act[k]), v0, v1)
            else:
                assert float(act[k]) == v0, (gi, k)


def test_rccl_c_abi_world1_allreduce():
    """The C ABI's RCCL wrappers (tfmq_comm_unique_id / tfmq_comm_init / tfmq_allreduce_sum_f32 / tfmq_comm_destroy) as
    linklink drives them: a one-rank communicator on cuda:0 (RCCL refuses two ranks on one device, so the test box
    can only form world 1; the N > 1 form runs in bench.py's calibration leg on the multi-GPU node).  The all-reduce
    is enqueued on the caller's stream between two kernels and must leave the buffer equal to itself."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _paths()
    import torch.distributed as dist
    import tfmq_dm_amd.linklink as link
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd._lib import handle
    port = 31500 + (os.getpid() % 2000)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        torch.cuda.set_device(0)
        link.init_comm(0)
        assert link.comm_device() == 0
        import ctypes as C
        r, w = C.c_int(-1), C.c_int(-1)
        handle(0).call("comm_info", C.byref(r), C.byref(w))
        assert (r.value, w.value) == (0, 1)
        s = torch.cuda.Stream(0)
        with torch.cuda.stream(s):
            x = torch.randn(1 << 20, device=DEV)
            y = x.clone()
            ops.axpy(y, x, 1.0)            # y = 2x on this stream ...
            link.allreduce(y)              # ... the collective is ordered after it on the same stream ...
            ops.axpy(y, x, 1.0)            # ... and before this one: y = 3x
        s.synchronize()
        assert torch.equal(y, 3.0 * x)
        # anything the wrapper does not take (CPU tensor) still goes to torch.distributed
        c = torch.ones(4)
        link.allreduce(c)
        assert torch.equal(c, torch.ones(4))
    finally:
        link.destroy_comm()
        assert link.comm_device() is None
        if own:
            dist.destroy_process_group()


def _rccl_world2_worker(rank, world, port, out_dir):
    """One process per GPU in the REFERENCE's order (rendezvous first, device second): the communicator is created by the
    first device all-reduce, on the tensor's device."""
    _paths()
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import ctypes as C
    import tfmq_dm_amd.linklink as link
    from tfmq_dm_amd._lib import handle
    from tfmq_dm_amd.engine import recon as R
    link.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    assert link.comm_device() is None                      # nothing bound while every rank still sits on device 0
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(1 << 18, generator=g)
    buf = local.to(dev)
    link.allreduce(buf)                                     # creates the communicator on cuda:<rank>, then ncclAllReduce
    torch.cuda.synchronize(rank)
    r, w = C.c_int(-1), C.c_int(-1)
    handle(rank).call("comm_info", C.byref(r), C.byref(w))
    # the chunked side-stream exchange of a reconstruction unit (two pieces): same sums as one flat all-reduce
    cuts = R._chunk_cuts([1 << 16, 1 << 16, 1 << 17], 2)
    flat = local.to(dev)
    side = torch.cuda.Stream(rank)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(rank))
    side.wait_event(ev)
    with torch.cuda.stream(side):
        for (_, _, e0, e1) in cuts:
            link.allreduce(flat[e0:e1])
    side.synchronize()
    # a reconstruction unit's own iterations over RCCL: engine.recon's chunked side-stream exchange (two pieces), each rank on its own shard
    # of the cached inputs; the all-reduced gradients make every rank take the same Adam steps
    import tfmq_dm_amd.ops as ops
    os.environ["TFMQ_EXCHANGE_CHUNKS"] = "2"
    gw = torch.Generator().manual_seed(5)                   # the same weights on every rank

    def ada(cout, cin, k):
        wgt = (torch.randn(cout, cin, k, k, generator=gw) * 0.05).to(dev)
        qp = ops.minmax_to_qparam(ops.minmax(wgt.reshape(cout, -1).contiguous(), cout), 16)
        return R.AdaLayer(wgt, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=dev))
    c1, c2 = ada(32, 32, 3), ada(32, 32, 3)
    gd = torch.Generator().manual_seed(200 + rank)          # a different data shard per rank
    x, y = torch.randn(4, 8, 8, 32, generator=gd).to(dev), torch.randn(4, 8, 8, 32, generator=gd).to(dev)
    gn = (torch.ones(32, device=dev), torch.zeros(32, device=dev))
    unit = R.ResnetUnit(c1, c2, gn, gn, None, x, torch.randn(4, 32, generator=gd).to(dev), y, eps=1e-5, iters=20, world_size=world,
                        allreduce=link.allreduce)
    a0 = [l.alpha.clone() for l in (c1, c2)]
    idx = torch.arange(4, device=dev)
    for _ in range(5):
        unit.iterate(idx)
    torch.cuda.synchronize(rank)
    torch.save({"sum": buf.cpu(), "chunked": flat.cpu(), "info": (r.value, w.value), "comm_dev": link.comm_device(),
                "alpha": [l.alpha.cpu() for l in (c1, c2)], "alpha0": [a.cpu() for a in a0], "cuts": len(unit._cuts)},
               os.path.join(out_dir, f"rccl{rank}.pt"))
    link.barrier()
    link.destroy_comm()


def test_rccl_c_abi_world2_allreduce_when_two_gpus():
    """tfmq_comm_init / tfmq_allreduce_sum_f32 with TWO ranks on two devices (skipped on a 1-GPU box: RCCL refuses two ranks on
    one device) against the sum of the locals: what `bench.py --gpus N` and cali_model_multi rely on at N > 1."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    world, port = 2, 32500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.start_processes(_rccl_world2_worker, args=(world, port, d), nprocs=world, join=True, start_method="spawn")
        res = [torch.load(os.path.join(d, f"rccl{r}.pt")) for r in range(world)]
    ref = sum(torch.randn(1 << 18, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for r in range(world):
        assert res[r]["info"] == (r, world) and res[r]["comm_dev"] == r
        assert torch.equal(res[r]["sum"], ref) and torch.equal(res[r]["chunked"], ref)      # two addends: one rounding, order-free
        assert res[r]["cuts"] == 2
        for a_, a0_, b_ in zip(res[r]["alpha"], res[r]["alpha0"], res[0]["alpha"]):
            assert torch.isfinite(a_).all() and not torch.equal(a_, a0_)                    # the unit trained ...
            assert torch.equal(a_, b_)                                                      # ... and every rank took the same steps


def test_side_stream_exchange_equals_in_stream_exchange(monkeypatch):
    """engine.recon._Unit.iterate with the gradient exchange cut in two pieces on a side stream (the next piece travels while the
    previous piece's fused AdaRound-backward + Adam kernels run) against the one-piece in-stream exchange: identical alphas after
    five iterations.  One device: the 'all-reduce' is a stand-in that doubles the buffer on the CALLER's current stream, which is
    exactly the ordering contract of tfmq_allreduce_sum_f32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _paths()
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd.engine import recon as R
    streams_seen = set()

    def fake_allreduce(t):
        streams_seen.add(torch.cuda.current_stream(0).cuda_stream)
        ops.axpy(t, t.clone(), 1.0)           # t <- 2 t: "sum over two identical ranks"

    def build(chunks):
        monkeypatch.setenv("TFMQ_EXCHANGE_CHUNKS", str(chunks))
        gen = torch.Generator().manual_seed(5)

        def ada(cout, cin, k):
            w = (torch.randn(cout, cin, k, k, generator=gen) * 0.05).to(DEV)
            qp = ops.minmax_to_qparam(ops.minmax(w.reshape(cout, -1).contiguous(), cout), 16)
            return R.AdaLayer(w, qp[:, 0].contiguous(), qp[:, 1].contiguous(), torch.zeros(cout, device=DEV))
        c1, c2 = ada(32, 32, 3), ada(32, 32, 3)
        x = torch.randn(4, 8, 8, 32, generator=gen).to(DEV)
        y = torch.randn(4, 8, 8, 32, generator=gen).to(DEV)
        gn = (torch.ones(32, device=DEV), torch.zeros(32, device=DEV))
        return R.ResnetUnit(c1, c2, gn, gn, None, x, torch.randn(4, 32, generator=gen).to(DEV), y, eps=1e-5, iters=20, world_size=2,
                            allreduce=fake_allreduce), (c1, c2)
    idx = torch.arange(4, device=DEV)
    out = {}
    for chunks in (1, 2):
        streams_seen.clear()
        unit, layers = build(chunks)
        for _ in range(5):
            unit.iterate(idx)
        torch.cuda.synchronize()
        out[chunks] = [l.alpha.clone() for l in layers]
        assert len(unit._cuts) == chunks
        if chunks == 2:
            assert torch.cuda.current_stream(0).cuda_stream not in streams_seen        # the exchange ran on the side stream
    for a, b in zip(out[1], out[2]):
        assert torch.equal(a, b)
