"""CPU, world_size 2 over gloo: the data-parallel exchange step (bucketed mean all-reduce of the flat
gradient arena) and the sharded-oracle identity it must satisfy."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from computervision.pytorch_amd.train import allreduce_mean_flat, bucket_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, n_buckets, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(n, generator=g)
        allreduce_mean_flat(grads, world, None, n_buckets, None)
        if rank == 0:
            torch.save(grads, out)
    finally:
        dist.destroy_process_group()


def test_bucket_bounds_cover_and_align():
    for n in (1, 5, 1000, 3157184 + 12):
        for nb in (1, 3, 4, 7):
            b = bucket_bounds(n, nb)
            assert b[0][0] == 0 and b[-1][1] == n and len(b) <= nb
            for (s0, e0), (s1, e1) in zip(b, b[1:]):
                assert e0 == s1 and s1 % 4 == 0


def test_allreduce_mean_two_ranks(tmp_path):
    n, world = 100003, 2
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, 4, out), nprocs=world, join=True)
    got = torch.load(out)
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    assert torch.allclose(got, want, atol=1e-6)


class _OracleEngine:
    """CPU stand-in for the HIP engine behind the segmented-backward interface (include/cvx_engine.h): the per-shard
    gradients come from the oracle, laid out in the engine's flat arena; ``backward_range`` releases the arena slice of
    the ops it covers into ``g`` -- exactly when the real engine would have finished them."""

    def __init__(self, layout, graph, full_arena, g):
        self.layout, self.graph, self.full, self.g = layout, graph, full_arena, g
        self.next = len(graph.ops) - 1
        self.ready = []

    def backward_begin(self, dpred, loss_scale):
        self.next = len(self.graph.ops) - 1

    def backward_range(self, op_hi, op_lo):
        assert op_hi == self.next and 0 <= op_lo <= op_hi
        p0, p1 = self.slices[(op_hi, op_lo)]
        self.g[p0:p1] += self.full[p0:p1]
        self.next = op_lo - 1

    def grads_ready(self, op_hi, op_lo, stream):
        assert self.next < op_lo
        self.ready.append((op_hi, op_lo))

    def backward_end(self):
        assert self.next == -1


def _dp_worker(rank, world, port, fixture, out):
    import numpy as np
    from computervision.pytorch_amd.graph import ParamLayout, build_yolov8_graph, grad_buckets
    from computervision.pytorch_amd.train import backward_with_overlapped_exchange
    from oracle import yolov8_ref as O
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f = np.load(fixture)
        x, bi, cls, bb = (torch.from_numpy(f[k]) for k in ("x", "batch_idx", "cls", "bboxes"))
        per = x.shape[0] // world
        sel = (bi >= rank * per) & (bi < (rank + 1) * per)
        shard = {"batch_idx": bi[sel] - rank * per, "cls": cls[sel], "bboxes": bb[sel]}
        grads = O.train_step(O.init_state_dict("n", 80, seed=0), x[rank * per:(rank + 1) * per], shard, {})[2]
        lay = ParamLayout("n", 80)
        graph = build_yolov8_graph(lay, 96, 96)
        g = torch.zeros(lay.n_params)
        eng = _OracleEngine(lay, graph, lay.scatter(grads), g)
        buckets = grad_buckets(graph, lay, 5)
        eng.slices = {(hi, lo): (p0, p1) for hi, lo, p0, p1 in buckets}
        backward_with_overlapped_exchange(eng, buckets, g, None, 1.0, None, None)
        assert eng.ready == [(hi, lo) for hi, lo, _, _ in buckets]
        g /= world                                                        # the 1/world the fused Adam kernel applies
        if rank == 0:
            torch.save({k: v.clone() for k, v in lay.views(g).items()}, out)
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_delivers_the_reference_shard_mean(tmp_path, gold):
    """The bucket loop of the data-parallel step (train.backward_with_overlapped_exchange -- the code the GPU path runs)
    under gloo with two ranks: the averaged arena must EQUAL the fixture the real reference produced shard by shard
    (tests/golden/dp_sim_96.npz, oracle/make_golden.py section 7): mean over ranks of per-shard gradients, each shard with
    its own loss normaliser and BN batch statistics (core/algorithms/yolo_v8.py:109,124)."""
    import numpy as np
    f = gold("dp_sim_96.npz")
    world = 2
    out = str(tmp_path / "mean.pt")
    path = os.path.join(os.path.dirname(__file__), "golden", "dp_sim_96.npz")
    mp.spawn(_dp_worker, args=(world, _free_port(), path, out), nprocs=world, join=True)
    mean = torch.load(out)
    keys = [str(k) for k in f["keys"]]
    flat = torch.cat([mean[k].flatten() for k in keys])
    ref_sub = torch.from_numpy(f[f"w{world}_sub"])
    assert float((flat[::211] - ref_sub).norm() / ref_sub.norm()) < 2e-4
    assert abs(float(flat.norm()) / float(f[f"w{world}_norm"]) - 1) < 2e-4
    for k in f.files:
        if k.startswith(f"w{world}:"):
            ref = torch.from_numpy(f[k])
            assert float((mean[k.split(":", 1)[1]] - ref).norm() / (ref.norm() + 1e-12)) < 2e-4, k
    # and it is a different quantity from the single-process step on the whole batch (world 1 in the same fixture)
    one = torch.from_numpy(f["w1_sub"])
    assert float((flat[::211] - one).norm() / one.norm()) > 0.1


def test_sharded_oracle_matches_reference_dp_simulation(gold):
    """4 and 8 shards without processes: the oracle run shard by shard and averaged equals the reference's own shard-by-
    shard run for every world size of the fixture (the gloo test above covers the exchange itself for world 2)."""
    from oracle import yolov8_ref as O
    f = gold("dp_sim_96.npz")
    x, bi, cls, bb = (torch.from_numpy(f[k]) for k in ("x", "batch_idx", "cls", "bboxes"))
    keys = [str(k) for k in f["keys"]]
    for world in (4, 8):
        per = 8 // world
        acc = None
        for r in range(world):
            sel = (bi >= r * per) & (bi < (r + 1) * per)
            sb = {"batch_idx": bi[sel] - r * per, "cls": cls[sel], "bboxes": bb[sel]}
            gr = O.train_step(O.init_state_dict("n", 80, seed=0), x[r * per:(r + 1) * per], sb, {})[2]
            acc = gr if acc is None else {k: acc[k] + gr[k] for k in keys}
        flat = torch.cat([(acc[k] / world).flatten() for k in keys])
        ref = torch.from_numpy(f[f"w{world}_sub"])
        assert float((flat[::211] - ref).norm() / ref.norm()) < 2e-4, world


def _bn_worker(rank, world, port, out):
    from computervision.pytorch_amd.train import broadcast_bn_statistics
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class _M:                                                  # the two arenas broadcast_bn_statistics touches
            flat_stats = torch.full((64,), float(rank + 1))
            _flat = {"nbt": torch.full((5,), rank + 7, dtype=torch.long)}
        broadcast_bn_statistics(_M, None, src=0)
        if rank == 1:
            torch.save((_M.flat_stats, _M._flat["nbt"]), out)
    finally:
        dist.destroy_process_group()


def test_bn_statistics_broadcast_before_checkpoint(tmp_path):
    """Per-rank BN running statistics diverge (no SyncBN in the reference); rank 0's are broadcast before a checkpoint."""
    out = str(tmp_path / "bn.pt")
    mp.spawn(_bn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    stats, nbt = torch.load(out)
    assert bool((stats == 1.0).all()) and bool((nbt == 7).all())


def test_grad_buckets_tile_ops_and_arena():
    """Units of the overlapped exchange (cvx_engine_backward_range / cvx_engine_grads_ready): op ranges in backward order
    whose parameters are contiguous, disjoint arena slices covering every parameter exactly once."""
    from computervision.pytorch_amd.graph import ParamLayout, build_yolov8_graph, grad_buckets
    for mt in ("n", "s"):
        lay = ParamLayout(mt, 80)
        g = build_yolov8_graph(lay, 256, 256)
        for nb in (1, 2, 4, 7, 100):
            b = grad_buckets(g, lay, nb)
            assert 1 <= len(b) <= nb
            if nb > 1:
                assert b[-1][0] == 1 and b[-1][1] == 0                      # the stem (model.0, model.1) closes the pass alone
            assert b[0][0] == len(g.ops) - 1 and b[-1][1] == 0            # ops: last ... first
            assert b[-1][2] == 0 and b[0][3] == lay.n_params               # arena: [0, n_params)
            for (hi0, lo0, s0, e0), (hi1, lo1, s1, e1) in zip(b, b[1:]):
                assert hi1 == lo0 - 1 and e1 == s0 and s1 % 4 == 0
            # every conv's parameters lie inside the slice of the bucket that runs the op
            for i, o in enumerate(g.ops):
                if o["type"] != 1:
                    continue
                spec = lay.convs[o["name"]]
                (hi, lo, s, e), = [x for x in b if x[1] <= i <= x[0]]
                assert s <= spec.w_off and (spec.beta_off if spec.bn else spec.bias_off) + spec.cout <= e
