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


def test_sharded_oracle_gradient_identity():
    """DP parity definition: mean over ranks of the per-shard gradients (each shard normalises its loss by
    its own target_scores_sum and batch size, yolo_v8.py:109,124) is what the all-reduce must deliver --
    and it is NOT the single-process full-batch gradient.  Checked on the oracle (tiny input)."""
    from oracle import synth
    from oracle import yolov8_ref as O
    x, batch = synth.images(4, 64, 64, seed=1), synth.targets(4, seed=2)
    full = O.train_step(O.init_state_dict("n", 80, seed=0), x, batch, {})[2]
    shards = []
    for r in range(2):
        sel = (batch["batch_idx"] >= 2 * r) & (batch["batch_idx"] < 2 * r + 2)
        sb = {"batch_idx": batch["batch_idx"][sel] - 2 * r, "cls": batch["cls"][sel], "bboxes": batch["bboxes"][sel]}
        shards.append(O.train_step(O.init_state_dict("n", 80, seed=0), x[2 * r:2 * r + 2], sb, {})[2])
    k = "model.22.cv3.0.2.bias"
    mean = (shards[0][k] + shards[1][k]) / 2
    assert torch.isfinite(mean).all() and mean.abs().sum() > 0
    assert not torch.allclose(mean, full[k], rtol=1e-3)      # per-shard normalisers + per-shard BN statistics


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
