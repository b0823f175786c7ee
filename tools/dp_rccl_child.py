"""One rank of the 2-rank RCCL data-parallel check (tests/test_gpu_parity.py::test_two_rank_rccl_step_against_the_dp_fixture starts two FRESH
processes of this script -- a process that has touched the GPU is never re-executed).  Each rank runs its shard of the fixture batch
through the engine, exchanges the gradients with cvx_engine_backward_exchange (RCCL behind the C ABI; `--exchange torch`: the
torch.distributed loop) and writes its averaged gradients; rank 0 compares them with tests/golden/dp_sim_96.npz (the reference run
shard by shard, gradients averaged).   python tools/dp_rccl_child.py <rank> <world> <port> <out_dir> [c|torch]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    exchange = sys.argv[5] if len(sys.argv) > 5 else "c"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl" if exchange == "torch" else "gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            **({"device_id": dev} if exchange == "torch" else {}))
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import CvxComm, FlatAdam, FusedTrainStep, V8DetectionLoss
    from configs import Yolo8DetConfig
    g = np.load(os.path.join(ROOT, "tests", "golden", "dp_sim_96.npz"))
    x = torch.from_numpy(g["x"])
    B = x.shape[0]
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    bi = torch.from_numpy(g["batch_idx"])
    sel = (bi >= lo) & (bi < hi)
    batch = {"batch_idx": (bi[sel] - lo).to(dev), "cls": torch.from_numpy(g["cls"])[sel].to(dev), "bboxes": torch.from_numpy(g["bboxes"])[sel].to(dev)}
    torch.manual_seed(0)
    m = Yolo8("n", 80).to(dev).train()
    comm = CvxComm(dev) if exchange == "c" else None
    opt = FlatAdam(m, lr=0.0)                       # lr 0: the step leaves the parameters alone, the exchanged gradients are what is compared
    step = FusedTrainStep(m, V8DetectionLoss(Yolo8DetConfig(), m), opt, n_buckets=4, comm=comm)
    # gradients are zeroed by the fused Adam launch: keep a copy taken after the exchange, before the optimiser step
    grabbed = {}
    orig = opt.step

    def grab(*a, **k):
        torch.cuda.current_stream(dev).synchronize()
        grabbed["g"] = (m.flat_grads / world).clone()
        return orig(*a, **k)
    opt.step = grab
    items = step(x[lo:hi].to(dev), batch)
    torch.cuda.synchronize(dev)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), grads=grabbed["g"].cpu().numpy(), items=items.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
