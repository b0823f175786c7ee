"""CPU: the trainer's schedule and checkpoint semantics (reference core/trainer/base.py:121-122,180-191,261-292,
core/trainer/warm_up.py, core/utils/ckpt.py) -- no engine call is made here."""
import torch

from core.trainer.base import LinearWarmup
from core.utils.ckpt import CheckPoint


class _Opt(torch.optim.SGD):
    pass


def _sched(lr=1e-3, milestones=(6, 9), warm=4):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = _Opt([{"params": [p], "initial_lr": lr}], lr=lr)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(milestones), gamma=0.1)
    return p, opt, sch, LinearWarmup(opt, warm)


def test_linear_warmup_dampens_the_multistep_schedule_per_iteration():
    """lr_t = initial * gamma^(milestones passed) * min(1, (t + 1) / warmup) -- pytorch_warmup semantics as used by the
    reference: scheduler.step() inside warmup.dampening() once per iteration."""
    _, opt, sch, warm = _sched()
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(11):
        with warm.dampening():
            sch.step()
        lrs.append(opt.param_groups[0]["lr"])
    want = [1e-3 * (0.1 ** ((t >= 6) + (t >= 9))) * min(1.0, (t + 1) / 4) for t in range(12)]
    assert all(abs(a - b) < 1e-12 for a, b in zip(lrs, want)), (lrs, want)


def test_iteration_milestones_and_constant_lr_without_warmup():
    """Milestones in epochs become (m + 1) * len(loader) iterations; without warm-up the scheduler never steps (the
    reference's quirk, base.py:261-263)."""
    from configs import Yolo8DetConfig
    from core.trainer.base import BaseTrainer

    class T(BaseTrainer):
        def set_model_algorithm(self):
            pass

        def load_data(self):
            self.train_dataloader = [None] * 7

        def initialize_model(self):
            self.p = torch.nn.Parameter(torch.zeros(1))

        def set_optimizer(self):
            self.optimizer = _Opt([{"params": [self.p], "initial_lr": 1e-3}], lr=1e-3)

        def set_criterion(self):
            pass

    cfg = Yolo8DetConfig()
    cfg.train.milestones = [1, 3]
    t = T(cfg, "cpu")
    assert t.milestones == [14, 28] and t.last_iter == 0 and t.warmup_scheduler is None


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    model = torch.nn.Linear(3, 2)
    p, opt, sch, warm = _sched()
    for _ in range(3):
        with warm.dampening():
            sch.step()
    path = str(tmp_path / "ck.pth")
    CheckPoint.save(model, path, optimizer=opt, scheduler=sch, warm_up=warm)
    obj = torch.load(path, weights_only=False)
    assert set(obj) == {"model", "optimizer", "scheduler", "warm_up"} and set(obj["model"]) == {"weight", "bias"}
    m2 = torch.nn.Linear(3, 2)
    _, opt2, sch2, warm2 = _sched()
    CheckPoint.load(path, "cpu", m2, optimizer=opt2, scheduler=sch2, warm_up=warm2)
    assert torch.equal(m2.weight, model.weight) and sch2.last_epoch == sch.last_epoch and warm2.last_step == warm.last_step
    bare = str(tmp_path / "bare.pth")
    CheckPoint.save(model, bare)
    m3 = torch.nn.Linear(3, 2)
    CheckPoint.load_pure(bare, "cpu", m3)
    CheckPoint.load_pure(path, "cpu", m3)                       # either format
    assert torch.equal(m3.bias, model.bias)


def test_checkpoint_interchange_of_the_engine_models(tmp_path):
    """`.pth` interchange (core/utils/ckpt.py:38-75): every engine-backed model's state_dict has the reference's keys
    (pinned in the per-model tests), so a checkpoint written by CheckPoint.save loads into a fresh model bit for bit -- through the
    views of the flat arenas, including BatchNorm statistics and num_batches_tracked -- in both of the reference's file formats, and
    FlatAdam's state travels with it."""
    from computervision.pytorch_amd.deeplab import DeepLabV3PlusR101
    from computervision.pytorch_amd.dla import CenterNetDLA34
    from computervision.pytorch_amd.ssd import SSD300VGG
    from computervision.pytorch_amd.train import FlatAdam
    from computervision.pytorch_amd.yolov7 import Yolo7L
    for i, make in enumerate((lambda: DeepLabV3PlusR101(21), lambda: CenterNetDLA34(20), lambda: SSD300VGG(20), lambda: Yolo7L(20))):
        torch.manual_seed(i)
        m = make()
        with torch.no_grad():
            m.flat_stats.add_(torch.rand_like(m.flat_stats))            # statistics that differ from a fresh model's
            m._flat["nbt"] += 7
        opt = FlatAdam(m, lr=3e-4)
        path, bare = str(tmp_path / f"m{i}.pth"), str(tmp_path / f"m{i}_bare.pth")
        CheckPoint.save(m, path, optimizer=opt)
        CheckPoint.save(m, bare)
        torch.manual_seed(100 + i)
        m2 = make()
        assert not torch.equal(m2.flat_params, m.flat_params)
        CheckPoint.load(path, "cpu", m2, optimizer=FlatAdam(m2, lr=1e-3))
        assert torch.equal(m2.flat_params, m.flat_params) and torch.equal(m2.flat_stats, m.flat_stats) and torch.equal(m2._flat["nbt"], m._flat["nbt"])
        m3 = make()
        CheckPoint.load_pure(bare, "cpu", m3)
        sd, sd3 = m.state_dict(), m3.state_dict()
        assert list(sd) == list(sd3) and all(torch.equal(sd[k], sd3[k]) for k in sd)


def test_optimizer_state_interchanges_with_torch_adam(tmp_path):
    """The "optimizer" entry of a full checkpoint is torch.optim.Adam's own layout (the reference saves optimizer.state_dict() of a
    torch.optim.Adam over model.parameters(), core/utils/ckpt.py:41-48, core/trainer/lr_scheduler.py:37-43): a REAL torch.optim.Adam
    checkpoint resumes in FlatAdam, and FlatAdam's checkpoint loads into a real torch.optim.Adam -- moments equal per parameter."""
    from computervision.pytorch_amd.model import Yolo8
    from computervision.pytorch_amd.train import FlatAdam
    torch.manual_seed(0)
    m = Yolo8("n", 80)
    params = list(m.parameters())
    # a reference-style optimizer over tensors of the reference's shapes: three Adam steps on random gradients
    ref_params = [torch.nn.Parameter(p.detach().clone().contiguous(), requires_grad=p.requires_grad) for p in params]
    ref_opt = torch.optim.Adam(ref_params, lr=2e-3)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        for p in ref_params:
            p.grad = torch.randn(p.shape, generator=g) if p.requires_grad else None
        ref_opt.step()
    path = str(tmp_path / "ref_full.pth")
    torch.save({"model": {k: v.detach().clone() for k, v in m.state_dict().items()}, "optimizer": ref_opt.state_dict()}, path)
    opt = FlatAdam(m, lr=1e-3)
    CheckPoint.load(path, "cpu", m, optimizer=opt)                       # used to die with KeyError: 'step'
    assert opt.device_step() == 3 and opt.param_groups[0]["lr"] == 2e-3
    for i, p in enumerate(params):
        if i in ref_opt.state:
            mv = opt._moment_views(p)
            assert torch.equal(mv[0], ref_opt.state[ref_params[i]]["exp_avg"]) and torch.equal(mv[1], ref_opt.state[ref_params[i]]["exp_avg_sq"])
    # the other direction: FlatAdam's file into a fresh torch.optim.Adam
    mine = str(tmp_path / "mine_full.pth")
    CheckPoint.save(m, mine, optimizer=opt)
    sd = torch.load(mine, weights_only=False)["optimizer"]
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(params)))
    fresh = torch.optim.Adam([torch.nn.Parameter(p.detach().clone().contiguous(), requires_grad=p.requires_grad) for p in params], lr=1e-3)
    fresh.load_state_dict(sd)
    for i, p in enumerate(fresh.param_groups[0]["params"]):
        if i in ref_opt.state_dict()["state"]:
            assert torch.equal(fresh.state[p]["exp_avg"], ref_opt.state[ref_params[i]]["exp_avg"]) and float(fresh.state[p]["step"]) == 3.0
    # the round-1/2 flat format still loads
    opt.load_state_dict({"step": 5, "exp_avg": None, "exp_avg_sq": None, "param_groups": [{"lr": 5e-4}]})
    assert opt.device_step() == 5 and float(opt._m.abs().max()) == 0.0
