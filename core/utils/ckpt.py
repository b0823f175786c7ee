"""Checkpoint files in the reference's format (core/utils/ckpt.py:7-75): ``torch.save`` of either a bare ``state_dict`` or
``{"model", "optimizer", "scheduler", "warm_up"}``; the model's ``state_dict`` has the reference's keys and shapes and the
optimizer entry is ``torch.optim.Adam``'s own layout (per-parameter ``state[i] = {step, exp_avg, exp_avg_sq}`` in
``model.parameters()`` order -- ``FlatAdam.state_dict`` slices its flat moment arenas accordingly), so full checkpoints are
interchangeable in both directions: a reference ``.pth`` resumes here, and a file written here loads into the reference's
``torch.optim.Adam`` (tests/test_trainer_cpu.py).

One deliberate difference: the reference loads the ``warm_up`` entry into the *scheduler* (ckpt.py:65-66, a bug that
overwrites the scheduler state just restored); here it goes to the warm-up object it was saved from.
"""
import os

import numpy as np
import torch


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


class CheckPoint:
    @staticmethod
    def check(path):
        return bool(path) and os.path.exists(path)

    @staticmethod
    def load_pretrained(model, weights):
        """Shape-filtered partial load (reference ckpt.py:19-36)."""
        assert CheckPoint.check(weights), f"The pretrained model weights {weights} does not exist."
        model_dict = model.state_dict()
        pretrained = torch.load(weights, map_location="cpu")
        if isinstance(pretrained, dict) and "model" in pretrained and not torch.is_tensor(pretrained["model"]):
            pretrained = pretrained["model"]
        ok, skipped, take = [], [], {}
        for k, v in pretrained.items():
            if k in model_dict and np.shape(model_dict[k]) == np.shape(v):
                take[k] = v
                ok.append(k)
            else:
                skipped.append(k)
        model_dict.update(take)
        model.load_state_dict(model_dict)
        print(f"Successfully loaded {len(ok)} keys, they are: {ok[:20]}...")
        print(f"Failed to load {len(skipped)} keys, they are: {skipped[:20]}...")

    @staticmethod
    def save(model, path, optimizer=None, scheduler=None, warm_up=None):
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}   # plain tensors: no arena views in the file
        if optimizer is None and scheduler is None:
            torch.save(sd, path)
            return
        obj = {"model": sd}
        if optimizer is not None:
            obj["optimizer"] = _to_cpu(optimizer.state_dict())
        if scheduler is not None:
            obj["scheduler"] = scheduler.state_dict()
        if warm_up is not None:
            obj["warm_up"] = warm_up.state_dict()
        torch.save(obj, path)

    @staticmethod
    def load(path, device, model, pure=False, optimizer=None, scheduler=None, warm_up=None):
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        if pure:
            model.load_state_dict(ckpt)
            return
        model.load_state_dict(ckpt["model"])
        if optimizer is not None and "optimizer" in ckpt:
            optimizer.load_state_dict(ckpt["optimizer"])
        if scheduler is not None and "scheduler" in ckpt:
            scheduler.load_state_dict(ckpt["scheduler"])
        if warm_up is not None and "warm_up" in ckpt:
            warm_up.load_state_dict(ckpt["warm_up"])

    @staticmethod
    def load_pure(path, device, model):
        """Accepts either file format (reference ckpt.py:70-75)."""
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(ckpt, dict) and "model" in ckpt and not torch.is_tensor(ckpt["model"]):
            ckpt = ckpt["model"]
        model.load_state_dict(ckpt)
