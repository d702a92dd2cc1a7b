"""Fused optimizers over the module's flat parameter buffer (one kernel launch per step), with the
data-parallel gradient all-reduce folded in.
  FusedAdam  torch.optim.Adam semantics (utils/training_utils.py:365-369): lr, betas (0.9, 0.999), eps 1e-8, L2 weight_decay
  FusedSGD   torch.optim.SGD semantics with momentum (training/train_rnn.py:541-546, train_swag_finetune)"""
import torch

from . import engine as E


class _FlatOptimizer:
    _buffers = ()

    def __init__(self, model, process_group=None, **hyper):
        self.model = model
        self.param_groups = [dict(params=list(model.parameters()), **hyper)]
        self.state = dict(step=0, **{k: None for k in self._buffers})
        self.process_group = process_group

    # -- torch.optim.Optimizer surface used by the reference's loops
    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def _flat_grads(self):
        """Gradients in flat-buffer order; zero-copy when autograd left them as views of one buffer."""
        ps = self.model._flat_params()
        g0 = ps[0].grad
        if g0 is None:
            raise RuntimeError(f"{type(self).__name__}.step(): parameters have no gradient (call loss.backward() first)")
        ptr, contiguous = g0.data_ptr(), True
        for p in ps:
            g = p.grad
            if g is None or g.data_ptr() != ptr or g.dtype != torch.float32 or not g.is_contiguous():
                contiguous = False
                break
            ptr += g.numel() * 4
        if contiguous:
            n = sum(p.numel() for p in ps)
            base = g0.reshape(-1)
            if base.untyped_storage().nbytes() - base.storage_offset() * 4 >= n * 4:
                return torch.as_strided(base, (n,), (1,))
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])

    def step(self):
        self.step_flat(self._flat_grads())

    def step_flat(self, grads, mean=True):
        """Update from a gradient already laid out like model.flat_parameters().  Under torch.distributed the ranks'
        gradients are summed by ONE all-reduce and, with mean=True, divided by the world size (exact for equal shards);
        mean=False takes the plain sum of contributions the caller has weighted itself (fused_train_step: any split)."""
        from .distributed import allreduce_mean_scale
        flat = self.model.flat_parameters()
        if grads.numel() != flat.numel():
            raise RuntimeError(f"{type(self).__name__}: flat gradient size does not match the parameters")
        # ONE all-reduce of the whole gradient (160-320 KB at H=32): latency-bound over xGMI, so a single call
        scale = allreduce_mean_scale(grads, self.process_group, mean=mean)
        st = self.state
        for k in self._buffers:
            if st[k] is None or st[k].numel() != flat.numel() or st[k].device != flat.device:
                st[k] = torch.zeros_like(flat)
        st["step"] += 1
        self._update(flat, grads, scale)

    def state_dict(self):
        return dict(state={k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state.items()},
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.state.update(sd["state"])
        self.param_groups[0].update(sd["param_groups"][0])


class FusedAdam(_FlatOptimizer):
    _buffers = ("exp_avg", "exp_avg_sq")

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None):
        super().__init__(model, process_group, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def _update(self, flat, grads, scale):
        g, st = self.param_groups[0], self.state
        E.adam_step(flat, grads, st["exp_avg"], st["exp_avg_sq"], st["step"], g["lr"], g["betas"], g["eps"],
                    g["weight_decay"], scale)


class FusedSGD(_FlatOptimizer):
    _buffers = ("momentum_buffer",)

    def __init__(self, model, lr=1e-3, momentum=0.0, weight_decay=0.0, process_group=None):
        super().__init__(model, process_group, lr=lr, momentum=momentum, weight_decay=weight_decay)

    def _update(self, flat, grads, scale):
        g, st = self.param_groups[0], self.state
        E.sgd_step(flat, grads, st["momentum_buffer"], st["step"], g["lr"], g["momentum"], g["weight_decay"], scale)
