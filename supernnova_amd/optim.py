"""Fused Adam over the module's flat parameter buffer (one kernel launch per step), with the
data-parallel gradient all-reduce folded in.  torch.optim.Adam semantics
(utils/training_utils.py:365-369): lr, betas (0.9, 0.999), eps 1e-8, L2 weight_decay."""
import torch

from . import engine as E


class FusedAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, process_group=None):
        self.model = model
        self.param_groups = [dict(params=list(model.parameters()), lr=lr, betas=betas, eps=eps,
                                  weight_decay=weight_decay)]
        self.state = dict(step=0, exp_avg=None, exp_avg_sq=None)
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
            raise RuntimeError("FusedAdam.step(): parameters have no gradient (call loss.backward() first)")
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

    def step_flat(self, grads):
        """Adam update from a gradient already laid out like model.flat_parameters()."""
        from .distributed import allreduce_mean_scale
        g = self.param_groups[0]
        flat = self.model.flat_parameters()
        if grads.numel() != flat.numel():
            raise RuntimeError("FusedAdam: flat gradient size does not match the parameters")
        # ONE all-reduce of the whole gradient (160-320 KB at H=32): latency-bound over xGMI, so a single call
        scale = allreduce_mean_scale(grads, self.process_group)
        st = self.state
        if st["exp_avg"] is None or st["exp_avg"].numel() != flat.numel() or st["exp_avg"].device != flat.device:
            st["exp_avg"] = torch.zeros_like(flat)
            st["exp_avg_sq"] = torch.zeros_like(flat)
        st["step"] += 1
        E.adam_step(flat, grads, st["exp_avg"], st["exp_avg_sq"], st["step"], g["lr"], g["betas"], g["eps"],
                    g["weight_decay"], scale)

    def state_dict(self):
        return dict(state={k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state.items()},
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.state.update(sd["state"])
        self.param_groups[0].update(sd["param_groups"][0])
