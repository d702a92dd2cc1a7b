"""SWAG (Maddox et al. 2019) around a VanillaRNN: the weight-space sampler of the reference's utils/swag_utils.py
(`SwagModel` :160-335, `swa_update` / `second_moment_update` :120-140), re-designed so that all MC samples of a
prediction run through ONE launch of the per-sample-weights LSTM kernel -- the path BayesianRNN already uses --
instead of one deep-copied module and one forward per sample (validate_rnn.get_batch_predictions_SWAG :89-112).

Kept from the reference: the interface (`SwagModel(model)`, `update_parameters(model)`, `sample(scale, cov)`,
`forward`, buffers `n_averaged`, `sec_moment_<i>`, `dev_<i>`, sub-module `module`, so state dicts interchange) and the
sampling formula
    w = mean + sqrt(scale) * ( sqrt(clamp(E[w^2] - mean^2, 1e-30)) * z1  +  D z2 / sqrt(n_averaged - 2) ),
with z1 ~ N(0, I_P), z2 ~ N(0, I_K), D the (P, K) matrix of deviations collected from the second update on.
`draws = "host"` draws z1 then z2 per sample with the reference's torch calls on the CPU generator (parity mode)."""
from copy import deepcopy

import torch
from torch import nn

from .. import engine as E


class SwagModel(nn.Module):
    draws = "device"

    def __init__(self, model, device=None):
        super().__init__()
        self.module = deepcopy(model)
        if device is not None:
            self.module = self.module.to(device)
        self.register_buffer("n_averaged", torch.tensor(0, dtype=torch.long, device=device))
        for i, p in enumerate(self.module.parameters()):
            self.register_buffer(f"sec_moment_{i}", torch.zeros_like(p))
            self.register_buffer(f"dev_{i}", p.new_zeros((p.numel(), 0)))

    def forward(self, *args, **kwargs):
        """The SWA (mean-weights) model, as the reference's SwagModel.forward (:221-222)."""
        return self.module(*args, **kwargs)

    @torch.no_grad()
    def update_parameters(self, model):
        """Running mean and second moment of the parameters; from the second call on also the deviation of the
        new parameters from the NEW mean, one column per call (swag_utils.py:224-266)."""
        n = int(self.n_averaged)
        for i, (p_swa, p_model) in enumerate(zip(self.module.parameters(), model.parameters())):
            p_new = p_model.detach().to(p_swa.device)
            if n == 0:
                p_swa.copy_(p_new)
                setattr(self, f"sec_moment_{i}", p_new * p_new)
            else:
                p_swa.copy_(p_swa + (p_new - p_swa) / (n + 1))
                m2 = getattr(self, f"sec_moment_{i}")
                setattr(self, f"sec_moment_{i}", m2 + (p_new * p_new - m2) / (n + 1))
                setattr(self, f"dev_{i}", torch.cat((getattr(self, f"dev_{i}"), (p_new - p_swa).view(-1, 1)), dim=1))
        for b_swa, b_model in zip(self.module.buffers(), model.buffers()):
            b_swa.detach().copy_(b_model.detach().to(b_swa.device))
        self.n_averaged += 1

    # -- sampling ----------------------------------------------------------------------------------------------
    def _stats(self):
        params = list(self.module.parameters())
        mean = torch.cat([p.detach().reshape(-1) for p in params])
        m2 = torch.cat([getattr(self, f"sec_moment_{i}").reshape(-1) for i in range(len(params))])
        dev = torch.cat([getattr(self, f"dev_{i}") for i in range(len(params))], dim=0)
        return mean, m2, dev

    @torch.no_grad()
    def sample_flat(self, n_samples, scale=0.5, cov=True, var_clamp=1e-30):
        """(n_samples, P) sampled weight vectors in module.parameters() order (= the kernels' canonical order)."""
        mean, m2, dev = self._stats()
        std = torch.clamp(m2 - mean ** 2, var_clamp).sqrt()
        P, K = mean.numel(), dev.size(1)
        if cov and int(self.n_averaged) < 3:
            raise ValueError("SWAG covariance sampling needs at least 3 averaged models (n_averaged - 2 > 0)")
        if self.draws == "host":        # the reference's draw order on the CPU generator: z1 (P), then z2 (K), per sample
            z1, z2 = [], []
            for _ in range(n_samples):
                z1.append(torch.randn(P))
                if cov:
                    z2.append(torch.empty(K).normal_())
            z1 = torch.stack(z1).to(mean.device)
            z2 = torch.stack(z2).to(mean.device) if cov else None
        else:
            z1 = torch.randn((n_samples, P), device=mean.device)
            z2 = torch.randn((n_samples, K), device=mean.device) if cov else None
        rand = std * z1
        if cov:
            rand = rand + (z2 @ dev.t()) / float(int(self.n_averaged) - 2) ** 0.5
        return (mean + scale ** 0.5 * rand).contiguous()

    @torch.no_grad()
    def sample(self, scale=0.5, cov=True, var_clamp=1e-30):
        """One sampled model as a module of its own (the reference's API, swag_utils.py:268-335)."""
        flat = self.sample_flat(1, scale, cov, var_clamp)[0]
        model, off = deepcopy(self.module), 0
        for p in model.parameters():
            p.copy_(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return model

    # -- one launch for all samples (the MC loop of validate_rnn.py:444-508) -------------------------------------
    swag_scale, swag_cov = 0.5, True
    inference_precision = "f32"

    def infer_cfg(self, masked=False):
        self.module.inference_precision = self.inference_precision
        return self.module.infer_cfg(masked)

    def mc_weights(self, n_samples, n_curves, device, mean_field_inference=False):
        """(kernel-layout weights (S, PK), no masks): what validation.validate_rnn.predict_mc asks a model for.
        mean_field_inference -> the SWA mean weights for every sample."""
        if mean_field_inference:
            return self.module.mc_weights(n_samples, n_curves, device)[0], None
        flats = self.sample_flat(n_samples, self.swag_scale, self.swag_cov).to(device)
        return E.prepare_weights_multi(self.infer_cfg(), flats), None

    def forward_mc(self, x, n_samples, scale=None, cov=None):
        """All n_samples SWAG forwards of one batch in one launch -> probs (B, n_samples, C)."""
        padded, lengths = self.module._unpack(x)
        cfg = self.infer_cfg()
        flats = self.sample_flat(n_samples, self.swag_scale if scale is None else scale,
                                 self.swag_cov if cov is None else cov).to(padded.device)
        plan = E.Plan(cfg, lengths, padded.device)
        _, probs = E.lstm_forward(cfg, plan, E.pack_x(cfg, plan, padded), E.prepare_weights_multi(cfg, flats), n_samples,
                                  want_logits=False, want_probs=True, sample_major=False)
        return probs
