"""Fused AdamW over the model's flat parameter arena (one HIP launch for all 53 M parameters) with the
torch.optim.Optimizer interface the reference's training loop uses (transduction_model.py:178-189:
param_groups[...]['lr'] writes, zero_grad(), step(); ReduceLROnPlateau works on it unchanged)."""
import torch

from . import torch_ops  # noqa: F401  (registers torch.ops.silent_speech.*)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        params = model.optimized_parameters()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._m = self._v = None
        self._t = 0

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() plus the fused state (first / second moments over the flat arena, step count):
        the moments live outside Optimizer.state, so the base class alone would silently drop them."""
        sd = super().state_dict()
        sd['fused'] = {'step': self._t, 'exp_avg': None if self._m is None else self._m.detach().cpu().clone(),
                       'exp_avg_sq': None if self._v is None else self._v.detach().cpu().clone()}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop('fused', None)
        super().load_state_dict(state_dict)
        if fused is not None:
            flat, _, n = self.model.flat_arenas()
            self._t = int(fused['step'])
            if fused['exp_avg'] is not None:
                if fused['exp_avg'].numel() != n:
                    raise ValueError('optimizer state of %d parameters, model has %d' % (fused['exp_avg'].numel(), n))
                self._m = fused['exp_avg'].to(flat.device, torch.float32).clone()
                self._v = fused['exp_avg_sq'].to(flat.device, torch.float32).clone()

    def zero_grad(self, set_to_none=False):
        flat, gflat, n = self.model.flat_arenas()
        gflat.zero_()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        flat, gflat, n = self.model.flat_arenas()
        if self._m is None or self._m.device != flat.device or self._m.numel() != n:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdamW updates the whole flat arena with one set of hyper-parameters: exactly one param group')
        g = self.param_groups[0]
        self._t += 1
        torch.ops.silent_speech.fused_adamw(flat, gflat, self._m, self._v, int(n), float(g['lr']), int(self._t), float(g['betas'][0]), float(g['betas'][1]),
                                            float(g['eps']), float(g['weight_decay']), float(grad_scale))
        self.model.mark_weights_updated()
