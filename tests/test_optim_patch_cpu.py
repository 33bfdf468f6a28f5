"""install_dropin(fused_adam=True): torch.optim.Adam hands out FusedAdamL1 only where FusedAdamL1 can stand in for it; on
CPU parameters it must stay torch's own Adam (reconstruction/main_nerf.py:119 is the call it serves)."""
import torch

import trinerflet_amd
from trinerflet_amd import optim


def test_patched_adam_is_torch_adam_on_cpu_parameters_and_unpatch_restores_it():
    real = torch.optim.Adam
    try:
        trinerflet_amd.install_backends()
        optim.patch_torch_adam()
        assert torch.optim.Adam is not real and issubclass(torch.optim.Adam, real)
        p = [torch.nn.Parameter(torch.zeros(4))]
        o = torch.optim.Adam(p, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        assert type(o) is real and o.defaults["betas"] == (0.9, 0.99) and o.defaults["eps"] == 1e-15
        p[0].grad = torch.ones(4)
        o.step()
        assert torch.allclose(p[0].detach(), torch.full((4,), -1e-2))
        g = torch.optim.Adam([{"params": p, "lr": 3e-3}], lr=1e-2, amsgrad=True)
        assert type(g) is real and g.param_groups[0]["lr"] == 3e-3 and g.defaults["amsgrad"]
        optim.patch_torch_adam()                     # idempotent
    finally:
        optim.unpatch_torch_adam()
    assert torch.optim.Adam is real
