"""Generates tests/golden/vdiffusion_golden.pt from the LIVE reference modules
(/root/reference/audio_diffusion_pytorch/diffusion.py and utils.py, loaded by oracle/reference_loader.py).
Run in the build container only (the GPU box has no /root/reference):  python tests/golden/make_golden.py

The U-Net half of the path has no importable reference (a_unet is absent: SURVEY.md section 0), so the fixtures pin
exactly what CAN be pinned against the reference itself: the v-objective loss, the sampler loop, the schedule
endpoint constants and the sinc resampler, all driven through a deterministic stub network."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.reference_loader import load_reference  # noqa: E402


class StubNet(torch.nn.Module):
    """Deterministic stand-in for the U-Net: mixes channels, depends on time, keeps the [B,C,L] shape."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([[0.6, -0.3], [0.2, 0.9]]))

    def forward(self, x, t, **kw):
        return torch.einsum("oc,bcl->bol", self.w, x) * (0.5 + t.view(-1, 1, 1)) + 0.1 * torch.roll(x, 1, dims=-1)


def main():
    D, U = load_reference()
    g = torch.Generator().manual_seed(1234)
    out = {}
    # schedule + endpoint constants
    out["linspace_51"] = D.LinearSchedule()(51, device="cpu")
    samp = D.VSampler(StubNet())
    a, b = samp.get_alpha_beta(out["linspace_51"])
    out["alpha_51"], out["beta_51"] = a, b
    # VDiffusion loss with injected sigmas / noise
    x = torch.randn(3, 2, 64, generator=g)
    sig = torch.tensor([0.0, 0.37, 1.0])

    class Fixed(D.Distribution):
        def __call__(self, num_samples, device=torch.device("cpu")):
            return sig[:num_samples]

    net = StubNet()
    diff = D.VDiffusion(net, sigma_distribution=Fixed())
    torch.manual_seed(99)
    noise = torch.randn_like(x)   # the draw VDiffusion.forward makes at diffusion.py:88 under this seed
    torch.manual_seed(99)
    loss = diff(x)
    out.update(vd_x=x, vd_sigmas=sig, vd_noise=noise, vd_loss=loss.detach())
    # VSampler
    n0 = torch.randn(2, 2, 64, generator=g)
    out["vs_noise"] = n0
    for steps in (1, 5, 50):
        out[f"vs_out_{steps}"] = samp(n0, num_steps=steps)
    # VInpainter (diffusion.py:306-354): seeded global generator -> the x_noisy draw and the per-resample draws
    src = torch.randn(2, 2, 64, generator=g)
    mask = torch.zeros(2, 2, 64, dtype=torch.bool)
    mask[..., :24] = True
    out["vi_source"], out["vi_mask"] = src, mask
    inp = D.VInpainter(StubNet())
    torch.manual_seed(77)
    out["vi_out_4_3"] = inp(src, mask, num_steps=4, num_resamples=3)
    torch.manual_seed(77)
    out["vi_out_6_1"] = inp(src, mask, num_steps=6, num_resamples=1)
    # resampler
    w = torch.randn(2, 2, 512, generator=g)
    out["rs_in"] = w
    for f in (2, 4, 16):
        out[f"rs_down_{f}"] = U.downsample(w, f)
        out[f"rs_up_{f}"] = U.upsample(w[..., :64], f)
    # kwargs routing
    out["groupby"] = U.groupby("diffusion_", dict(diffusion_a=1, sampler_b=2, c=3))
    out["closest_power_2"] = [U.closest_power_2(v) for v in (3.0, 5.9, 6.1, 1000.0)]
    torch.save(out, os.path.join(os.path.dirname(os.path.abspath(__file__)), "vdiffusion_golden.pt"))
    print("wrote", len(out), "entries; alpha0 =", float(a[0]), "sigma[1] =", float(out["linspace_51"][1]))


if __name__ == "__main__":
    main()
