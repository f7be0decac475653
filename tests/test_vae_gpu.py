"""SD-VAE decoder on the device: reed_amd/vae.py running on the GPU against oracle/vae.py (numpy fp64, an
independent restatement walking the checkpoint's keys) — until round 3 the pair only met on the CPU.  PARITY UNPINNED against
diffusers itself (neither the package nor a checkpoint is available offline): both files say so."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,shape", [(dict(block_out_channels=(16, 32, 32), layers_per_block=1, norm_num_groups=8), (2, 4, 5, 6)),
                                       (dict(block_out_channels=(32, 64, 64, 64), layers_per_block=2, norm_num_groups=16), (1, 4, 8, 8))])
def test_sd_vae_decoder_on_gpu_vs_oracle(dev, cfg, shape):
    from oracle import vae as ovae
    from reed_amd import vae as rvae
    torch.manual_seed(1)
    dec = rvae.SDVAEDecoder(**cfg)
    for p in dec.parameters():
        p.data.normal_(0, 0.15)
    z = torch.randn(*shape)
    want = ovae.Decoder({k: v.double().numpy() for k, v in dec.state_dict().items()}, groups=cfg["norm_num_groups"]).decode(
        z.double().numpy())
    with torch.no_grad():
        got = dec.to(dev).decode(z.to(dev)).float().cpu().numpy()
    up = 2 ** (len(cfg["block_out_channels"]) - 1)
    assert got.shape == (shape[0], 3, shape[2] * up, shape[3] * up)
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    print(f"SD-VAE decoder on the GPU (fp32 convolutions) vs the fp64 oracle: max abs {err:.3e} on outputs of scale {scale:.2f}")
    assert err <= 2e-4 * scale
