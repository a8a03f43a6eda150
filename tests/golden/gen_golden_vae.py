"""Golden vectors for the first-stage decoder, produced by the reference's own Decoder.

    python tests/golden/gen_golden_vae.py

Tiny KL-style first stage: Decoder(ch 32, mult [1,2], 1 res block per level, z_channels 4, resolution 16) behind a
1x1 post_quant_conv, i.e. AutoencoderKL.decode (ldm/models/autoencoder.py:329-332: post_quant_conv then decoder;
the class itself derives from pytorch_lightning, absent here, so its two-line decode is spelled out) and
LatentDiffusion.decode_first_stage's 1/scale_factor (ldm/models/diffusion/ddpm.py:706-708)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import save  # noqa: E402

DD = dict(ch=32, out_ch=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=[], dropout=0.0, in_channels=3,
          resolution=16, z_channels=4, double_z=True)
SCALE = 0.18215


def main():
    from ldm.modules.diffusionmodules.model import Decoder
    torch.manual_seed(33)
    dec = Decoder(**DD).eval()
    pq = torch.nn.Conv2d(4, DD["z_channels"], 1).eval()
    H.rerandomize_zero_params(dec, seed=4)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(3, 4, 8, 8, generator=g) * SCALE * 4.0
    with torch.no_grad():
        q = pq(z / SCALE)
        dec.give_pre_end = True
        pre = dec(q)
        dec.give_pre_end = False
        img = dec(q)
    out = {"z": z, "scale_factor": torch.tensor(SCALE), "pre_end": pre, "img": img}
    out.update({"sd/decoder." + k: v.detach().clone() for k, v in dec.state_dict().items()})
    out.update({"sd/post_quant_conv." + k: v.detach().clone() for k, v in pq.state_dict().items()})
    save("f14_vae_decoder_tiny", **out)


if __name__ == "__main__":
    main()
