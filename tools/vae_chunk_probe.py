"""VAE decode of a 16-frame clip at 512x512 by chunk size (frames per decoder call): the 512^2-level tensors of 8 frames are
537 MB each - twice the 256 MB Infinity Cache; smaller chunks keep producer -> consumer traffic in it.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import v_express_amd as vx
    from v_express_amd import synth
    dev = torch.device("cuda", 0)
    elem = torch.bfloat16
    vcfg = synth.VaeConfig()
    vae = vx.AutoencoderKLDecoder(vcfg).to(dev).to(elem)
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg, seed=44, device=dev, dtype=elem, draw_on_device=True))
    vae._prepared()
    lat = torch.randn(1, 4, 16, 64, 64, device=dev)
    ref = None
    for chunk in (8, 4, 2, 1, 16, 8):
        for _ in range(2):
            out = vae.decode_video(lat, chunk=chunk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = vae.decode_video(lat, chunk=chunk)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        print(f"chunk {chunk:2d}: {e0.elapsed_time(e1) / 3:7.2f} ms per 16-frame clip   same bits as chunk 8: {torch.equal(out, ref)}", flush=True)


if __name__ == "__main__":
    main()
