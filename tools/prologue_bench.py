"""Once-per-clip prologue at full size on one MI355X (reported separately from the fps metric):
VAE encode of the reference image, VKpsGuider on the clip's keypoint images, wav2vec2-base + interpolation/windows +
AudioProjection on the clip's audio.  Synthetic weights; prints one JSON line.
usage: python tools/prologue_bench.py [frames] [size]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v_express_amd as vx  # noqa: E402
from v_express_amd import synth  # noqa: E402
from v_express_amd.prologue import audio_windows  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


def main():
    F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    fps = 25
    g = torch.Generator().manual_seed(0)
    res = {"frames": F_, "size": size, "audio_seconds": F_ / fps}
    vcfg = synth.VaeConfig()
    vae = vx.AutoencoderKL(vcfg).to("cuda")
    vae.load_state_dict({**synth.vae_decoder_state_dict(vcfg), **synth.vae_encoder_state_dict(vcfg)})
    img = (torch.rand(1, 3, size, size, generator=g) * 2 - 1).cuda()
    res["vae_encode_ms"], _ = timed(lambda: vae.encode(img).latent_dist.mean)
    del vae
    guider = vx.VKpsGuider(320, block_out_channels=(16, 32, 96, 256)).to("cuda")
    guider.load_state_dict(synth.kps_guider_state_dict())
    kps = torch.rand(1, 3, F_, size, size, generator=g).cuda()
    res["kps_guider_ms"], _ = timed(lambda: guider.forward_tokens(kps)[0])
    del guider, kps
    enc = vx.Wav2Vec2Model(synth.Wav2Vec2Config()).to("cuda")
    enc.load_state_dict(synth.wav2vec2_state_dict())
    proj = vx.AudioProjection(dim=768, depth=4, dim_head=64, heads=12, num_queries=5, embedding_dim=768,
                              output_dim=768, max_seq_len=10).to("cuda")
    proj.load_state_dict(synth.audio_projection_state_dict())
    wav = vx.WaveformProcessor()(torch.randn(int(16000 * F_ / fps), generator=g) * 0.1)["input_values"].cuda()
    res["wav2vec2_ms"], st = timed(lambda: enc(wav).last_hidden_state)
    res["wav2vec2_frames"] = st.shape[1]
    res["audio_windows_projection_ms"], tok = timed(lambda: proj(audio_windows(st, F_, 2)))
    res["audio_tokens"] = list(tok.shape)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
