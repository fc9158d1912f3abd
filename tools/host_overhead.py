"""Host enqueue time vs GPU time of one UNet3D forward (b = 2 CFG pair and b = 1 lone half) at 512^2, f = 16: is the
Python / ctypes launch path ahead of the GPU?  usage: python tools/host_overhead.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import v_express_amd as vx  # noqa: E402
from v_express_amd import ops, synth  # noqa: E402

dev = torch.device("cuda", 0)
cfg = synth.UNetConfig()
unet = vx.UNet3DConditionModel(cfg).to(dev)
refnet = vx.UNet2DConditionModel(cfg).to(dev)
unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=torch.bfloat16, draw_on_device=True))
unet.release_raw_weights()
refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=torch.bfloat16, draw_on_device=True))
refnet.release_raw_weights()
F, h, w = 16, 64, 64
inp = synth.synthetic_inputs(cfg, F, h, w, seed=42, device=dev)
writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                      reference_attention_weight=0.95, audio_attention_weight=3.0)
refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=torch.zeros(1, 1, 768, device=dev), return_dict=False)
reader.update(writer, True)
c0 = cfg.block_out_channels[0]
kps_all = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(2, F, h * w, c0)
aud = inp["audio_embeddings"].to(torch.bfloat16).contiguous()
res = {}
for b, rows in ((2, [0, 1]), (1, [1]), (1, [0]), (3, [0, 1, 0]), (3, [1, 0, 1]), (4, [0, 1, 0, 1])):
    x = ops.ncfhw_to_nhwc(inp["latents"].repeat(b, 1, 1, 1, 1), 8)
    kps = kps_all[rows].reshape(b * F, h * w, c0).contiguous()
    ehs = aud[rows].reshape(-1, 768).contiguous()
    akv = unet.precompute_audio_kv(ehs)
    az = [r == 0 for r in rows]

    def fwd(t):
        return unet.forward_tokens(x, t, ehs, kps, b=b, f=F, H=h, W=w, batch_rows=rows, audio_kv=akv, audio_zero=az)
    for t in (999, 959):
        fwd(t)
    torch.cuda.synchronize()
    host, gpu = [], []
    for t in (919, 879, 839, 799):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        fwd(t)
        e1.record()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1) * 1e-3)
    res[f"b{b}_rows{rows}"] = dict(host_enqueue_ms=1e3 * sum(host) / len(host), gpu_ms=1e3 * sum(gpu) / len(gpu))
print(json.dumps(res, indent=1))
