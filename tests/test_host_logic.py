"""CPU tests of the host-side logic: C-ABI loading, weight re-layouts, window plan, scheduler, unit sharding."""
import ctypes
import os

import pytest
import torch

from oracle import loop as OL
from v_express_amd import context, distributed, synth, weights
from v_express_amd.scheduler import DDIMScheduler

SCHED_KW = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, steps_offset=1,
                prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from v_express_amd import lib
    names = lib.declared_symbols()
    assert len(names) >= 17 and "vx_gemm" in names and "vx_attention" in names
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), n
    assert lib.lib.vx_abi_version() == 15
    # the binary says which sources it was compiled from (stamped by csrc/Makefile) and lib.py has compared that with the
    # sources on disk at import: a stale .so does not get this far
    src, _, defs = lib.lib.vx_build_id().decode().partition("|")
    assert src == lib.source_id() and defs == "" and lib.LIB_SHA256 == src
    assert lib.lib.vx_last_kernel() == b"" or lib.lib.vx_last_kernel().startswith((b"gemm", b"attn", b"ff_", b"tblock", b"temporal"))
    assert ctypes.sizeof(lib.GemmParams) % 8 == 0
    # argument validation happens before any launch, so it works without a GPU and never aborts the process
    p = lib.GemmParams()
    rc = lib.lib.vx_gemm(ctypes.byref(p), None)
    assert rc < 0 and b"vx_gemm" in lib.lib.vx_last_error_string()
    with pytest.raises(lib.VxError):
        lib.check(rc, "vx_gemm")


def test_gemm_params_struct_matches_header_layout():
    """Compile a 3-line C probe against include/vexpress_hip.h and compare sizeof/offsets with the ctypes mirror."""
    import subprocess
    import tempfile
    from v_express_amd import lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "vexpress_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(vx_gemm_params), offsetof(vx_gemm_params, w),
  offsetof(vx_gemm_params, alpha), offsetof(vx_gemm_params, residual), offsetof(vx_gemm_params, part_out),
  offsetof(vx_gemm_params, vt_pitch), offsetof(vx_gemm_params, ring_hint), offsetof(vx_gemm_params, ln_eps),
  offsetof(vx_gemm_params, coop_epoch)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        inc = os.path.join(os.path.dirname(lib.HEADER))
        subprocess.check_call(["gcc", "-I", inc, os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        got = list(map(int, subprocess.check_output([os.path.join(d, "p")]).split()))
    G = lib.GemmParams
    assert got == [ctypes.sizeof(G), G.w.offset, G.alpha.offset, G.residual.offset, G.part_out.offset,
                   G.vt_pitch.offset, G.ring_hint.offset, G.ln_eps.offset, G.coop_epoch.offset]


def test_axattn_params_struct_matches_header_layout():
    """vx_axattn_params (ABI 15): the ctypes mirror against a C probe compiled from include/vexpress_hip.h; and the entry
    point validates its arguments before any launch (no GPU needed)."""
    import subprocess
    import tempfile
    from v_express_amd import lib
    fields = ["ldx", "out", "rows", "rows_per_frame", "ln_stats", "ln_eps", "kq", "vo", "bias_o", "alpha", "row_stats_out",
              "row_stats_parts", "row_stats_eps"]
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "vexpress_hip.h"\nint main(void){ printf("%zu", sizeof(vx_axattn_params));\n' +
           "".join(f'printf(" %zu", offsetof(vx_axattn_params, {f}));\n' for f in fields) + 'printf("\\n"); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.dirname(lib.HEADER), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        got = list(map(int, subprocess.check_output([os.path.join(d, "p")]).split()))
    A = lib.AxAttnParams
    assert got == [ctypes.sizeof(A)] + [getattr(A, f).offset for f in fields]
    assert lib.lib.vx_audio_xattn_supported(320, 8, 5, 4096) == 1 and lib.lib.vx_audio_xattn_supported(1280, 8, 5, 64) == 1
    assert lib.lib.vx_audio_xattn_supported(320, 8, 4, 4096) == 0 and lib.lib.vx_audio_xattn_supported(64, 8, 5, 64) == 0
    assert lib.lib.vx_audio_xattn_packed_bytes(320, 16) == 16 * 320 * 96
    rc = lib.lib.vx_audio_xattn(ctypes.byref(A()), None)
    assert rc < 0 and b"vx_audio_xattn" in lib.lib.vx_last_error_string()


def test_windows_and_alignment_match_reference_context_py():
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "windows.pt"), weights_only=False)
    for (F, cs, co), ref in gold.items():
        got = list(context.uniform(step=0, num_frames=F, context_size=cs, context_stride=1, context_overlap=co,
                                   closed_loop=False))
        assert got == [list(map(int, r)) for r in ref]
    assert context.aligned_video_length(128, 16, 4) == 124      # inference.py:255-264
    assert context.aligned_video_length(930, 24, 4) == 924


@pytest.mark.parametrize("F,cs,co", [(16, 16, 4), (64, 16, 4), (124, 16, 4), (128, 16, 4), (11, 4, 2), (100, 24, 4)])
def test_overlap_plan_replays_reference_bookkeeping(F, cs, co):
    """Plan -> (sum of terms / count, DDIM) must equal the oracle's replay of v_express_pipeline.py:552-572,
    including the duplicated frame of a reflected last window (last write wins)."""
    windows = OL.uniform_windows(F, cs, co)
    plan = context.overlap_plan(windows, F)
    f = len(windows[0])
    g = torch.Generator().manual_seed(F)
    lat = torch.randn(1, 4, F, 2, 2, generator=g)
    outs = [torch.randn(2, 4, f, 2, 2, generator=g) for _ in windows]
    it = iter(outs)
    ddim = OL.DDIM()
    ddim.set_timesteps(25)
    ref = OL.mean_overlap(lambda x, t, e, k: next(it), lat, [479], ddim, windows, 3.5,
                          torch.zeros(2, 1, F, 2, 2), torch.zeros(2, F, 1, 8))
    s = DDIMScheduler(**SCHED_KW)
    s.set_timesteps(25)
    sa, s1a, sap, s1ap = s.step_coefficients(479)
    preds = [o[0:1] + 3.5 * (o[1:2] - o[0:1]) for o in outs]
    got = lat.clone()
    for fr in plan["step_frames"]:
        v = None
        for (wi, li) in plan["terms"][fr]:
            term = preds[wi][:, :, li] / float(plan["counts"][fr])
            v = term if v is None else v + term
        x = lat[:, :, fr]
        x0 = sa * x - s1a * v
        eps = sa * v + s1a * x
        got[:, :, fr] = sap * x0 + s1ap * eps
    assert torch.allclose(got, ref, atol=1e-6, rtol=1e-6)


def test_scheduler_matches_oracle_and_golden():
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "ddim.pt"), weights_only=False)
    s = DDIMScheduler(**SCHED_KW)
    s.set_timesteps(25)
    assert s.timesteps.tolist() == g["timesteps"].tolist()
    assert torch.allclose(s.alphas_cumprod, g["alphas_cumprod"], atol=1e-7)
    d = OL.DDIM()
    d.set_timesteps(25)
    x, v = torch.randn(1, 4, 3, 8, 8), torch.randn(1, 4, 3, 8, 8)
    for t in (999, 519, 39):
        assert torch.allclose(s.step(v, t, x).prev_sample, d.step(v, t, x), atol=1e-6)
    with pytest.raises(NotImplementedError):
        s.step(v, 999, x, eta=0.5)


def test_weight_relayouts():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 16, generator=g)
    wi = weights.geglu_interleave(w)
    assert torch.equal(wi[0:8], w[0:8]) and torch.equal(wi[8:16], w[32:40]) and torch.equal(wi[16:24], w[8:16])
    sd = {"c.weight": torch.randn(6, 4, 3, 3, generator=g), "c.bias": torch.randn(6, generator=g)}
    pc = weights.prep_conv(sd, "c", "cpu")
    assert pc.w.shape == (8, 3 * 3 * 8) and pc.b.shape == (8,) and pc.cout == 6
    w4 = pc.w.float().view(8, 3, 3, 8)
    assert torch.allclose(w4[:6, :, :, :4], sd["c.weight"].permute(0, 2, 3, 1).to(torch.bfloat16).float())
    assert (w4[6:] == 0).all() and (w4[..., 4:] == 0).all() and (pc.b[6:] == 0).all()
    cfg = synth.UNetConfig(block_out_channels=(64, 128, 256, 256))
    sd3 = synth.unet3d_state_dict(cfg)
    a = weights.prep_self_attn(sd3, "down_blocks.0.attentions.0.transformer_blocks.0.attn1", "cpu")
    assert a.wqkv.shape == (192, 64) and a.bqkv is None and a.out.b.dtype == torch.float32
    m = weights.prep_motion(sd3, "mid_block.motion_modules.0", "cpu")
    assert m.attn[0].pe.shape == (32, 256) and torch.allclose(m.attn[1].pe, synth.pe_table(32, 256)[0])


def test_model_surface_and_errors_without_gpu():
    import v_express_amd as vx
    cfgd = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=8, cross_attention_dim=768)
    unet = vx.UNet3DConditionModel.from_config_2d(cfgd, dict(use_motion_module=True, motion_module_kwargs=dict(
        temporal_position_encoding_max_len=32)))
    assert unet.config.cross_attention_dim == 768 and unet.in_channels == 4
    sd = synth.unet3d_state_dict(unet.cfg)
    r = unet.load_state_dict({k: v for k, v in sd.items() if "motion_modules" not in k}, strict=False)
    assert r.missing_keys and all("motion_modules" in k for k in r.missing_keys)
    r = unet.load_state_dict({k: v for k, v in sd.items() if "motion_modules" in k}, strict=False)   # inference.py:90-93
    with pytest.raises(RuntimeError):
        unet.load_state_dict({"nope": torch.zeros(1)}, strict=True)
    with pytest.raises(RuntimeError, match="MI355X"):
        unet._prepared()                                   # still on the CPU: there is no CPU path
    with pytest.raises(NotImplementedError):
        vx.UNet3DConditionModel.from_config_2d(cfgd, dict(use_motion_module=False))


def test_unit_partition_properties():
    for W in (1, 2, 5, 10, 11):
        for R in (1, 2, 4, 8):
            parts = distributed.partition_units(W, R)
            flat = sorted(u for p in parts for u in p)      # (order across ranks: contiguous, or balanced by cost)
            assert flat == [(w, h) for w in range(W) for h in range(2)]
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
            sch = distributed.UnitSchedule(W, R)
            assert sch.rounds() == -(-2 * W // R)
            for r in range(R):
                for w, halves in sch.calls(r):
                    assert halves in ([0], [1], [0, 1])
    assert distributed.UnitSchedule(10, 8).rounds() == 3          # 124 frames on 8 GPUs: 20 units -> 3 rounds
    assert distributed.split_frames(124, 8)[0] == (0, 16) and distributed.split_frames(124, 8)[-1] == (112, 124)


def test_audio_windows_match_reference_golden():
    """prepare_audio_embeddings' window construction (pipelines/v_express_pipeline.py:381-401) vs the reference's own
    output stored in tests/golden/prologue.pt."""
    import os
    import cases
    from v_express_amd.prologue import audio_windows
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prologue.pt"), weights_only=False)
    assert torch.equal(audio_windows(cases.prologue_inputs()["wav2vec_states"], 7, 2), g["audio_windows_F7"])


def test_checkpoint_ingestion_order_and_legacy_key_remaps(tmp_path):
    """checkpoints.py: (1) the legacy remaps equal the reference's own train.py:122-161 function (executed from the
    reference source when it is available, otherwise against the hand-written expectation); (2) the two-file load
    order of inference.py:84-96 (denoising UNet, then motion module on top, both strict=False) fills every key; (3)
    file formats (.bin through torch.load, .safetensors) and the VAE's deprecated attention key names."""
    import ast
    import copy
    from v_express_amd import checkpoints as CK
    sd = {"down.attentions.0.attn1.to_q.weight": torch.ones(2), "down.norm1.weight": torch.full((2,), 2.0),
          "down.attentions.0.attn2.to_q.weight": torch.zeros(2),
          "down.attentions.0.attn2.processor.to_q_aud.weight": torch.full((2,), 5.0),
          "down.attentions.0.attn2.to_out.0.bias": torch.zeros(2),
          "down.attentions.0.attn2.processor.to_out_aud.0.bias": torch.full((2,), 7.0)}
    ref_path = "/root/reference/train.py"
    ref_fn = None
    if os.path.exists(ref_path):
        tree = ast.parse(open(ref_path).read())
        node = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_denoising_unet_state_dict"][0]
        ns = {"copy": copy}
        exec(compile(ast.Module(body=[node], type_ignores=[]), ref_path, "exec"), ns)
        ref_fn = ns["get_denoising_unet_state_dict"]
    for kind in ("old_attn", "moore_pretrained", "new_attn"):
        got = CK.get_denoising_unet_state_dict(sd, kind)
        if ref_fn is not None:
            want = ref_fn(sd, kind)
            assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want), kind
    got = CK.get_denoising_unet_state_dict(sd, "old_attn")
    assert torch.equal(got["down.attentions.0.attn1_5.to_q.weight"], torch.ones(2))
    assert torch.equal(got["down.norm1_5.weight"], torch.full((2,), 2.0))
    assert torch.equal(got["down.attentions.0.attn2.to_q.weight"], torch.full((2,), 5.0))
    assert torch.equal(got["down.attentions.0.attn2.to_out.0.bias"], torch.full((2,), 7.0))
    assert "down.attentions.0.attn1_5.to_q.weight" not in CK.get_denoising_unet_state_dict(sd, "new_attn")
    with pytest.raises(ValueError):
        CK.get_denoising_unet_state_dict(sd, "nope")
    # two-file load order: spatial weights first, motion-module weights second, both partial
    from v_express_amd import UNet3DConditionModel
    cfg = synth.UNetConfig(block_out_channels=(64, 128, 256, 256))
    full = synth.unet3d_state_dict(cfg)
    motion = {k: v for k, v in full.items() if "motion_modules" in k}
    spatial = {k: v for k, v in full.items() if "motion_modules" not in k}
    assert motion and spatial
    torch.save(spatial, tmp_path / "denoising_unet.bin")
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in motion.items()}, str(tmp_path / "motion_module.safetensors"))
    unet = UNet3DConditionModel(cfg)
    r1 = unet.load_state_dict(CK._load_file(str(tmp_path / "denoising_unet.bin")), strict=False)
    assert set(r1.missing_keys) == set(motion)
    r2 = unet.load_state_dict(CK._load_file(str(tmp_path / "motion_module.safetensors")), strict=False)
    assert set(r2.missing_keys) == set(spatial) and not r2.unexpected_keys
    assert set(unet._raw) == set(full) and all(torch.equal(unet._raw[k], full[k]) for k in full)
    assert CK.convert_vae_attention_key("decoder.mid_block.attentions.0.proj_attn.weight") == \
        "decoder.mid_block.attentions.0.to_out.0.weight"
    assert CK.convert_vae_attention_key("encoder.mid_block.attentions.0.query.bias") == \
        "encoder.mid_block.attentions.0.to_q.bias"
    assert CK.convert_vae_attention_key("decoder.conv_in.weight") == "decoder.conv_in.weight"


@pytest.mark.parametrize("tag", ["small", "base"])
def test_wav2vec2_host_composition_with_emulated_kernels(tag, monkeypatch):
    """The host side of v_express_amd.Wav2Vec2Model - weight re-layouts (tap-major conv matrices, weight-norm, per-group
    positional-conv matrices, fused QKV), the overlapping-row window views that turn conv1d into plain GEMMs, the
    group-major zero-padded positional-conv buffer, call order - run here with tests/fake_ops.py standing in for the
    HIP wrappers (fp32 math, bf16 rounding at each kernel boundary) and compared with the fp32 oracle.  The GPU suite
    runs the same code on the real kernels (tests/test_gpu_prologue.py)."""
    import cases
    import fake_ops
    from oracle import wav2vec2 as OW
    from v_express_amd import ops
    from v_express_amd.wav2vec2 import Wav2Vec2Model, WaveformProcessor
    fake_ops.install(monkeypatch, ops)
    monkeypatch.setattr(Wav2Vec2Model, "_need_gpu", lambda self: None)
    kw, samples = cases.W2V_CASES[tag]
    cfg = synth.Wav2Vec2Config(**kw)
    sd = synth.wav2vec2_state_dict(cfg)
    m = Wav2Vec2Model(cfg).to("cpu")
    m.load_state_dict({("wav2vec2." + k): v for k, v in sd.items()} | {"lm_head.weight": torch.zeros(2, 2)})  # CTC ckpt
    wav = cases.waveform(samples)
    got = m(wav).last_hidden_state
    want = OW.forward(sd, wav, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_conv_pos_embedding_groups,
                      cfg.conv_stride, cfg.layer_norm_eps)
    assert got.shape == want.shape == (1, cfg.num_frames(samples), cfg.hidden_size)
    assert ((got - want).norm() / want.norm()).item() < 2e-2
    feats = m.extract_features(wav[0]).float()
    fw = OW.feature_encoder(sd, wav, cfg.conv_stride)[0]
    assert ((feats - fw).norm() / fw.norm()).item() < 1.5e-2
    # the processor stand-in == Wav2Vec2FeatureExtractor(do_normalize=True)
    raw = torch.randn(samples, generator=torch.Generator().manual_seed(5)) * 0.3 + 0.1
    pv = WaveformProcessor()(raw, return_tensors="pt", sampling_rate=16000)["input_values"]
    assert pv.shape == (1, samples) and torch.allclose(pv, OW.normalize_waveform(raw)[None], atol=1e-6)
    with pytest.raises(ValueError):
        WaveformProcessor()(raw, sampling_rate=8000)
    with pytest.raises(NotImplementedError):
        Wav2Vec2Model(synth.Wav2Vec2Config(do_stable_layer_norm=True))


def test_bench_algorithmic_flops_match_the_survey_figures():
    """bench.py's whole-path denominator: SURVEY.md 8d gives 66.4 / 82.3 / 84.9 TFLOP per decoded frame for the 16-, 64-
    and 124-frame clips at 512x512 and 183.8 at 768x768 (the SDPA terms grow with the square of the pixel count)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vx_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert abs(bench.flop_per_frame(16, 1, 25, 1.0) - 66.4) < 0.05
    assert abs(bench.flop_per_frame(64, 5, 25, 1.0) - 82.3) < 0.05
    assert abs(bench.flop_per_frame(124, 10, 25, 1.0) - 84.9) < 0.05
    assert abs(bench.flop_per_frame(16, 1, 25, 2.25) - 183.8) < 0.1


def test_bench_prints_one_compact_line_the_driver_can_parse(tmp_path):
    """BENCH_r05.json was `parsed: null` because the printed line had grown to 23 KB (ranking, per-kernel,
    per-instantiation and block-path tables).  The line is now bench.compact_line(result): the contract's keys + `roofline`
    (the dominant kernel only) + `cpu_baseline`, under bench.LINE_LIMIT bytes whatever the tables hold; the tables go to the
    side file bench.write_detail writes.  Checked on the committed 23 KB round-5 result and on a synthetic 8-GPU result."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r05x_bench_driver_flags.json")) as f:
        full = json.loads(f.read())
    assert len(json.dumps(full)) > 20000                      # the line that broke the driver's parser
    # worst case on top: hundreds of kernels / block paths, a very long free-text sample
    full["roofline"]["ranking"] = full["roofline"]["ranking"] * 20
    full["block_paths"] = {f"block{i}": "x" * 200 for i in range(300)}
    full["cpu_baseline"]["sample"] = "s" * 5000
    detail = bench.write_detail(full, str(tmp_path / "sub" / "bench_detail.json"))
    line = bench.compact_line(full, detail)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 4096, len(text)
    back = json.loads(text)
    assert back == line and "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail", "lib_sha256"):
        assert key in back, key
    assert back["config"]["workload"] and "model" not in back["config"]
    rf = back["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "whole_path", "rocprof",
                "algorithmic_flop_per_launch", "avg_launch_us", "launches", "share_of_clip_kernel_time"):
        assert key in rf, key
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and "ranking" not in rf and "per_kernel" not in rf
    assert rf["rocprof"] is None or set(rf["rocprof"]) == {"avg_launch_us", "frac", "source"}
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert abs(back["value"] - full["value"]) < 1e-4 * full["value"]
    with open(tmp_path / "sub" / "bench_detail.json") as f:
        assert json.load(f)["roofline"]["ranking"] == full["roofline"]["ranking"]       # nothing is lost, only moved
    # a multi-GPU result: per-rank lists and collective tables stay small too
    multi = dict(full, n_gpus=8, scaling="strong",
                 per_rank={"ms_per_step_wall": [1.0] * 8, "ms_per_step_min": 1.0, "ms_per_step_max": 2.0, "compute_ms_min": 1.0,
                           "compute_ms_max": 2.0, "instrumented_clip_gpu_ms": [1.0] * 8, "note": "n" * 500},
                 collectives_rank0={f"all_gather_{i}": {"calls": 25, "ms": 1.0, "mb": 1.0, "share_of_clip": 0.01} for i in range(4)},
                 collective_backend="nccl", same_clip_1gpu_fps=10.0, speedup_vs_1gpu_same_clip=6.0)
    multi.pop("cpu_baseline")
    t2 = json.dumps(bench.compact_line(multi, None))
    assert len(t2) < bench.LINE_LIMIT and json.loads(t2)["per_rank"]["ms_per_step_max"] == 2.0


def test_cpu_leg_building_blocks():
    """bench.py's CPU leg (N pinned worker processes): the core list it pins to is one CPU per physical
    core inside the affinity mask; the oracle's SDPA form (what AttnProcessor2_0 calls; used by the timed leg only) equals the
    explicit softmax form the parity tests run; the timing-only weight pool gives every tensor its own memory and the
    schema of the seeded draw."""
    import bench
    from oracle import leaf
    cores, logical = bench._core_cpus()
    allowed = os.sched_getaffinity(0)
    assert cores and set(cores) <= allowed and len(set(cores)) == len(cores) and logical == len(allowed)
    g = torch.Generator().manual_seed(0)
    w = {"a.to_q.weight": torch.randn(64, 64, generator=g) * 0.1, "a.to_k.weight": torch.randn(64, 48, generator=g) * 0.1,
         "a.to_v.weight": torch.randn(64, 48, generator=g) * 0.1, "a.to_out.0.weight": torch.randn(64, 64, generator=g) * 0.1,
         "a.to_out.0.bias": torch.randn(64, generator=g)}
    x, ctx = torch.randn(3, 50, 64, generator=g), torch.randn(3, 7, 48, generator=g)
    ref = leaf.attention(w, "a", x, ctx, 8)
    leaf.USE_SDPA[0] = True
    try:
        got = leaf.attention(w, "a", x, ctx, 8)
    finally:
        leaf.USE_SDPA[0] = False
    assert torch.allclose(got, ref, atol=2e-6, rtol=1e-5)
    cfg = synth.VaeConfig()
    fast, seeded = synth.vae_decoder_state_dict(cfg, timing_only=True), synth.vae_decoder_state_dict(cfg)
    assert {k: v.shape for k, v in fast.items()} == {k: v.shape for k, v in seeded.items()}
    big = [v for v in fast.values() if v.numel() > 1000]
    assert len({v.data_ptr() for v in big}) == len(big) and all(torch.isfinite(v).all() for v in big)
    k = "decoder.conv_in.weight"
    assert abs(fast[k].std().item() / seeded[k].std().item() - 1) < 0.1


def test_bench_picks_the_dominant_kernel_over_all_kernels_by_clip_weighted_time():
    """`roofline.kernel` was chosen among vx_gemm launches only, so the line named an 8 % GEMM instantiation
    while the 12 % attention kernel had no roofline anywhere.  bench.rank_kernels sees every profiled launch - GEMM,
    attention, the one-launch blocks, HBM-bound kernels - and weights each by how often it runs per clip."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vx_bench3", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    gemm, attn, gn, vae = "gemm_kernel<128, 160, 2, 2, 2, 0, true, false, false, false>", \
        "attn3_kernel<2, false, true, false>", "groupnorm_apply", "gemm_kernel<256, 256, 4, 2, 2, 0, true, false, false, false>"
    launches = [dict(symbol=gemm, seconds=60e-6, weight=25.0, flops=44.2e9, bytes=60e6) for _ in range(77)]       # 115 ms per clip
    launches += [dict(symbol=attn, seconds=600e-6, weight=25.0, flops=0.515e12, bytes=200e6) for _ in range(10)]  # 150 ms per clip
    launches += [dict(symbol=gn, seconds=23e-6, weight=25.0, flops=0.0, bytes=92e6) for _ in range(77)]           # 44 ms per clip
    launches += [dict(symbol=vae, seconds=9000e-6, weight=4.0, flops=9.0e12, bytes=1e9)]                           # 36 ms per clip,
    #                                                                       the most raw time of the instrumented leg
    ranking, allk, clip_s, total = bench.rank_kernels(launches)
    assert [r["kernel"] for r in ranking] == [attn, gemm, gn, vae]
    top = ranking[0]
    assert top["bound"] == "mfma" and abs(top["achieved"] - 0.515e12 / 600e-6 / 1e12) < 1e-6
    assert abs(top["frac"] - top["achieved"] / bench.PEAK_BF16_TFLOPS) < 1e-12 and top["launches"] == 10
    assert abs(top["share_of_clip_kernel_time"] - 0.150 / total) < 1e-9 and abs(sum(clip_s.values()) - total) < 1e-12
    hb = ranking[2]
    assert hb["bound"] == "hbm" and hb["unit"] == "GB/s" and abs(hb["achieved"] - 92e6 / 23e-6 / 1e9) < 1e-6
    assert abs(hb["frac"] - hb["achieved"] / bench.HBM_PEAK_GBS) < 1e-12
    assert allk[gemm]["launches"] == 77 and abs(allk[gemm]["flops"] - 77 * 44.2e9) < 1.0


def test_bench_quotes_committed_profiles_only_for_the_loaded_kernel_build(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` / `roofline.rocprof` come from files under profiles/ - but only from files that carry
    the identity of the kernel sources the loaded library was built from (`lib_sha256`); anything else is refused with a
    reason instead of being quoted next to a live number (the round-3 line mixed two commits)."""
    import importlib.util
    import json
    from v_express_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vx_bench2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sym = "gemm_ring_kernel<0, true, false, false, false, false>"
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kernels = {sym: dict(launches=7, fetch_bytes_per_launch=2.0e8, write_bytes_per_launch=0.5e8)}
    (prof / "r98_pmc_traffic.json").write_text(json.dumps(dict(lib_sha256="0" * 16, kernels=kernels)))
    (prof / "r98_trace_summary.txt").write_text("# lib_sha256=0000000000000000 command=x\n"
                                                f"   10.00 ms  5.0% n=   100 avg=    99.0 us  void {sym}(vx_gemm_params)\n")
    traffic, why = bench._pmc_traffic(sym)
    assert traffic is None and "no counter file" in why and "r98" in why
    assert bench._rocprof_launch_avg(sym) is None
    (prof / "r99_pmc_traffic.json").write_text(json.dumps(dict(lib_sha256=lib.LIB_SHA256, kernels=kernels)))
    (prof / "r99_trace_summary.txt").write_text(f"# lib_sha256={lib.LIB_SHA256} command=x\n"
                                                "total kernel time 1.0 ms over 3 launches\n"
                                                f"   10.00 ms  5.0% n=   100 avg=    80.5 us  void {sym}(vx_gemm_params)\n"
                                                f"    1.00 ms  0.5% n=    10 avg=    11.0 us  void {sym[:-6]}true>(vx_gemm_params)\n")
    traffic, src = bench._pmc_traffic(sym)
    assert traffic == 2.5e8 and src == os.path.join("profiles", "r99_pmc_traffic.json")
    rp = bench._rocprof_launch_avg(sym)
    assert rp == dict(avg_launch_us=80.5, launches=100, file=os.path.join("profiles", "r99_trace_summary.txt"))
    assert bench._pmc_traffic("gemm_ring_kernel<1, false, false, false, true, false>")[0] is None
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "lib_id.py")], capture_output=True, text=True)
    assert out.stdout.strip() == lib.LIB_SHA256          # the scripts stamp profiles with the value bench.py checks
    # non-template kernels are printed without "void" by rocprofv3; the fused blocks must be found as well
    (prof / "r99_trace_summary.txt").write_text(f"# lib_sha256={lib.LIB_SHA256} command=x\n"
                                                "    93.97 ms   3.6% n=   510 avg=    184.3 us  tblock_kernel(vx_tblock_params, float)\n"
                                                "   312.23 ms  12.0% n=   520 avg=    600.4 us  void attn3_kernel<2, false, true, false>(Attn3Params)\n")
    assert bench._rocprof_launch_avg("tblock_kernel")["avg_launch_us"] == 184.3
    assert bench._rocprof_launch_avg("attn3_kernel<2, false, true, false>")["launches"] == 520


def test_a_stale_or_variant_library_is_never_paired_with_committed_profiles(monkeypatch):
    """the identity bench.py keys profiles by is the identity of the loaded BINARY (vx_build_id), not of the
    sources on disk: an unstamped variant build, a build with extra -D flags and a stale build all get an identity that
    no committed file carries; a stale build is an ImportError unless VX_ALLOW_STALE_LIB=1."""
    import types
    import warnings
    from v_express_amd import lib
    real = lib.lib

    def with_id(s):
        monkeypatch.setattr(lib, "lib", types.SimpleNamespace(vx_build_id=lambda: s.encode()))
        try:
            return lib._build_identity()
        finally:
            monkeypatch.setattr(lib, "lib", real)
    sid = lib.source_id()
    assert with_id(f"{sid}|") == sid
    assert with_id(f"{sid}|-DVX_GELU_PK") == f"{sid}+-DVX_GELU_PK"
    assert with_id("unstamped|").startswith("unstamped:")
    monkeypatch.delenv("VX_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(ImportError, match="rebuild"):
        with_id("0123456789abcdef|")
    monkeypatch.setenv("VX_ALLOW_STALE_LIB", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert with_id("0123456789abcdef|") == "stale:0123456789abcdef"
    assert w and "rebuild" in str(w[0].message)


def test_audio_encoder_directory_loader(tmp_path, monkeypatch):
    """checkpoints.load_audio_encoder / Wav2Vec2Model.from_pretrained on a transformers-style directory (config.json,
    preprocessor_config.json, model.safetensors of a Wav2Vec2ForCTC checkpoint: `wav2vec2.` prefix + CTC head), then
    one emulated forward against the oracle."""
    import json
    import cases
    import fake_ops
    from safetensors.torch import save_file
    from oracle import wav2vec2 as OW
    from v_express_amd import checkpoints, ops
    from v_express_amd.wav2vec2 import Wav2Vec2Model
    fake_ops.install(monkeypatch, ops)
    monkeypatch.setattr(Wav2Vec2Model, "_need_gpu", lambda self: None)
    kw, samples = cases.W2V_CASES["small"]
    cfg = synth.Wav2Vec2Config(**kw)
    sd = synth.wav2vec2_state_dict(cfg)
    d = tmp_path / "wav2vec2-tiny"
    d.mkdir()
    conf = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.__dict__.items()}
    conf.update(architectures=["Wav2Vec2ForCTC"], vocab_size=32, model_type="wav2vec2")      # extra keys are ignored
    (d / "config.json").write_text(json.dumps(conf))
    (d / "preprocessor_config.json").write_text(json.dumps({"do_normalize": True, "sampling_rate": 16000}))
    ckpt = {("wav2vec2." + k): v.contiguous() for k, v in sd.items()}
    ckpt["lm_head.weight"], ckpt["lm_head.bias"] = torch.zeros(32, cfg.hidden_size), torch.zeros(32)
    save_file(ckpt, str(d / "model.safetensors"))
    enc, proc = checkpoints.load_audio_encoder(str(d), device="cpu")
    assert enc.cfg == cfg and proc.sampling_rate == 16000 and proc.do_normalize
    raw = torch.randn(samples, generator=torch.Generator().manual_seed(1)) * 0.1
    wav = proc(raw, return_tensors="pt", sampling_rate=16000)["input_values"]
    got = enc(wav).last_hidden_state
    want = OW.forward(sd, wav, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_conv_pos_embedding_groups,
                      cfg.conv_stride, cfg.layer_norm_eps)
    assert ((got - want).norm() / want.norm()).item() < 2e-2
    with pytest.raises(FileNotFoundError):
        (d / "model.safetensors").unlink()
        checkpoints.load_audio_encoder(str(d), device="cpu")


def test_ring_dma_schedule_is_race_free():
    """tools/ring_schedule_check.py replays gemm_ring_kernel's LDS-DMA issue / vmcnt / barrier schedule for both wave
    rows and proves read-after-write and write-after-read ordering for every block length; the vmcnt immediates are
    tight, so bumping any of them by one must be caught."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ring_schedule_check.py")
    spec = importlib.util.spec_from_file_location("ring_schedule_check", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for nk in (1, 2, 3, 5, 9, 10, 45, 90):
        for tiles in (1, 2, 3, 4):
            assert m.check(nk * tiles, nk) is None, (nk, tiles)
            for variant in (1, 2, 3):   # the experimental issue placements (VX_RING_MISSUE)
                assert m.check(nk * tiles, nk, variant) is None, (nk, tiles, variant)
    orig = m.program
    for variant, bumps in ((0, ((13, 14), (11, 12), (15, 16), (9, 10), (3, 4))), (1, ((14, 15), (10, 11))),
                           (2, ((12, 13), (14, 15), (10, 11))), (3, ((9, 10), (11, 12)))):
        for old, new in bumps:
            m.program = lambda g, S, nk, mi=0, o=old, n=new: [("wait", n) if e == ("wait", o) else e
                                                              for e in orig(g, S, nk, mi)]
            assert m.check(10, 5, variant) is not None, f"variant {variant}: vmcnt({old}) -> vmcnt({new}) went unnoticed"
    m.program = orig
    # the immediates of the kernel source are the ones the checker replays
    src = open(os.path.join(os.path.dirname(path), "..", "v-express_amd", "csrc", "vx_gemm_ring.hip")).read()
    import re
    in_kernel = sorted({int(v) for v in re.findall(r"RING_WAIT_VM\((\d+)\)", src)} |
                       {int(v) for v in re.findall(r"ring_wait_vm<(\d+)>\(\)", src)})
    in_checker = sorted({e[1] for g in (0, 1) for S, nk in ((1, 1), (2, 1), (10, 5)) for mi in (0, 1, 2, 3)
                         for e in orig(g, S, nk, mi) if e[0] == "wait"})
    assert in_kernel == in_checker, (in_kernel, in_checker)


def test_bench_refuses_a_world_size_that_differs_from_gpus(tmp_path):
    """bench.py: `--gpus N` without a torchrun environment re-launches N ranks - and says so loudly when the box has
    fewer GPUs; inside a torchrun environment a WORLD_SIZE that differs from --gpus is an error, never a silent
    single-rank run that prints n_gpus = 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VX_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only 0 GPU(s) are visible" in (r.stdout + r.stderr)
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], cwd=root, env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr), r.stdout[-500:] + r.stderr[-500:]


def test_float16_is_a_compute_dtype_and_device_spellings_compare_equal():
    """Reference inference.py:44,150-151: `--dtype fp16` is the default and every model gets `.to(dtype=dtype, device=device)`.
    Round 6: float16 is a COMPUTE dtype - a float16 model runs the IEEE-half build of the kernel library
    (libvexpress_hip_f16.so: same sources, -DVX_ELEM_F16, same ABI), a bfloat16 / float32 model the bfloat16 one; every model
    entry point is wrapped to run under its element type; changing the element type after the device layouts were built
    rebuilds them from the source tensors (or fails loudly when those were released); 'cuda' and 'cuda:<current>' are the
    same device for `.to()`."""
    import ctypes
    import v_express_amd as vx
    from v_express_amd import lib as L, module_base as MB, ops
    cfgd = dict(block_out_channels=[64, 128, 256, 256], attention_head_dim=8, cross_attention_dim=768)
    unet = vx.UNet3DConditionModel.from_config_2d(cfgd, dict(use_motion_module=True))
    assert unet.dtype == torch.bfloat16 and unet._elem == torch.bfloat16
    unet.to(dtype=torch.float16, device="cpu")
    assert unet.dtype == torch.float16 and unet._elem == torch.float16
    assert unet.half()._elem == torch.float16 and unet.to(torch.float32)._elem == torch.bfloat16      # fp32: I/O dtype only
    assert vx.UNet2DConditionModel.from_config(cfgd).to(torch.float16)._elem == torch.float16
    with pytest.raises(TypeError):
        unet.to(torch.float64)
    # both libraries load, export every declared symbol, agree on the ABI and on the sources they were built from
    l16 = L.lib_f16()
    assert l16.vx_element_type() == b"f16" and L.lib.vx_element_type() == b"bf16"
    assert l16.vx_abi_version() == L.lib.vx_abi_version() == 15 and l16.vx_build_id() == L.lib.vx_build_id()
    assert all(hasattr(l16, sym) for sym in L.declared_symbols())
    # the element type in force selects library and allocation dtype, and nests
    assert L.current() is L.lib and ops.BF16 is torch.bfloat16
    with L.element_type(torch.float16):
        assert L.current() is l16 and ops.BF16 is torch.float16 and ops._lib.vx_element_type() == b"f16"
        with L.element_type(torch.bfloat16):
            assert L.current() is L.lib
        assert L.current() is l16
    assert L.current() is L.lib
    with pytest.raises(TypeError):
        L.element_type(torch.float32)
    # model entry points run under the model's element type (wrapped once per class)
    seen = []

    class Probe(MB.DeviceModule):
        def forward(self):
            seen.append(L.ELEM[0])
            return self._prepared()

        def _prepared(self):
            seen.append(L.ELEM[0])
    pr = Probe().to(torch.float16)
    pr.forward()
    assert seen == [torch.float16, torch.float16] and L.ELEM[0] is torch.bfloat16
    pr._P = object()
    pr.to(torch.bfloat16)
    assert pr._P is None                                  # element type changed: layouts rebuilt at the next call
    pr._released = True
    with pytest.raises(RuntimeError):
        pr.to(torch.float16)                              # ... which needs the source tensors
    assert MB._norm_device("cuda") == MB._norm_device("cuda:0") == MB._norm_device(0) == torch.device("cuda", 0)
    unet = vx.UNet3DConditionModel.from_config_2d(cfgd, dict(use_motion_module=True))
    unet.to("cuda")
    unet._released = True                                 # after release_raw_weights() a real move must fail ...
    unet.to("cuda:0")                                     # ... the other spelling of the same GPU must not
    with pytest.raises(RuntimeError):
        unet.to("cpu")
    del ctypes


def test_gelu_tail_polynomial_of_the_kernels_matches_erf_gelu():
    """vx_common.h::gelu_f (round 3): erf GELU as relu(x) - |x| 2^q(|x|) with q a degree-6 polynomial for the base-2
    logarithm of the Gaussian tail.  The coefficients are parsed from the kernel header and evaluated here in float32
    (Horner with fused multiply-adds emulated through float64) against x Phi(x) in float64: absolute error <= 1e-6 over
    the whole line, relative error <= 1e-3 where |gelu| >= 1e-3 (a bf16 output rounds at 4e-3)."""
    import re
    import numpy as np
    from scipy.special import erfc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "v-express_amd", "csrc", "vx_common.h")).read()
    body = src[src.index("float gelu_f(float x) {"):src.index("// value * gelu(gate) for four (value, gate) pairs")]
    lead = re.search(r"float q = fmaf\(([-0-9.e+]+)f, u, ([-0-9.e+]+)f\);", body)
    rest = re.findall(r"q = fmaf\(q, u, ([-0-9.e+]+)f\);", body)
    coef = [np.float32(lead.group(1)), np.float32(lead.group(2))] + [np.float32(c) for c in rest]
    assert len(coef) == 7 and "fmed3f(fabsf(x), 0.0f, 8.0f)" in body
    # the packed-FMA A/B form (VX_GELU_PK) carries the same seven coefficients, in Horner order, in its register pairs
    pk = src[src.index("struct GeluPk {"):src.index("__device__ __forceinline__ void mul2(")]
    pairs = re.findall(r"[ABCD] = vx_f2\{([-0-9.e+]+)f, ([-0-9.e+]+)f\};", pk)
    flat = [np.float32(v) for pr in pairs for v in pr]
    assert len(pairs) == 4 and flat[:7] == coef
    x = np.concatenate([np.linspace(-12, 12, 200001), np.random.default_rng(0).standard_normal(100000) * 3,
                        np.array([0.0, -0.0, 1e-8, -1e-8, 50.0, -50.0, 3e4, -3e4])]).astype(np.float32)
    u = np.minimum(np.abs(x), np.float32(8.0))
    q = np.full_like(u, coef[0])
    for c in coef[1:]:
        q = (q.astype(np.float64) * u + c).astype(np.float32)            # one rounding per FMA
    got = (np.maximum(x, 0).astype(np.float64) - u.astype(np.float64) * np.exp2(q.astype(np.float64))).astype(np.float32)
    ref = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2.0))
    err = np.abs(got - ref)
    fin = np.abs(ref) < 1e3                                              # beyond: float32 ulp of the result itself
    assert err[fin].max() <= 1e-6, err[fin].max()
    big = (np.abs(ref) >= 1e-3) & fin
    assert (err[big] / np.abs(ref[big])).max() <= 1e-3
    far = np.abs(x) > 9                                                   # beyond the clamp: relu - 8 T(8) = relu - 5e-15
    assert np.all(np.isfinite(got)) and np.abs(got[far] - np.maximum(x[far], 0)).max() <= 1e-13


def test_kernel_routing_decisions_do_not_depend_on_the_batch():
    """The host-side routing rules of round 3 (`ops.gn_fold_applies`: GroupNorm folded into proj_in; `ops.qk_on_ring`:
    Q | K on the ring kernel) decide from ONE batch item's rows, like `_ring_hint`: a CFG half computed alone on another
    GPU must take the same kernels as its rows inside the batched call (bit-identity of the sharded loop).  Also pins
    where they apply: the 64x64 / 96x96 levels for the fold (per-frame weight copies smaller than the tensor), the
    64x64 ... 16x16 levels for Q | K."""
    from v_express_amd import ops
    f = 16
    for hw, c in ((4096, 320), (1024, 640), (256, 1280), (64, 1280), (9216, 320), (2304, 640)):
        got = []
        for b in (1, 2, 4):
            with ops.frame_rows(hw, items=b):
                got.append((ops.gn_fold_applies(b * f * hw, hw, c, c), ops.qk_on_ring(b * f * hw, c)))
        assert got[0] == got[1] == got[2], (hw, c, got)
    with ops.frame_rows(4096, items=2):
        assert ops.gn_fold_applies(2 * f * 4096, 4096, 320, 320) and ops.qk_on_ring(2 * f * 4096, 320)
    with ops.frame_rows(1024, items=2):
        assert not ops.gn_fold_applies(2 * f * 1024, 1024, 640, 640) and ops.qk_on_ring(2 * f * 1024, 640)
    with ops.frame_rows(2304, items=2):                                       # 768x768: 48x48 level, C = 640
        assert ops.gn_fold_applies(2 * f * 2304, 2304, 640, 640)
    with ops.frame_rows(64, items=2):
        assert not ops.gn_fold_applies(2 * f * 64, 64, 1280, 1280) and not ops.qk_on_ring(2 * f * 64, 1280)
    assert not ops.gn_fold_applies(2 * f * 4096, 4096, 320, 320)              # outside a frame_rows context: never
    with ops.frame_rows(4096, items=2):                                       # a frame-sharded half window (f = 8)
        assert ops.gn_fold_applies(2 * 8 * 4096, 4096, 320, 320)


def test_tblock_lane_level_emulation_matches_plain_math():
    """tools/tblock_emulate.py: the fused temporal-attention kernel's pack layouts, fragment addresses, MFMA operand roles
    and LDS hand-over restated lane by lane in numpy (same index formulas as csrc/vx_tblock.hip) against a float64
    statement of the block with the same rounding points.  The GPU test compares vx_tblock_pack with this pack()."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tblock_emulate as E
    assert E.self_check(seed=1) < 2e-3
    assert E.self_check(seed=2, F=24) < 2e-3          # the reference's default window: two blocks per pixel, masked padding
    # every column of q | k | v appears exactly once in the packed order, 8 pad rows per head
    cols = [E.src_col(h, blk, r) for h in range(8) for blk in range(8) for r in range(16)]
    assert sorted(c for c in cols if c >= 0) == list(range(960)) and cols.count(-1) == 64


def test_tblock_weight_stream_protocol_happens_before():
    """tools/tblock_schedule_check.py: the counted vmcnt waits, the per-chunk barrier and the three-slot ring of
    csrc/vx_tblock.hip restated as a per-wave sequence of vector-memory operations (constants read from the kernel source):
    every chunk read sits behind a wait that guarantees the reader's own copies of it (loads retire in order: a load is
    complete iff at least N loads are younger at a vmcnt(N)), no slot is refilled before the barrier that ends the reads of
    its previous content, the O^T / statistics parking areas are never written under a possible reader."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tblock_schedule_check as S
    K, reads, reuses = S.main()
    assert K["NW"] == 8 and reads == 3 * 44 and reuses == 3 * 44
    # the check is sharp: a wait that leaves the x loads of the next tile out of the count one iteration too long is caught
    bad = dict(K)
    ev = S.wave_program(K, 2)
    k = [i for i, e in enumerate(ev) if e == ("B", (0, 35))][0] - 1
    assert ev[k] == ("W", K["CPW"])
    orig = S.wave_program
    try:
        S.wave_program = lambda KK, tiles: [("W", K["CPW"] + K["NX"]) if i == k else e for i, e in enumerate(orig(KK, tiles))]
        with pytest.raises(AssertionError):
            S.check_waits(bad, 2)
    finally:
        S.wave_program = orig


def test_conv3_emulation_and_schedule():
    """tools/conv3/vx_conv3.hip (round 5's one-pass GroupNorm + SiLU + 3x3 convolution: correct, 0.8 % slower, kept as a
    tool outside the shipped ABI since round 6) before any GPU run: the lane-level emulation of its address arithmetic (plane copies and their slot
    swizzle, in-place normalisation, tap offsets, weight permutation, K order over the two plane buffers, accumulator ->
    output mapping) reproduces a float64 convolution of the normalised zero-padded input exactly, and the happens-before
    replay of its copy / wait / barrier protocol finds no violation with the immediates written in the kernel - and does
    find one when any of them is raised by one."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(root, "tools", "conv3", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    emu = load("conv3_emulate")
    for W, H, frames, c1, c2, n in ((32, 8, 2, 64, 0, 320), (32, 16, 1, 32, 96, 640), (64, 8, 1, 64, 0, 320)):
        assert emu.check(W, H, frames, c1, c2, n) < 1e-9, (W, H, frames, c1, c2, n)
    # every fragment read is conflict-free under the real (non-contiguous) lane groups of ds_read_b128
    assert emu.fragment_read_conflicts(64) == 0 and emu.fragment_read_conflicts(32) == 0
    assert emu.fragment_read_conflicts(64, "quad") > 0          # (the first version's swizzle was not)
    sch = load("conv3_schedule_check")
    pw, imm = sch.kernel_immediates()
    assert sch.run(pw, imm) == []
    for key in imm:
        assert sch.run(pw, dict(imm, **{key: imm[key] + 1})), key
    assert sch.run([pw[0] + 1, pw[1]], imm) and sch.run([pw[0], pw[1] + 1], imm)
    # the weight permutation the tool's host side applies (tools/conv3/conv3_ops.py) is the emulation's
    w = torch.randn(320, 9 * 64).to(torch.bfloat16)
    wp = w.view(320, 9, 2, 32).permute(0, 2, 1, 3).reshape(320, 9 * 64).contiguous()
    assert torch.equal(wp.float(), torch.from_numpy(emu.permute_weight(w.float().numpy())))
    with open(os.path.join(root, "tools", "conv3", "conv3_ops.py")) as f:
        assert "w.view(n, 9, c // 32, 32).permute(0, 2, 1, 3).reshape(n, k)" in f.read()


def test_ring_coop_split_policy_is_a_function_of_per_item_facts():
    """ops.ring_coop_applies (the cooperative two-way K split of the persistent kernel, vx_gemm_params.ring_hint = 2): the
    16x16-level long-K launches take it whatever the number of CFG halves / windows in the call; shorter K, other levels,
    an odd number of 64-channel chunks, a folded LayerNorm or fp32 output do not.  Host code only (the library's
    eligibility function needs no GPU)."""
    import ctypes as C
    from v_express_amd import lib as L, ops

    def params(m, n, k, **kw):
        p = L.GemmParams()
        p.m, p.n, p.k, p.c1, p.c2, p.kh, p.kw, p.stride = m, n, k, k, 0, 1, 1, 1
        p.nb, p.h_in, p.w_in, p.h_out, p.w_out = m // 256, 16, 16, 16, 16
        p.lda1, p.ldc, p.epi, p.alpha = k, n, L.VX_EPI_STORE, 1.0
        p.a = p.w = p.out = C.c_void_p(256)            # (never dereferenced by the eligibility functions)
        for key, v in kw.items():
            setattr(p, key, v)
        return p

    assert L.lib.vx_gemm_ring_coop_ok(C.byref(params(8192, 1280, 11520))) == 1
    assert L.lib.vx_gemm_ring_coop_ok(C.byref(params(8192, 1280, 64 * 45))) == 0          # odd chunk count
    assert L.lib.vx_gemm_ring_coop_ok(C.byref(params(8192, 1280, 11520, out_f32=1))) == 0
    assert L.lib.vx_gemm_ring_coop_ok(C.byref(params(8192, 1280, 11520, epi=L.VX_EPI_GEGLU))) == 0
    assert L.lib.vx_gemm_ring_coop_ok(C.byref(params(8192, 1200, 11520))) == 0            # n % 320
    for items in (1, 2, 4, 6):
        with ops.frame_rows(256, items=items):
            assert ops.ring_coop_applies(params(items * 4096, 1280, 11520))
            assert ops.ring_coop_applies(params(items * 4096, 1280, 23040))
            assert not ops.ring_coop_applies(params(items * 4096, 1280, 5120))         # below COOP_MIN_K
            assert not ops.ring_coop_applies(params(items * 4096, 2560, 11520))        # 256 tiles per pair: plain ring
            assert not ops.ring_coop_applies(params(items * 1024, 1280, 11520))        # the 8x8 level (classic split-K)
    assert not ops.ring_coop_applies(params(8192, 1280, 11520))                        # no frame_rows context: no per-item facts
    with ops.frame_rows(256, items=2):
        ops.RING_COOP[0] = False
        try:
            assert not ops.ring_coop_applies(params(8192, 1280, 11520))
        finally:
            ops.RING_COOP[0] = True


def test_cooperative_split_rendezvous_protocol_all_interleavings():
    """tools/coop_protocol_check.py (the epoch protocol of ABI 14): every interleaving of the two partner waves of the
    cooperative K split (and of the asynchronous drain of the first wave's stores), three launches in a row on the same
    workspace slot, from every state an aborted earlier launch can leave behind - the second wave reads only complete partner
    data of its own launch, exactly one runs the epilogue, nobody blocks; and each of the three ingredients (wait before the
    flag, the epoch in the flag, data before flag) is necessary."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "coop_protocol_check.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    for brk in (1, 2, 3):
        r = subprocess.run([sys.executable, tool, "--break", str(brk)], capture_output=True, text=True)
        assert r.returncode == 1 and "FAILED" in r.stdout, (brk, r.stdout)
