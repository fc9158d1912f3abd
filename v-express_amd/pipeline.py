"""Drop-in for the reference `VExpressPipeline` (pipelines/v_express_pipeline.py:71-646) with the hot loop on
libvexpress_hip kernels and everything resident in HBM.

Same constructor kwargs and `__call__` signature/defaults as the reference (SURVEY.md §8b surface #1); returns
a float32 `[1, 3, F, H, W]` tensor in [0, 1] (on the CPU like the reference unless `output_device` is given).

Differences that change memory/time but not results: latents, kps features and
predictions never visit the host (the reference keeps latents and kps features on the CPU and copies a window
up/down every step: :363,:521,:531,:538,:572); the per-frame Python bookkeeping of :552-572 is replayed once on
the host into a static plan (context.overlap_plan) and executed by two small kernels; CFG + 1/count + sum + DDIM
step are fused; the VAE decodes frames in batches.  With `torch.distributed` initialised the (window, CFG-half)
units of a timestep are sharded over the ranks (distributed.py) — the reference's
`do_multi_devices_inference` flag is accepted and, as in the reference, changes nothing by itself.

Once-per-clip prologue: VKpsGuider, AudioProjection and the VAE encoder run on the HIP
kernels when the v_express_amd classes are passed (prologue.py, vae.AutoencoderKL); wav2vec2 stays a user-provided
transformers module.  The benchmark and the loop parity tests pass the prologue outputs in directly
(`reference_latents=`, `kps_features=`, `audio_embeddings=`, `latents=` keyword arguments).
"""
from typing import Callable, List, Optional, Union

import os

import torch

from . import lib as L
from . import ops
from .context import get_context_scheduler, overlap_plan
from .distributed import (DistContext, MixedUnitSchedule, UnitSchedule, choose_frame_shards, choose_mixed_shards,
                          split_frames)
from .mutual_self_attention import ReferenceAttentionControl


def _in_unet_element_type(fn):
    """Pipeline entry points launch kernels themselves (gather / combine / DDIM, layout changes): they run under the
    denoising UNet's 16-bit element type (lib.element_type: bfloat16, or IEEE half for a float16 model)."""
    import functools

    @functools.wraps(fn)
    def run(self, *args, **kwargs):
        with L.element_type(_pipeline_element(self)):
            return fn(self, *args, **kwargs)
    return run


def _pipeline_element(pipe):
    """The denoising UNet's element type; a partial pipeline (the prologue helpers are callable on any object that
    carries the components they use) falls back to the first component that has one, then to the type in force."""
    for name in ("denoising_unet", "reference_net", "vae", "audio_projection", "audio_encoder", "v_kps_guider"):
        elem = getattr(getattr(pipe, name, None), "_elem", None)
        if elem is not None:
            return elem
    return L.ELEM[0]


class VExpressPipeline:
    def __init__(self, vae, reference_net, denoising_unet, v_kps_guider=None, audio_processor=None,
                 audio_encoder=None, audio_projection=None, scheduler=None, image_proj_model=None, tokenizer=None,
                 text_encoder=None):
        self.vae = vae
        self.reference_net = reference_net
        self.denoising_unet = denoising_unet
        self.v_kps_guider = v_kps_guider
        self.audio_processor = audio_processor
        self.audio_encoder = audio_encoder
        self.audio_projection = audio_projection
        self.scheduler = scheduler
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self.dist = DistContext.from_env()
        # ranks per (window, CFG-half) unit, each holding 1/S of the window's frames; None = automatic (S > 1 only
        # when the clip has fewer units than ranks, distributed.choose_frame_shards)
        self.frame_shards = None
        # uneven clips (units % world != 0): frame-shard only the left-over units so that every rank carries the same
        # load (distributed.MixedUnitSchedule); None = automatic, 1 = never (whole units only, round-2 behaviour).  An
        # explicit frame_shards = 1 ("no frame sharding") also means whole units only unless mixed_shards is set as well.
        self.mixed_shards = None
        # the schedule the last denoise() call chose (for logs / bench.py): dict(kind, frame_shards, mixed_shards, ...)
        self.last_schedule = {}
        # batch rows per UNet call: 2 = the two CFG halves of one window; 4 (default), 6, ... also merge consecutive
        # windows of this rank into one call.  Every kernel is batch-invariant, so the rows come out bit-identical;
        # merged calls measure 4-5 % faster (profiles/r02e_host_overhead.json: b = 3 74.0 ms vs 49.2 + 28.2 ms,
        # b = 4 94.1 vs 2 x 49.2 ms at 512x512, f = 16), which is what the 3-unit ranks of the 8-GPU config-4 run and
        # the multi-window single-GPU clips execute
        self.units_per_call = int(os.environ.get("VX_UNITS_PER_CALL", "4"))
        self.last_timing = {}

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return self.denoising_unet.device

    @property
    def dtype(self):
        return self.denoising_unet.dtype

    def to(self, *args, **kwargs):
        for m in (self.vae, self.reference_net, self.denoising_unet):
            m.to(*args, **kwargs)
        return self

    # ------------------------------------------------------------------ once-per-clip prologue (reference hooks)
    @staticmethod
    def _preprocess_image(image, height, width, normalize):
        """diffusers VaeImageProcessor.preprocess as configured at pipelines/v_express_pipeline.py:112-119
        (do_convert_rgb, resize with LANCZOS, [0,1]; `normalize`: 2x-1 for the reference image only)."""
        import numpy as np
        from PIL import Image
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image[None]
        else:
            arr = np.asarray(image.convert("RGB").resize((width, height), resample=Image.LANCZOS), dtype=np.float32)
            t = torch.from_numpy(arr / 255.0).permute(2, 0, 1)[None]
        return 2.0 * t - 1.0 if normalize else t

    @_in_unet_element_type
    def prepare_reference_latent(self, reference_image, height, width):
        """pipelines/v_express_pipeline.py:343-348: VAE-encode the reference image (posterior mean) * 0.18215.  Runs on
        the HIP VAE encoder when `vae` is a v_express_amd.AutoencoderKL (encoder weights loaded)."""
        if not hasattr(self.vae, "encode"):
            raise NotImplementedError("this VAE has no encoder half (AutoencoderKLDecoder): construct "
                                      "v_express_amd.AutoencoderKL and load encoder.* / quant_conv.*, or pass "
                                      "reference_latents=[1,4,h/8,w/8] (already scaled by 0.18215)")
        x = self._preprocess_image(reference_image, height, width, normalize=True)
        return self.vae.encode(x).latent_dist.mean * 0.18215

    @_in_unet_element_type
    def prepare_kps_tokens(self, kps_images, height, width, do_classifier_free_guidance):
        """prepare_kps_feature (:350-372) on the device, returning the token layout the loop consumes:
        bf16 `[2, F, hw, C0]` (row 0 = the all-zero unconditional half).  Needs a v_express_amd.VKpsGuider."""
        frames = [self._preprocess_image(img, height, width, normalize=False).unsqueeze(2) for img in kps_images]
        x = torch.cat(frames, dim=2)                                          # [1, 3, F, H, W]
        toks = []
        for i in range(0, x.shape[2], 16):                                    # :359-366 (chunks of 16 frames)
            t, h, w = self.v_kps_guider.forward_tokens(x[:, :, i:i + 16])
            toks.append(t)
        tok = torch.cat(toks, dim=0).view(1, x.shape[2], h * w, -1)
        if do_classifier_free_guidance:
            tok = torch.cat([torch.zeros_like(tok), tok], dim=0)
        return tok

    @_in_unet_element_type
    def prepare_kps_feature(self, kps_images, height, width, do_classifier_free_guidance):
        """pipelines/v_express_pipeline.py:350-372 with the reference's return layout `[2, C, F, h, w]` float32."""
        if self.v_kps_guider is None:
            raise NotImplementedError("no v_kps_guider given; pass kps_features=[2,320,F,h/8,w/8]")
        if hasattr(self.v_kps_guider, "forward_tokens"):
            tok = self.prepare_kps_tokens(kps_images, height, width, do_classifier_free_guidance)
            b2, F_, hw, c = tok.shape
            h = height // self.vae_scale_factor
            return tok.float().view(b2, F_, h, hw // h, c).permute(0, 4, 1, 2, 3).contiguous()
        frames = [self._preprocess_image(img, height, width, normalize=False).unsqueeze(2) for img in kps_images]
        x = torch.cat(frames, dim=2).to(self.device)                          # a user-provided torch module
        feats = [self.v_kps_guider(x[:, :, i:i + 16].to(next(self.v_kps_guider.parameters()).dtype)).float()
                 for i in range(0, x.shape[2], 16)]
        feat = torch.cat(feats, dim=2)
        if do_classifier_free_guidance:
            feat = torch.cat([torch.zeros_like(feat), feat], dim=0)
        return feat

    @_in_unet_element_type
    def prepare_audio_embeddings(self, audio_waveform, video_length, num_pad_audio_frames,
                                 do_classifier_free_guidance):
        """pipelines/v_express_pipeline.py:374-407.  `audio_processor` / `audio_encoder` are v_express_amd's
        WaveformProcessor / Wav2Vec2Model (HIP kernels) or the transformers objects the reference uses - anything with
        the same call signature; the window construction and the AudioProjection run here."""
        if self.audio_encoder is None or self.audio_projection is None or self.audio_processor is None:
            raise NotImplementedError("no audio modules given; pass audio_embeddings=[2,F,5,768]")
        from .prologue import audio_windows
        wav = self.audio_processor(audio_waveform, return_tensors="pt", sampling_rate=16000)["input_values"]
        if getattr(self.audio_encoder, "wants_fp32_input", False):
            enc_dtype = torch.float32            # the HIP encoder reads the raw waveform in float32
        else:
            enc_dtype = next(self.audio_encoder.parameters()).dtype
        emb = self.audio_encoder(wav.to(self.device, enc_dtype)).last_hidden_state
        per_frame = audio_windows(emb, video_length, num_pad_audio_frames).to(enc_dtype)
        out = self.audio_projection(per_frame).unsqueeze(0)
        if do_classifier_free_guidance:
            out = torch.cat([torch.zeros_like(out), out], dim=0)
        return out

    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device,
                        generator, latents=None):
        """pipelines/v_express_pipeline.py:189-224: N(0,1) drawn on the CPU generator, times init_noise_sigma.
        Drawn in fp32 so the draw does not depend on the compute dtype (SURVEY.md §8c RNG note)."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = torch.randn(shape, generator=generator, device="cpu", dtype=torch.float32)
        return latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------ the hot loop
    @_in_unet_element_type
    def denoise(self, latents, kps_tokens, audio, timesteps, windows, guidance_scale, callback=None,
                callback_steps=1):
        """pipelines/v_express_pipeline.py:526-583.  latents fp32 [1,4,F,h,w] (device, updated in place);
        kps_tokens bf16 [b, F, hw, C0]; audio bf16 [b, F, n_ctx, 768] with b = 2 (uncond, cond) under classifier-free
        guidance (guidance_scale > 1, :443) and b = 1 (the conditional row only) without."""
        unet, dc, dev = self.denoising_unet, self.dist, latents.device
        _, C, F, H, W = latents.shape
        hw = H * W
        f = len(windows[0])
        if any(len(w) != f for w in windows):
            raise ValueError("all context windows must have the same length")
        nW = len(windows)
        plan = overlap_plan(windows, F)
        win_ids = torch.tensor(windows, dtype=torch.int32, device=dev)
        win_ids_long = win_ids.long()
        sf = plan["step_frames"]
        terms = torch.full((len(sf), plan["max_terms"], 2), -1, dtype=torch.int32)
        for i, fr in enumerate(sf):
            for j, (wi, li) in enumerate(plan["terms"][fr]):
                terms[i, j, 0], terms[i, j, 1] = wi, li
        terms = terms.to(dev)
        frame_ids = torch.tensor(sf, dtype=torch.int32, device=dev)
        counts = torch.tensor([float(plan["counts"][fr]) for fr in sf], dtype=torch.float32, device=dev)
        # work units of this rank; S > 1: the window's frames are split over S ranks per unit (short clips)
        do_cfg = guidance_scale > 1.0
        halves_n = 2 if do_cfg else 1
        if kps_tokens.shape[0] != halves_n or audio.shape[0] != halves_n:
            raise ValueError(f"guidance_scale={guidance_scale} needs {halves_n} batch row(s) of kps features / audio "
                             f"embeddings, got {kps_tokens.shape[0]} / {audio.shape[0]}")
        min_hw = (H // 8) * (W // 8)
        S = self.frame_shards or choose_frame_shards(nW, dc.world_size, f, min_hw, halves_n)
        if S < 1 or dc.world_size % S or f % S or min_hw % S:
            raise ValueError(f"frame_shards={S} must divide the world size ({dc.world_size}), the window length ({f}) "
                             f"and the {H // 8}x{W // 8} tokens of the coarsest UNet level")
        # Three schedules, one exchange format.  G = frame granules per unit in the exchange buffer:
        #   uniform, S = 1: every unit whole on one rank (G = 1);  uniform, S > 1: every unit on S ranks (G = S);
        #   mixed: floor(units / world) whole units per rank + the left-over units sharded Sm ways (G = Sm): every rank
        #   then carries the same load (the 20 units of the config-4 clip on 8 GPUs: 2 + 1/2 per rank instead of 3 | 2).
        Sm = 1
        whole_units_only = self.mixed_shards == 1 or (self.frame_shards == 1 and self.mixed_shards is None)
        if S == 1 and dc.enabled and not whole_units_only:
            Sm = self.mixed_shards or choose_mixed_shards(nW * halves_n, dc.world_size, f, min_hw)
        self.last_schedule = dict(kind="mixed" if Sm > 1 else ("frame-sharded" if S > 1 else "whole units"),
                                  frame_shards=S, mixed_shards=Sm, units=nW * halves_n, world=dc.world_size)
        if Sm > 1:
            sched_m = MixedUnitSchedule(nW, dc.world_size, Sm, halves_n)
            G, max_slots, unit_slots = Sm, sched_m.max_slots, sched_m.slots
            # (call groups, frame shards of these calls): this rank's whole units, then its share of one sharded unit
            su = sched_m.split_unit(dc.rank)
            plan_calls = [(sched_m.whole_calls(dc.rank), 1), ([(su[0], [su[1]])], Sm)]
        else:
            sched_u = UnitSchedule(nW, dc.world_size, S, halves_n)
            G, max_slots = S, sched_u.max_units
            unit_slots = {}
            for u in sched_u.slot:
                ranks, slot = sched_u.unit_ranks(u)
                unit_slots[u] = [(r, slot) for r in ranks]
            plan_calls = [(sched_u.calls(dc.rank), S)]
        g_frames = f // G                              # frames per exchange granule
        # per-timestep exchange: only conv_out's C real channels travel (the GEMM pads them to 8); unit_index tells the
        # combine kernel which gathered slot holds frame granule j of (window, CFG half)
        local = torch.zeros((max_slots, g_frames * hw, C), device=dev, dtype=torch.float32)
        preds = torch.empty((nW, C, f, hw), device=dev, dtype=torch.float32)
        uidx = torch.empty((nW, halves_n, G), dtype=torch.int32)
        for wi in range(nW):
            for hlf in range(halves_n):
                for j, (r, slot) in enumerate(unit_slots[(wi, hlf)]):
                    uidx[wi, hlf, j] = r * max_slots + slot
        uidx = uidx.to(dev)
        # per-call constants (window ids, conditioning slices) do not depend on the timestep: build them once so
        # the timestep loop issues kernels only (no host->device copies, no syncs)
        # which CFG halves carry all-zero audio tokens (the unconditional half, :403-405): one device reduction per clip
        audio_is_zero = [bool((audio[hh] == 0).all().item()) for hh in range(audio.shape[0])]
        # UNet calls of this rank: the units of one window always share a call; `units_per_call` > 2 also merges
        # consecutive windows into one batch (every kernel is batch-invariant, so the rows come out identical - only
        # the launches get fatter, which helps the 16x16 / 8x8 levels of multi-window clips)
        limit = int(self.units_per_call)
        calls = []
        for my_calls, Sc in plan_calls:
            shard = dc.frame_shard(Sc)                 # collective when it creates the groups: every rank gets here
            f_loc = f // Sc
            lo = (dc.rank % Sc) * f_loc                # this rank's frames of the windows of these calls: [lo, lo+f_loc)
            merged, cur = [], []
            for wi, halves in my_calls:
                if cur and (limit <= 2 or sum(len(h) for _, h in cur) + len(halves) > limit):
                    merged.append(cur)
                    cur = []
                cur.append((wi, halves))
            if cur:
                merged.append(cur)
            for group in merged:
                rows = [(wi, hlf) for wi, halves in group for hlf in halves]      # batch rows of the call, in order
                kps_l, ehs_l = [], []
                for wi, halves in group:
                    hsel = torch.tensor(halves, device=dev)
                    ids_long = win_ids_long[wi][lo:lo + f_loc]
                    kps_l.append(kps_tokens.index_select(0, hsel).index_select(1, ids_long)
                                 .reshape(len(halves) * f_loc, hw, -1))
                    e = audio.index_select(0, hsel).index_select(1, ids_long)
                    ehs_l.append(e.reshape(-1, e.shape[-1]))
                kps = torch.cat(kps_l, dim=0).contiguous()
                ehs = torch.cat(ehs_l, dim=0).contiguous()
                gathers = [(win_ids[wi][lo:lo + f_loc].contiguous(), len(halves)) for wi, halves in group]
                # send slots of the call: its units occupy consecutive slots in row order (a whole unit of a mixed
                # schedule = G consecutive granules, a sharded unit this rank's one granule)
                s0 = min(slot for (r, slot) in unit_slots[rows[0]] if r == dc.rank)
                n_slots = len(rows) * (G // Sc)
                # the audio K | V of all 16 transformer blocks is step-invariant: once per clip and call
                calls.append((rows, gathers, kps, ehs, unet.precompute_audio_kv(ehs), f_loc, shard, s0, n_slots))
        for i, t in enumerate(timesteps):
            t = int(t)
            for rows, gathers, kps, ehs, akv, f_loc, shard, s0, n_slots in calls:
                parts = [ops.gather_latents(latents, ids, reps=reps) for ids, reps in gathers]
                x_in = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
                out = unet.forward_tokens(x_in, t, ehs, kps, b=len(rows), f=f_loc, H=H, W=W,
                                          batch_rows=[hlf for _, hlf in rows], audio_kv=akv,
                                          audio_zero=[audio_is_zero[hlf] for _, hlf in rows], frame_shard=shard)
                # a call's units occupy consecutive send slots, in row order: one strided pack per call
                ops.pack_rows(out, C, local[s0:s0 + n_slots])
            gathered = dc.all_gather_units(local, max_slots)          # [world, max_slots, (f/G)*hw, C]
            # CFG combine of every window in one launch (:548-550; without CFG the prediction itself)
            ops.combine_units(gathered, uidx, C, f, hw, guidance_scale if do_cfg else 1.0, preds)
            ops.overlap_ddim_step(latents, preds, terms, frame_ids, counts, self.scheduler.step_coefficients(t))
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        return latents

    @torch.no_grad()
    @_in_unet_element_type
    def decode_latents(self, latents, chunk=8):
        """pipelines/v_express_pipeline.py:152-166; frames are split evenly over the ranks when distributed."""
        dc = self.dist
        F = latents.shape[2]
        if not dc.enabled:
            return self.vae.decode_video(latents, chunk=chunk)
        spans = split_frames(F, dc.world_size)
        lo, hi = spans[dc.rank]
        per = spans[0][1] - spans[0][0]
        part = self.vae.decode_video(latents[:, :, lo:hi].contiguous(), chunk=chunk) if hi > lo else None
        Hh, Ww = latents.shape[-2] * self.vae_scale_factor, latents.shape[-1] * self.vae_scale_factor
        buf = torch.zeros((per, 3, Hh, Ww), device=latents.device, dtype=torch.float32)
        if part is not None:
            buf[:hi - lo].copy_(part[0].permute(1, 0, 2, 3))
        allf = dc.all_gather_frames(buf).reshape(-1, 3, Hh, Ww)[:F]
        return allf.permute(1, 0, 2, 3).unsqueeze(0)

    # ------------------------------------------------------------------ reference call surface
    @torch.no_grad()
    @_in_unet_element_type
    def __call__(self, reference_image, kps_images, audio_waveform, width, height, video_length,
                 num_inference_steps, guidance_scale, strength=1., num_images_per_prompt=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: Optional[int] = 1, context_schedule="uniform", context_frames=24,
                 context_overlap=4, reference_attention_weight=1., audio_attention_weight=1.,
                 num_pad_audio_frames=2, do_multi_devices_inference=False, save_gpu_memory=False,
                 reference_latents=None, kps_features=None, audio_embeddings=None, latents=None,
                 output_device="cpu", decode=True, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is unused by V-Express")
        dev = self.device
        do_cfg = guidance_scale > 1.0
        # timesteps (retrieve_timesteps + get_timesteps, :448-449)
        self.scheduler.set_timesteps(num_inference_steps)
        init_t = min(int(num_inference_steps * strength), num_inference_steps)
        timesteps = self.scheduler.timesteps[max(num_inference_steps - init_t, 0):].tolist()
        writer = ReferenceAttentionControl(self.reference_net, do_classifier_free_guidance=do_cfg, mode="write",
                                           batch_size=1, fusion_blocks="full")
        reader = ReferenceAttentionControl(self.denoising_unet, do_classifier_free_guidance=do_cfg, mode="read",
                                           batch_size=1, fusion_blocks="full",
                                           reference_attention_weight=reference_attention_weight,
                                           audio_attention_weight=audio_attention_weight)
        if reference_latents is None:
            reference_latents = self.prepare_reference_latent(reference_image, height, width)
        kps_tokens = None
        if kps_features is None:
            if hasattr(self.v_kps_guider, "forward_tokens"):
                kps_tokens = self.prepare_kps_tokens(kps_images, height, width, do_cfg)   # stays in the token layout
            else:
                kps_features = self.prepare_kps_feature(kps_images, height, width, do_cfg)
        if audio_embeddings is None:
            audio_embeddings = self.prepare_audio_embeddings(audio_waveform, video_length, num_pad_audio_frames,
                                                             do_cfg)
        windows = list(get_context_scheduler(context_schedule)(
            step=0, num_frames=video_length, context_size=context_frames, context_stride=1,
            context_overlap=context_overlap, closed_loop=False))
        # ReferenceNet once per clip (:502-509)
        ehs0 = torch.zeros((1, 1, self.denoising_unet.cfg.cross_attention_dim), dtype=torch.float32, device=dev)
        self.reference_net(reference_latents.to(dev), timestep=0, encoder_hidden_states=ehs0, return_dict=False)
        reader.update(writer, do_cfg, dtype=self.dtype)
        lat = self.prepare_latents(num_images_per_prompt, self.denoising_unet.in_channels, width, height,
                                   video_length, self.dtype, dev, generator, latents)
        if self.dist.enabled and latents is None:
            # every rank drew from its own CPU generator; the loop needs identical step-start latents on all ranks
            # (each rank's UNet inputs are gathered from them and every rank applies the DDIM update): rank 0's draw wins
            lat = self.dist.broadcast(lat.contiguous(), src=0)
        if kps_tokens is None:
            b2, c0, F, h, w = kps_features.shape
            kps_tokens = ops.ncfhw_to_nhwc(kps_features.to(dev), c0).view(b2, F, h * w, c0)
        audio = audio_embeddings.to(device=dev, dtype=ops.BF16).contiguous()
        timed = lat.is_cuda          # (host-logic tests run this method on CPU tensors over emulated kernels)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if timed:
            ev[0].record()
        self.denoise(lat, kps_tokens, audio, timesteps, windows, guidance_scale, callback, callback_steps or 1)
        if timed:
            ev[1].record()
        reader.clear()
        writer.clear()
        if not decode:
            return lat
        video = self.decode_latents(lat)
        if timed:
            ev[2].record()
        self._events = ev
        if output_device is not None:
            video = video.to(output_device)
        return video

    def timings_ms(self):
        """(denoise loop, decode) GPU milliseconds of the last call."""
        e = self._events
        e[2].synchronize()
        return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
