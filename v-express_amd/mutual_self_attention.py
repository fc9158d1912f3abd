"""Drop-in for the reference `ReferenceAttentionControl` (modules/mutual_self_attention.py:18-387).

The reference monkey-patches every (Temporal)BasicTransformerBlock.forward; in this package the write / read
branches are the native block forwards (blocks.py), so the controller only
  * switches a UNet into 'write' or 'read' mode and sets the two attention weights (ctor),
  * `update(writer)`: pairs writer banks with reader blocks BY BLOCK NAME — what the reference's stable sort on
    `-norm1.normalized_shape[0]` over DFS order amounts to (both UNets register children as down -> up -> mid:
    mutual_self_attention.py:341-357; verified in SURVEY.md App. E3) — and precomputes, once per clip, the
    step-invariant K / V^T of every reader `attn1_5` from the bank,
  * `clear()`: drops the banks.
With classifier-free guidance the reader bank is `cat([zeros, v])` (:357-359): batch row 0 sees an all-zero bank,
for which attention returns exactly `to_out.bias`; that row is recorded as None and never runs the SDPA.
"""
import torch

from . import blocks as B
from . import lib as L


class ReferenceAttentionControl:
    def __init__(self, unet, mode="write", do_classifier_free_guidance=False, attention_auto_machine_weight=float("inf"),
                 gn_auto_machine_weight=1.0, style_fidelity=1.0, reference_attn=True, reference_adain=False,
                 fusion_blocks="midup", batch_size=1, reference_attention_weight=1., audio_attention_weight=1.,
                 reference_drop_rate=0.):
        assert mode in ["read", "write"]
        assert fusion_blocks in ["midup", "full"]
        if fusion_blocks != "full":
            raise NotImplementedError("the pipeline always uses fusion_blocks='full' (v_express_pipeline.py:456,463)")
        if reference_drop_rate != 0.:
            raise NotImplementedError("reference_drop_rate is a training-only option")
        self.unet = unet
        self.mode = mode
        self.reference_attn = reference_attn
        self.reference_adain = reference_adain
        self.fusion_blocks = fusion_blocks
        self.reference_attention_weight = reference_attention_weight
        self.audio_attention_weight = audio_attention_weight
        unet.reference_mode = mode
        unet.banks = {}
        if mode == "read":
            unet.reference_attention_weight = float(reference_attention_weight)
            unet.audio_attention_weight = float(audio_attention_weight)

    def update(self, writer, do_classifier_free_guidance=True, do_unconditional_forward=False, dtype=torch.float16):
        if not self.reference_attn:
            return
        reader, wbanks = self.unet, writer.unet.banks
        if not wbanks:
            raise RuntimeError("writer has no banks: run the ReferenceNet forward before update()")
        if writer.unet._elem != reader._elem:
            raise TypeError(f"ReferenceNet ({writer.unet.dtype}) and denoising UNet ({reader.dtype}) compute on different "
                            "16-bit element types: give both models the same dtype (inference.py:150-161 does)")
        with L.element_type(reader._elem):                      # the bank K / V projections run in the reader's element type
            self._update(reader, wbanks, do_classifier_free_guidance, do_unconditional_forward)

    def _update(self, reader, wbanks, do_classifier_free_guidance, do_unconditional_forward):
        P = reader._prepared()
        heads = reader.cfg.heads
        banks = {}
        for name, tokens in wbanks.items():
            if name not in P:
                raise KeyError(f"reader has no block {name}")
            if do_unconditional_forward and not do_classifier_free_guidance:
                rows = [None]                                   # zeros_like bank (:360-361)
            else:
                kv = B.bank_kv(P[name].attn1_5, tokens, heads)
                rows = [None, kv] if do_classifier_free_guidance else [kv]
            banks[name] = rows
        reader.banks = banks

    def clear(self):
        self.unet.banks = {}
