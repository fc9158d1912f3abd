"""Oracle restatement of the wav2vec2 audio encoder — TEST INFRASTRUCTURE.

The reference feeds `Wav2Vec2Model(audio).last_hidden_state` into the audio-window construction
(pipelines/v_express_pipeline.py:374-378; loaded from `facebook/wav2vec2-base-960h` by inference.py:165-166).  The
model lives in a third-party dependency (transformers==4.41.1, requirements.txt:10; 5.x is what this image has) and is
restated here from its published architecture for the one variant V-Express uses: `feat_extract_norm="group"`,
`do_stable_layer_norm=False`, no convolution bias, eval mode (no SpecAugment masking, no dropout, no LayerDrop).
Plain fp32 PyTorch over a flat weight dict with the transformers state_dict key names.  Pinned against the installed
transformers `Wav2Vec2Model` on the same weights (tests/test_oracle_vs_reference.py) and through
tests/golden/wav2vec2.pt.
"""
import torch
import torch.nn.functional as F


def pos_conv_weight(w):
    """Effective weight of `encoder.pos_conv_embed.conv` under weight_norm(dim=2): g * v / ||v|| with the norm taken
    over (out, in) for every tap.  Accepts the parametrization keys (torch >= 2.1) and the legacy weight_g/weight_v."""
    p = "encoder.pos_conv_embed.conv."
    if p + "weight" in w:
        return w[p + "weight"]
    if p + "parametrizations.weight.original0" in w:
        g, v = w[p + "parametrizations.weight.original0"], w[p + "parametrizations.weight.original1"]
    else:
        g, v = w[p + "weight_g"], w[p + "weight_v"]
    return g * v / v.norm(dim=(0, 1), keepdim=True)


def normalize_waveform(wav):
    """Wav2Vec2FeatureExtractor(do_normalize=True): zero mean, unit variance, eps 1e-7 (the `audio_processor` step of
    pipelines/v_express_pipeline.py:375)."""
    wav = wav.float()
    return (wav - wav.mean()) / torch.sqrt(wav.var(unbiased=False) + 1e-7)


def feature_encoder(w, wav, strides=(5, 2, 2, 2, 2, 2, 2)):
    """Wav2Vec2FeatureEncoder: conv1d stack on the raw waveform [B, T]; layer 0 is conv -> GroupNorm(C groups) -> GELU,
    the rest conv -> GELU; no bias.  Returns [B, T', C]."""
    h = wav[:, None, :]
    for i, s in enumerate(strides):
        p = f"feature_extractor.conv_layers.{i}"
        h = F.conv1d(h, w[p + ".conv.weight"], w.get(p + ".conv.bias"), stride=s)
        if i == 0:
            c = h.shape[1]
            h = F.group_norm(h, c, w[p + ".layer_norm.weight"], w[p + ".layer_norm.bias"], eps=1e-5)
        h = F.gelu(h)
    return h.transpose(1, 2)


def _ln(w, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], eps)


def _lin(w, p, x):
    return F.linear(x, w[p + ".weight"], w[p + ".bias"])


def encoder_layer(w, p, x, heads, eps):
    """Wav2Vec2EncoderLayer (post-LN): x = LN(x + attn(x)); x = final_LN(x + FF(x))."""
    b, t, c = x.shape
    d = c // heads

    def split(y):
        return y.view(b, t, heads, d).transpose(1, 2)
    q, k, v = (split(_lin(w, f"{p}.attention.{n}_proj", x)) for n in ("q", "k", "v"))
    a = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1) @ v
    a = _lin(w, p + ".attention.out_proj", a.transpose(1, 2).reshape(b, t, c))
    x = _ln(w, p + ".layer_norm", x + a, eps)
    f = _lin(w, p + ".feed_forward.output_dense", F.gelu(_lin(w, p + ".feed_forward.intermediate_dense", x)))
    return _ln(w, p + ".final_layer_norm", x + f, eps)


def forward(w, wav, layers, heads, groups=16, strides=(5, 2, 2, 2, 2, 2, 2), eps=1e-5):
    """Wav2Vec2Model.forward(input_values [B, T]).last_hidden_state -> [B, T', hidden]."""
    feats = feature_encoder(w, wav.float(), strides)
    x = _lin(w, "feature_projection.projection", _ln(w, "feature_projection.layer_norm", feats, eps))
    wp = pos_conv_weight(w)
    k = wp.shape[-1]
    pos = F.conv1d(x.transpose(1, 2), wp, w["encoder.pos_conv_embed.conv.bias"], padding=k // 2, groups=groups)
    if k % 2 == 0:
        pos = pos[:, :, :-1]                       # Wav2Vec2SamePadLayer
    x = _ln(w, "encoder.layer_norm", x + F.gelu(pos).transpose(1, 2), eps)
    for i in range(layers):
        x = encoder_layer(w, f"encoder.layers.{i}", x, heads, eps)
    return x
