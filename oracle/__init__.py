"""CPU oracle for the V-Express denoising hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 functional restatement of the reference's per-step path
(`VExpressPipeline.mean_overlap` loop, `UNet3DConditionModel.forward`, the ReferenceNet
bank write, DDIM v-prediction step, sd-vae-ft-mse decode) over a flat weight dict that uses
the reference's own state_dict key names.  Each function cites the reference file:line it
follows (paths relative to /root/reference).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package, and only as the checker / reported CPU baseline.  The product path
(`v-express_amd/`) never imports it and fails loudly if its HIP library is missing.

Pinning: the reference ships no tests/golden vectors for this path, so the
oracle is pinned against the reference ITSELF: /root/reference/modules + pipelines are
imported unmodified (over tests/_shim/diffusers, a stand-in for the un-installable
diffusers==0.29.2 leaf classes) in tests/test_oracle_vs_reference.py, and golden outputs of
that run are committed under tests/golden/ (generator: tests/make_golden.py) so the GPU
box — which has no /root/reference — can detect drift.  The diffusers leaf arithmetic
(Attention, GEGLU FeedForward, Timesteps, ResnetBlock2D, DDIMScheduler, AutoencoderKL) lives
in an absent third-party dependency (diffusers==0.29.2, requirements.txt:1) and is restated
from its published behaviour in BOTH the stand-in and here; for those leaves parity is
"restated, not pinned by reference-owned vectors" — stated as such in DESIGN.md.
"""
from .config import UNetConfig, VaeConfig  # noqa: F401
