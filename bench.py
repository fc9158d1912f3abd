"""Benchmark of the V-Express denoising hot path on MI355X — BASELINE.json metric:
decoded frames/sec at 512x512, 25 DDIM steps.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...

One "step" = one full pass of the hot path over one synthetic clip: 25 DDIM steps of the sliding-window
mean-overlap loop (CFG UNet3D forwards + fused CFG/overlap/DDIM update) followed by the sd-vae-ft-mse decode of
every frame.  Inputs (latents, kps features, audio embeddings, reference banks) are resident in HBM when the
timed region starts; the once-per-clip prologue (ReferenceNet + bank K/V precompute) is timed separately.

Workload: N=1 is BASELINE.json configs[1] (16 frames = one 16-frame window).  N>1 defaults to BASELINE configs[3]
("strong" scaling, the north-star multi-GPU number): the F = 124 clip inference.py:255-264 makes of a 128-frame
request = 10 windows of 16 / overlap 4 = 20 (window, CFG-half) units sharded over the N ranks, one RCCL all-gather of
the 4-channel predictions per DDIM step, decode split over the ranks; rank 0 then also runs the SAME clip alone
(`same_clip_1gpu_fps`) so that the multi-GPU speed-up is a ratio of like with like.  `--scaling weak` keeps round 1's
F = 12*N + 4 (one window per GPU per step).  value = F*K / max-over-ranks time.
`python bench.py --gpus N` without a torchrun environment re-launches itself under torch.distributed.run with N ranks
(one per GPU, RCCL); a world size that differs from --gpus is an error.

The printed line is ONE compact JSON object (< 4 KB: `compact_line`; the 23 KB line of round 5 could not be parsed by the
driver): the contract's keys + `roofline` + `cpu_baseline`; every table (ranking, per-kernel, per-instantiation, HBM kernels,
block paths, the CPU workers) goes to the side file named by its `detail` key (`--detail`, default gpurun_out/bench_detail.json).
  roofline      dominant kernel = the ONE kernel instantiation with the most time in a whole clip, chosen over ALL profiled
                kernels - every vx_gemm launch (2*M*N*K FLOP), the attention kernels (4*B*H*Nq*Nkv*d at the true head dim),
                the one-launch feed-forward / temporal blocks and the HBM-bound kernels (algorithmic bytes) - from HIP-event
                durations of ONE extra instrumented DDIM step + the decode of 4 frames after the timed region (events on
                the launch stream, each launch weighted by how often it runs per clip); `rocprof` / `traffic` only from
                committed rocprofv3 files stamped with the loaded binary's identity; `whole_path` = fps * algorithmic FLOP
                per frame (66.4 TFLOP/frame at F=16) / peak.  Side file: `ranking`, `per_kernel`, ...
  cpu_baseline  kind "port": the fp32 oracle (`oracle/`, attention through F.scaled_dot_product_attention as the reference's
                AttnProcessor2_0 runs it) on the host cores, USING THE BOX: N worker processes pinned to disjoint sets of 8
                physical cores (frames are independent), each one frame pair - a CFG UNet3D forward at 512^2 with a 2-frame
                window (warmed at a quarter of the pixels) + one frame of VAE decode - extrapolated linearly (x8 frame pairs
                x25 steps + x16 frames of decode per clip; N clips side by side); `cores` = N x 8; the weights are built once
                and shared copy-on-write by the forked workers; hard 30 s bound (workers report how far they got).  The
                reference-module figure of SURVEY.md E6 (dev container) is in the side file beside it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 (MX K=128 forms)
UNET_TFLOP_PER_FRAME_FWD = 1.277   # 40.86 TFLOP per CFG forward of 2x16 frames at 512^2
VAE_TFLOP_PER_FRAME = 2.515


UNET_SDPA_TFLOP_PER_FRAME_FWD = 0.245   # attn1 + attn1_5 SDPA (2 x 3.92 TFLOP / 32 frame-forwards): quadratic in the tokens
VAE_SDPA_TFLOP_PER_FRAME = 0.0344       # single-head mid attention over 4096 tokens, d = 512


def flop_per_frame(num_frames, windows, steps, scale, ctx=16):
    """Algorithmic TFLOP per decoded frame.  `scale` = pixel count relative to 512x512: the SDPA terms
    grow with scale^2, everything else with scale (768x768: 113.97 TFLOP per CFG forward, 5.754 per decoded frame)."""
    fwd = (UNET_TFLOP_PER_FRAME_FWD - UNET_SDPA_TFLOP_PER_FRAME_FWD) * scale + UNET_SDPA_TFLOP_PER_FRAME_FWD * scale ** 2
    vae = (VAE_TFLOP_PER_FRAME - VAE_SDPA_TFLOP_PER_FRAME) * scale + VAE_SDPA_TFLOP_PER_FRAME * scale ** 2
    return steps * 2 * fwd * (ctx * windows / num_frames) + vae


def _lib_sha():
    from v_express_amd import lib
    return lib.LIB_SHA256 + (":f16" if lib.ELEM[0] is torch.float16 else "")      # (committed profiles are of the bf16 build)


def _symbol_forms(symbol):
    """The names a kernel instantiation can carry in a profiler file: the library prints a gemm_ring_kernel instantiation
    without a trailing default template argument (the cooperative-split flag, false for every launch but ring_hint = 2);
    the demangled symbol has it."""
    forms = [symbol]
    if symbol.startswith("gemm_ring_kernel<") and symbol.endswith(">") and symbol.count(",") == 5:
        forms.append(symbol[:-1] + ", false>")
    return forms


def _pmc_traffic(symbol):
    """HBM-side bytes per launch of one kernel instantiation from a committed rocprofv3 PMC file (tools/exp_pmc_bench.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; the
    counter sits on the L2's fabric side, so Infinity-Cache hits are included) - only a file taken with THIS build of the
    kernel library (its `lib_sha256` equals the loaded library's) is quoted; (None, reason) otherwise."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)
    sha = _lib_sha()
    seen_other = None
    for path in files:
        with open(path) as f:
            d = json.load(f)
        if d.get("lib_sha256") != sha:
            seen_other = seen_other or os.path.relpath(path, ROOT)
            continue
        ks = d.get("kernels", {})
        # (GEMM instantiations are keyed without their parameter list, every other kernel with it)
        v = next((x for f in _symbol_forms(symbol) for k, x in ks.items() if k == f or k.startswith(f + "(")), None)
        if v and v.get("launches") and v.get("fetch_bytes_per_launch", 0) > 0:
            return v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, (f"no counter file for library build {sha}" + (f" (newest other build: {seen_other})" if seen_other else ""))


def _rocprof_launch_avg(symbol):
    """Average launch duration of one kernel instantiation in a committed rocprofv3 kernel-trace summary
    (profiles/*_trace_summary.txt, tools/trace_summary.py over `rocprofv3 --kernel-trace --stats` of this command) taken
    with THIS build of the kernel library (header `# lib_sha256=`): same command + same library = the same launch
    population, so this run's algorithmic work per launch of the instantiation over that duration is a like-for-like
    figure without the HIP-event overhead.  None when no summary of this build is committed."""
    import glob
    import re
    sha = _lib_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_trace_summary.txt")), reverse=True):
        with open(path) as f:
            lines = f.read().splitlines()
        if not lines or f"lib_sha256={sha}" not in lines[0]:
            continue
        for ln in lines:
            if ln.startswith("#"):                        # header / gap-attribution comments of tools/trace_summary.py
                continue
            m = re.search(r"n=\s*(\d+)\s+avg=\s*([0-9.]+) us\s+(?:void )?(.*)$", ln)     # (non-template kernels: no "void")
            if m and any(m.group(3).startswith(f + "(") for f in _symbol_forms(symbol)):
                return dict(avg_launch_us=float(m.group(2)), launches=int(m.group(1)), file=os.path.relpath(path, ROOT))
    return None


HBM_PEAK_GBS = 8000.0              # HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches


def _core_cpus():
    """One logical CPU per physical core inside the affinity mask, in id order: the first sibling of every
    /sys/.../topology/thread_siblings_list group (a virtualised box whose sysfs lists no siblings counts every second CPU
    as a core when /proc/cpuinfo says "siblings" = 2 x "cpu cores", every CPU otherwise)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    firsts, ok = [], True
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                sib += list(range(int(lo), int(hi or lo) + 1))
            if c == min(x for x in sib if x in allowed):
                firsts.append(c)
        except (OSError, ValueError):
            ok = False
            break
    if ok and firsts:
        return firsts, len(allowed)
    return allowed[:max(1, len(allowed) // 2)] if len(allowed) >= 16 else allowed, len(allowed)


def _cpu_quota():
    """CPUs' worth of run time the container may use per unit of wall time (cgroup CFS quota), or None when unlimited.  The
    GPU boxes show 256 logical CPUs and a quota of 16 (cpu.max = "1600000 100000", profiles/r06c_cpu_box_probe.txt): every
    busy thread beyond 16 is throttled, which is why 128 threads were SLOWER than 16 in rounds 3-5 and why 16 pinned
    8-thread workers did not get through a quarter-size warm-up in 30 s (r06b)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                                  # cgroup v2
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:                     # cgroup v1
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


CPU_WORKER_THREADS = 8     # threads per worker process = the core count the dev-container Tier-A figure was taken on


def _reader_banks_for(cfg, h, OU):
    from v_express_amd.synth import block_plan
    plan = block_plan(cfg)
    g = torch.Generator().manual_seed(1)
    banks, hh = {}, h
    for blk in plan["down"]:
        if blk["attn"]:
            for j in range(len(blk["layers"])):
                banks[f"{blk['prefix']}.attentions.{j}"] = torch.randn(1, hh * hh, blk["c"], generator=g)
        if blk["sampler"]:
            hh //= 2
    banks["mid_block.attentions.0"] = torch.randn(1, hh * hh, plan["mid"]["c"], generator=g)
    for blk in plan["up"]:
        if blk["attn"]:
            for j in range(len(blk["layers"])):
                banks[f"{blk['prefix']}.attentions.{j}"] = torch.randn(1, hh * hh, blk["c"], generator=g)
        if blk["sampler"]:
            hh *= 2
    return OU.reader_banks(banks)


def cpu_leg(size, seconds_budget, max_workers=16):
    """The CPU leg's own process (`python bench.py --cpu-leg`: no GPU state, nothing multi-threaded before the forks).  Builds
    the fp32 weights ONCE on one thread (timing-only pool values), then forks N workers that share them copy-on-write; worker
    i pins itself to its CPU_WORKER_THREADS physical cores (os.sched_setaffinity), starts its own thread team and runs the
    fp32 oracle (`oracle/`, a port of the reference path; attention through F.scaled_dot_product_attention as
    AttnProcessor2_0 runs it) on ONE frame pair: a CFG UNet3D forward at size x size with a 2-frame window - warmed by the
    same forward at a quarter of the pixels - and one frame of VAE decode (warmed the same way).  Workers report every stage
    through a pipe, so a worker killed at the deadline still says how far it got.  Prints one JSON object."""
    import select
    import signal
    t_all = time.time()
    torch.set_num_threads(1)
    import oracle
    from oracle import leaf as OLF
    from oracle import unet as OU
    from oracle import vae as OV
    from v_express_amd import synth
    OLF.USE_SDPA[0] = True
    cfg, ocfg = synth.UNetConfig(), oracle.UNetConfig()
    f = 2
    sd3 = synth.unet3d_state_dict(cfg, timing_only=True)
    sdv = synth.vae_decoder_state_dict(synth.VaeConfig(), timing_only=True)
    weights_s = time.time() - t_all
    cores, logical = _core_cpus()
    quota = _cpu_quota()
    usable = len(cores) if quota is None else max(1, min(len(cores), int(quota)))     # what the container may keep busy
    T = min(CPU_WORKER_THREADS, usable)
    n = max(1, min(usable // T, max_workers))

    def worker(i, wfd):
        def say(**kw):
            os.write(wfd, (json.dumps(dict(worker=i, **kw)) + "\n").encode())
        cpus = cores[i * T:(i + 1) * T]
        if hasattr(os, "sched_setaffinity"):
            os.sched_setaffinity(0, cpus)
        torch.set_num_threads(T)

        def fwd(h):
            inp = synth.synthetic_inputs(cfg, f, h, h)
            rb = _reader_banks_for(cfg, h, OU)
            x = inp["latents"].repeat(2, 1, 1, 1, 1)
            ehs = inp["audio_embeddings"].reshape(-1, 5, 768)
            t0 = time.time()
            OU.unet3d_forward(sd3, ocfg, x, 519, ehs, inp["kps_features"], rb, 0.95, 3.0)
            return time.time() - t0, inp
        with torch.no_grad():
            warm_s, _ = fwd(size // 16)
            say(stage="warm", warm_s=warm_s)
            unet_s, inp = fwd(size // 8)
            say(stage="unet", unet_forward_s=unet_s)
            z = inp["latents"][0, :, :1].permute(1, 0, 2, 3).contiguous()
            OV.vae_decode(sdv, oracle.VaeConfig(), z[:, :, :size // 16, :size // 16].contiguous())     # warm
            t0 = time.time()
            OV.vae_decode(sdv, oracle.VaeConfig(), z)
            say(stage="done", vae_frame_s=time.time() - t0, cpus=cpus)

    pids, fds = {}, {}
    for i in range(n):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.close(r)
            code = 0
            try:
                worker(i, w)
            except BaseException as e:                                  # noqa: BLE001 - reported, the child must not return
                os.write(w, (json.dumps(dict(worker=i, stage="error", error=repr(e)[:200])) + "\n").encode())
                code = 1
            os._exit(code)
        os.close(w)
        pids[i], fds[r] = pid, i
    state = {i: dict(stage="started") for i in range(n)}
    buf = {r: b"" for r in fds}
    deadline = t_all + seconds_budget
    open_fds = set(fds)
    while open_fds and time.time() < deadline:
        ready, _, _ = select.select(list(open_fds), [], [], max(0.05, min(1.0, deadline - time.time())))
        for r in ready:
            chunk = os.read(r, 65536)
            if not chunk:
                open_fds.discard(r)
                continue
            buf[r] += chunk
            while b"\n" in buf[r]:
                line, buf[r] = buf[r].split(b"\n", 1)
                msg = json.loads(line)
                state[msg.pop("worker")].update(msg)
    killed = 0
    for i, pid in pids.items():
        if state[i].get("stage") not in ("done", "error"):
            try:
                os.kill(pid, signal.SIGKILL)
                killed += 1
            except ProcessLookupError:
                pass
        try:
            os.waitpid(pid, 0)
        except ChildProcessError:
            pass
    print(json.dumps(dict(workers=[dict(worker=i, **state[i]) for i in range(n)], killed=killed, started=n, threads=T,
                          weights_s=weights_s, physical_cores=len(cores), host_cpus=logical, cpu_quota=quota, frames=f,
                          leg_process_s=time.time() - t_all)))


def cpu_baseline(size, seconds_budget, max_workers=16):
    """The reference path's port (`oracle/`) on the host cores, using the box: cpu_leg in its own
    process - N forked workers pinned to disjoint sets of CPU_WORKER_THREADS physical cores (frames of a window are independent
    work, so N frame pairs run side by side), every worker one frame pair.  Whole-box figure: N workers finish N x 2
    frame-forwards in the mean worker time, so a 16-frame clip of 25 steps + decode takes clip_s = mean(unet_s) x 8 x 25 +
    mean(vae_s) x 16 per worker and the box delivers N clips in that time.  Hard bound: the leg kills its workers
    `seconds_budget` after its own start and reports how far each one got."""
    import subprocess
    t_all = time.time()
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("OMP_NUM_THREADS", None)
    res = None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", "--size", str(size), "--cpu-budget",
                            str(seconds_budget), "--cpu-max-workers", str(max_workers)], capture_output=True, text=True,
                           timeout=seconds_budget + 30, env=env)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(rows[-1]) if rows else None
        err = r.stderr[-300:]
    except subprocess.TimeoutExpired:
        err = "the CPU leg's process did not return"
    leg_s = time.time() - t_all
    if res is None:
        return dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"oracle fp32 CPU leg failed: {err}", cpu_leg_s=leg_s)
    T, n = res["threads"], res["started"]
    done = [w for w in res["workers"] if w.get("stage") == "done"]
    common = dict(unit="frames/s", kind="port", threads_per_process=T, workers=res["workers"], workers_killed=res["killed"],
                  workers_started=n, weights_s=res["weights_s"], physical_cores=res["physical_cores"],
                  host_cpus=res["host_cpus"], cpu_quota=res.get("cpu_quota"), cpu_leg_s=leg_s,
                  reference_modules_dev_container=dict(
                      value=16.0 / (25 * 78.1 + 16 * 5.5), cores=8, unet_forward_s=78.1, unet_core_seconds_per_frame=78.1 * 8 / 16,
                      note="SURVEY.md E6: the reference's own modules (fp32, f=16 CFG forward) on the 8 cores of the dev "
                           "container, measured once during the survey; not re-measured on the GPU box"))
    if not done:
        stages = sorted({w.get("stage", "?") for w in res["workers"]})
        return dict(value=None, cores=n * T, processes=n,
                    sample=f"oracle fp32 CPU leg: no worker of {n} finished within {seconds_budget} s (stages reached: {stages})",
                    **common)
    f = res["frames"]
    unet_s = sum(d["unet_forward_s"] for d in done) / len(done)
    vae_s = sum(d["vae_frame_s"] for d in done) / len(done)
    clip_s = unet_s * (16 / f) * 25 + vae_s * 16
    nd = len(done)
    lim = (f" (the container's CPU quota: {res['cpu_quota']:g} of {res['host_cpus']} visible CPUs)"
           if res.get("cpu_quota") else "")
    return dict(value=nd * 16.0 / clip_s, cores=nd * T, processes=nd,
                sample=(f"oracle/ (fp32 port of the reference path, SDPA attention) in {nd} pinned processes x {T} threads{lim}, "
                        f"each: CFG UNet3D forward {size}x{size} f={f} ({unet_s:.1f} s mean, warmed at quarter size) + 1 frame "
                        f"VAE decode ({vae_s:.1f} s); per process x{16 // f} frame pairs x25 steps + x16 frames = {clip_s:.0f} s "
                        f"per 16-frame clip, {nd} clips side by side"),
                unet_forward_s=unet_s, vae_frame_s=vae_s, unet_core_seconds_per_frame=unet_s * T / f, **common)


def kernel_rate(v):
    """(achieved, peak, unit, bound) of one kernel's launches: algorithmic FLOP/s against the dense bf16 MFMA peak for the
    MFMA kernels (flops > 0), algorithmic bytes/s against HBM3E otherwise."""
    if v["flops"] > 0:
        return v["flops"] / v["seconds"] / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s", "mfma"
    return v["bytes"] / v["seconds"] / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"


def rank_kernels(launches, top=12):
    """launches: dicts (symbol, seconds, weight, flops, bytes) - one per profiled launch of the instrumented leg, `weight` =
    how many times the launch runs per clip.  Returns (ranking, per-symbol totals, clip seconds per symbol, clip total):
    `ranking` = the `top` kernels by clip-weighted time over ALL kernels, MFMA or not (the line named an 8 %
    GEMM instantiation while a 12 % attention kernel existed); ranking[0] is the dominant kernel of the roofline."""
    allk, clip_s = {}, {}
    for ln in launches:
        d = allk.setdefault(ln["symbol"], dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0))
        d["launches"] += 1
        d["seconds"] += ln["seconds"]
        d["flops"] += ln["flops"]
        d["bytes"] += ln["bytes"]
        clip_s[ln["symbol"]] = clip_s.get(ln["symbol"], 0.0) + ln["weight"] * ln["seconds"]
    clip_total = sum(clip_s.values())
    ranking = []
    for sym, sec in sorted(clip_s.items(), key=lambda kv: -kv[1])[:top]:
        a, p, u, b = kernel_rate(allk[sym])
        ranking.append({"kernel": sym, "share_of_clip_kernel_time": sec / clip_total, "bound": b, "achieved": a, "unit": u,
                        "frac": a / p, "launches": allk[sym]["launches"],
                        "avg_launch_us": 1e6 * allk[sym]["seconds"] / allk[sym]["launches"]})
    return ranking, allk, clip_s, clip_total


LINE_LIMIT = 4096                  # bytes of the printed JSON line (tests/test_host_logic.py holds it there)


def _rounded(obj, sig=6):
    """Floats to `sig` significant digits, recursively (the line is read by people and a parser, not re-computed from)."""
    if isinstance(obj, float):
        return float(f"{obj:.{sig}g}") if obj == obj and abs(obj) != float("inf") else None
    if isinstance(obj, dict):
        return {k: _rounded(v, sig) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_rounded(v, sig) for v in obj]
    return obj


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


def write_detail(result, path):
    """The full result (every table) as a side file; returns the path relative to the repo root, or None if unwritable."""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
        return os.path.relpath(os.path.abspath(path), ROOT)
    except OSError as e:
        print(f"bench.py: detail file {path} not written: {e}", file=sys.stderr)
        return None


def compact_line(result, detail_path=None):
    """The ONE JSON line bench.py prints: the contract's keys + `roofline` (dominant kernel) + `cpu_baseline`, nothing
    that grows with the number of kernels or blocks.  Everything else is in the side file named by `detail`."""
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                          "vs_baseline", "dtype", "data"))
    line["config"] = _pick(result["config"], ("workload", "frames", "windows", "parallelism"))
    line.update(_pick(result, ("lib_sha256", "prologue_ms")))
    rf = result.get("roofline")
    if rf:
        c = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_source", "traffic", "traffic_source",
                       "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch", "avg_launch_us", "launches",
                       "share_of_clip_kernel_time", "all_mfma_tflops"))
        rp = rf.get("rocprof")
        c["rocprof"] = dict(avg_launch_us=rp["avg_launch_us"], frac=rp["frac"], source=rp["file"]) if rp else None
        c["whole_path"] = _pick(rf["whole_path"], ("tflop_per_frame", "achieved", "frac"))
        line["roofline"] = c
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "processes", "threads_per_process",
                                          "cpu_quota", "host_cpus", "unet_core_seconds_per_frame", "cpu_leg_s"))
        line["speedup_vs_cpu"] = result.get("speedup_vs_cpu")
    if result.get("n_gpus", 1) > 1:
        pr = result.get("per_rank")
        if pr:
            line["per_rank"] = _pick(pr, ("ms_per_step_min", "ms_per_step_max", "compute_ms_min", "compute_ms_max"))
        co = result.get("collectives_rank0")
        if co:
            line["collectives_rank0"] = {k: _pick(v, ("calls", "ms", "share_of_clip")) for k, v in co.items()}
        line.update(_pick(result, ("collective_backend", "measurement", "note", "same_clip_1gpu_fps",
                                   "speedup_vs_1gpu_same_clip")))
    line["detail"] = detail_path
    line = _rounded(line)
    if len(json.dumps(line)) >= LINE_LIMIT:              # never let free text push the line over what the driver parses
        for path in (("cpu_baseline", "sample"), ("config", "workload"), ("roofline", "traffic_source")):
            d = line.get(path[0])
            if isinstance(d, dict) and isinstance(d.get(path[1]), str):
                d[path[1]] = d[path[1]][:160]
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--frames", type=int, default=0, help="override the clip length")
    ap.add_argument("--context-frames", type=int, default=16,
                    help="window length (BASELINE configs use 16; the reference's own default is 24: inference.py:67)")
    ap.add_argument("--context-overlap", type=int, default=4)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N>1: strong = the F=124 clip of BASELINE configs[3] for every N; weak = F = 12*N+4")
    ap.add_argument("--no-same-clip-1gpu", action="store_true",
                    help="N>1: skip rank 0's single-GPU run of the same clip after the timed region")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[4]: attention q/k/v/out projections on the fp8 (e4m3) MFMA GEMM")
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default="bf16",
                    help="16-bit element type of the whole path: bf16 (BASELINE configs[1]) or fp16, the reference's own default "
                         "(inference.py:44) - the IEEE-half build of the kernel library, same kernels, same MFMA rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-leg", action="store_true", help=argparse.SUPPRESS)        # the CPU leg's own process (cpu_leg)
    ap.add_argument("--cpu-budget", type=float, default=30.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-max-workers", type=int, default=16, help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-shapes", default="", help="write the per-shape vx_gemm timing table of the roofline leg here")
    ap.add_argument("--detail", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="side file for the full tables (ranking, per-kernel, HBM kernels, block paths, CPU-leg probes)")
    args = ap.parse_args()

    if args.cpu_leg:
        cpu_leg(args.size, args.cpu_budget, args.cpu_max_workers)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get("VX_DIST_BACKEND", "nccl") == "nccl":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus})")
    # VX_DIST_BACKEND=gloo (+ ranks folded onto the visible GPUs) exists only to exercise the multi-rank control flow on
    # a single-GPU box; the measured configuration is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("VX_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import v_express_amd as vx
    from v_express_amd import lib as vxlib, ops, synth
    from v_express_amd.context import uniform

    elem = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    vxlib.ELEM[0] = elem                        # a bench process runs ONE element type: every ops call below uses its library
    if elem is torch.float16:
        vxlib.lib_f16()
    cfg, vcfg = synth.UNetConfig(), synth.VaeConfig()
    ctx, ovl = args.context_frames, args.context_overlap
    if args.frames:
        F = args.frames
    elif world == 1:
        F = ctx                                           # BASELINE configs[1]: one 16-frame window
    elif args.scaling == "strong":
        F = 124                                           # BASELINE configs[3]: inference.py:255-264 of a 128-frame request
    else:
        F = (ctx - ovl) * world + ovl
    scaling = "weak" if (world > 1 and args.scaling == "weak" and not args.frames) else "strong"
    h = w = args.size // 8
    t_build = time.time()
    unet = vx.UNet3DConditionModel(cfg).to(dev).to(elem)
    refnet = vx.UNet2DConditionModel(cfg).to(dev).to(elem)
    vae = vx.AutoencoderKLDecoder(vcfg).to(dev).to(elem)
    unet.load_state_dict(synth.unet3d_state_dict(cfg, seed=42, device=dev, dtype=elem, draw_on_device=True))
    unet.release_raw_weights()
    unet.fp8_projections = bool(args.fp8)
    refnet.load_state_dict(synth.refnet_state_dict(cfg, seed=43, device=dev, dtype=elem, draw_on_device=True))
    refnet.release_raw_weights()
    vae.load_state_dict(synth.vae_decoder_state_dict(vcfg, seed=44, device=dev, dtype=elem, draw_on_device=True))
    vae._prepared()
    sched = vx.DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                             steps_offset=1, prediction_type="v_prediction", rescale_betas_zero_snr=True,
                             timestep_spacing="trailing")
    pipe = vx.VExpressPipeline(vae=vae, reference_net=refnet, denoising_unet=unet, scheduler=sched)
    inp = synth.synthetic_inputs(cfg, F, h, w, seed=42, device=dev)
    torch.cuda.synchronize()
    t_build = time.time() - t_build

    # ---- once-per-clip prologue (ReferenceNet -> banks -> K/V of every attn1_5), timed separately
    writer = vx.ReferenceAttentionControl(refnet, do_classifier_free_guidance=True, mode="write", fusion_blocks="full")
    reader = vx.ReferenceAttentionControl(unet, do_classifier_free_guidance=True, mode="read", fusion_blocks="full",
                                          reference_attention_weight=0.95, audio_attention_weight=3.0)
    ehs0 = torch.zeros(1, 1, 768, device=dev)
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=ehs0, return_dict=False)   # warm
    torch.cuda.synchronize()
    t0 = time.time()
    refnet(inp["ref_latents"], timestep=0, encoder_hidden_states=ehs0, return_dict=False)
    reader.update(writer, True)
    torch.cuda.synchronize()
    prologue_s = time.time() - t0

    sched.set_timesteps(args.ddim_steps)
    timesteps = sched.timesteps.tolist()
    windows = list(uniform(step=0, num_frames=F, context_size=ctx, context_stride=1, context_overlap=ovl,
                           closed_loop=False))
    c0 = cfg.block_out_channels[0]
    kps_tokens = ops.ncfhw_to_nhwc(inp["kps_features"], c0).view(2, F, h * w, c0)
    audio = inp["audio_embeddings"].to(elem).contiguous()

    def one_clip():
        lat = inp["latents"].clone()
        pipe.denoise(lat, kps_tokens, audio, timesteps, windows, 3.5)
        return pipe.decode_latents(lat)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        video = one_clip()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        video = one_clip()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        # every rank's own wall time of the K clips (they leave the last barrier together, so the spread shows up in the
        # per-rank GPU-busy time below, not here) and the max that defines the reported value
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        rank_ms = [1e3 * t.item() / args.steps for t in allt]
        elapsed = max(t.item() for t in allt)
    assert video.shape == (1, 3, F, args.size, args.size) and torch.isfinite(video).all()
    fps = F * args.steps / elapsed
    scale = (args.size / 512.0) ** 2
    fpf = flop_per_frame(F, len(windows), args.ddim_steps, scale, ctx)
    from v_express_amd.distributed import choose_frame_shards, choose_mixed_shards
    fshards = pipe.frame_shards or choose_frame_shards(len(windows), world, ctx, (args.size // 64) ** 2)
    mshards = 1
    if fshards == 1 and world > 1 and pipe.mixed_shards != 1 and not (pipe.frame_shards == 1 and pipe.mixed_shards is None):
        mshards = pipe.mixed_shards or choose_mixed_shards(2 * len(windows), world, ctx, (args.size // 64) ** 2)

    result = {
        "metric": f"decoded frames/sec at {args.size}x{args.size}, {args.ddim_steps} DDIM steps", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": ("fp8-proj/" if args.fp8 else "") + args.dtype, "data": "synthetic",
        "config": {"workload": (f"{args.size}x{args.size}, {F} frames ({len(windows)} window(s) of {ctx}, overlap {ovl}), "
                                f"{args.ddim_steps} DDIM steps, CFG 3.5, random-init UNet3D + ReferenceNet banks + "
                                "sd-vae-ft-mse decode"),
                   "frames": F, "windows": len(windows), "frame_shards": fshards, "mixed_shards": mshards,
                   "parallelism": (f"window x CFG-half units over {world} GPU(s)" +
                                   (f", {fshards} frame shards per unit" if fshards > 1 else "") +
                                   (f", {2 * len(windows) // world} whole units per GPU + the {2 * len(windows) % world} "
                                    f"left-over units frame-sharded {mshards} ways" if mshards > 1 else ""))},
        "prologue_ms": 1e3 * prologue_s, "model_build_s": t_build, "lib_sha256": _lib_sha(),
        # which implementation each block of each UNet level took in this run (ops.BLOCK_PATHS: one-launch kernels vs the
        # multi-launch forms, GroupNorm folds) - a geometry that falls off a fused path shows here and as a log warning
        "block_paths": ops.block_paths(),
    }
    if args.fp8:
        # VERDICT r03 item 7, closed in LABNOTES 11.4: the e4m3 projections are a parity-complete OPTION, not a speed-up.
        result["fp8_note"] = ("fp8 (e4m3) q/k/v/out projections: parity-complete (tests/test_gpu_kernels.py, "
                              "test_gpu_models.py), not a speed-up - these are K = 320..1280 launches bound by HBM and "
                              "launch latency, fp8 halves one operand's bytes and adds a quantisation pass; measured "
                              "768x768: 4.01 (fp8) vs 4.50 (bf16) frames/s, profiles/r03f_bench_768_*.json.  Compare "
                              "this line's value with the same command without --fp8.")
    if world > 1:
        # One more clip with every collective of the data path timed (events on the launch stream) and this rank's own
        # compute time between the collectives: what a first real 8-GPU run needs to be diagnosable from this one line.
        from v_express_amd.distributed import CommTimer
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with CommTimer() as ct:
            ev0.record()
            one_clip()
            ev1.record()
        torch.cuda.synchronize()
        comm = ct.summary()
        clip_ms = ev0.elapsed_time(ev1)
        comm_ms = sum(v["ms"] for v in comm.values())
        mine = torch.tensor([clip_ms, comm_ms], device=dev, dtype=torch.float64)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        result["per_rank"] = {
            "ms_per_step_wall": rank_ms, "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms),
            "instrumented_clip_gpu_ms": [t[0].item() for t in allc],
            "instrumented_clip_collective_ms": [t[1].item() for t in allc],
            "compute_ms_min": min(t[0].item() - t[1].item() for t in allc),
            "compute_ms_max": max(t[0].item() - t[1].item() for t in allc),
            "note": "one extra clip after the timed region; collective_ms includes the wait for the slowest peer"}
        result["collectives_rank0"] = {k: {"calls": v["calls"], "ms": v["ms"], "mb": v["bytes"] / 1e6,
                                            "share_of_clip": v["ms"] / clip_ms} for k, v in comm.items()}
        result["schedule"] = dict(pipe.last_schedule)
        result["collective_backend"] = backend
        if backend != "nccl":
            # VX_DIST_BACKEND=gloo folds ranks onto the visible GPU(s) and stages every collective through the host: it
            # exercises the control flow on a 1-GPU box and is NEVER a multi-GPU measurement
            result["measurement"] = False
            result["note"] = f"collectives over {backend} (host-staged): control-flow run, not a multi-GPU measurement"
        else:
            assert pipe.dist.enabled and pipe.dist.backend == "nccl", "the measured multi-GPU path must be RCCL"

    if not args.no_roofline:
        # one instrumented DDIM step (all of this rank's UNet calls) + the decode of 4 frames.  EVERY rank runs the step
        # (denoise contains the per-timestep all-gather); only rank 0 records and reports.
        import contextlib
        prof = ops.GemmProfile() if rank == 0 else None
        opprof = ops.OpProfile() if rank == 0 else None
        n_dec = min(F, 4)
        with (prof if prof is not None else contextlib.nullcontext()), \
                (opprof if opprof is not None else contextlib.nullcontext()):
            lat = inp["latents"].clone()
            # the instrumented leg runs 1 of the clip's DDIM steps and decodes n_dec of its F frames: every launch carries
            # the number of times it runs per clip (ops.PROFILE_WEIGHT), the choice of the dominant kernel is over clip time
            ops.PROFILE_WEIGHT[0] = float(args.ddim_steps)
            pipe.denoise(lat, kps_tokens, audio, timesteps[:1], windows, 3.5)
            ops.PROFILE_WEIGHT[0] = F / float(n_dec)
            if rank == 0:
                pipe.vae.decode_video(lat[:, :, :n_dec].contiguous(), chunk=4)
            ops.PROFILE_WEIGHT[0] = 1.0
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        summ = prof.summary()
        syms = prof.by_symbol()
        if args.gemm_shapes:
            with open(args.gemm_shapes, "w") as fsh:
                for (m_, n_, k_, kern), (cnt, sec, fl) in prof.by_shape().items():
                    fsh.write(f"{m_:8d} {n_:6d} {k_:6d} x{cnt:4d} {1e6 * sec / cnt:9.1f} us {1e3 * sec:8.2f} ms "
                              f"{fl / sec / 1e12:7.1f} TF/s  {kern}\n")
        tot_s = sum(v["seconds"] for v in summ.values())
        tot_f = sum(v["flops"] for v in summ.values())
        # ---- every profiled launch, GEMM or not, keyed by the kernel instantiation as rocprofv3 prints it (MFMA kernels) or
        # by the wrapper name (HBM-bound kernels); clip-weighted HIP-event time decides which ONE is the dominant kernel
        osyms = opprof.by_symbol()
        launches = [dict(symbol=r[5], seconds=r[0].elapsed_time(r[1]) * 1e-3, weight=r[6], flops=r[2],
                         bytes=prof.bytes_of.get(id(r[0]), 0.0)) for r in prof.records]
        launches += [dict(symbol=r[5], seconds=r[2].elapsed_time(r[3]) * 1e-3, weight=r[6], flops=r[4], bytes=r[1])
                     for r in opprof.records]
        ranking, allk, clip_s, clip_total = rank_kernels(launches)
        dom_sym = ranking[0]["kernel"]
        dom = (dom_sym, allk[dom_sym])
        ach, peak, unit, bound = kernel_rate(dom[1])
        traffic, traffic_src = _pmc_traffic(dom[0])
        rp = _rocprof_launch_avg(dom[0])
        if rp is not None:
            # this run's algorithmic work per launch of the instantiation / the profiler's duration of the same launches
            work = dom[1]["flops"] / 1e12 if bound == "mfma" else dom[1]["bytes"] / 1e9
            rp["achieved"] = work / dom[1]["launches"] / (rp["avg_launch_us"] * 1e-6)
            rp["frac"] = rp["achieved"] / peak
            rp["note"] = ("committed rocprofv3 --kernel-trace --stats summary of this command with this library build "
                          "(another box); durations without the HIP-event overhead")
        # the family the instantiation belongs to (all instantiations of the same kernel template; GEMM: and epilogue kind)
        fam_key = dom[0].split(",")[0] if dom[0].startswith("gemm") else dom[0].split("<")[0]
        fam = [v for k, v in allk.items() if k == dom[0] or k.startswith(fam_key + ",") or k.startswith(fam_key + "<")]
        fam_fl, fam_s, fam_n = sum(v["flops"] for v in fam), sum(v["seconds"] for v in fam), sum(v["launches"] for v in fam)
        # the dominant GEMM instantiation (what `roofline.kernel` was before round 5, kept for continuity)
        gsym = max((k for k in clip_s if k in syms), key=lambda k: clip_s[k])
        gv = syms[gsym]
        # HBM-bound kernels: achieved GB/s = algorithmic bytes / HIP-event time against 8 TB/s
        hbm = {}
        for name, v in opprof.summary().items():
            hbm[name] = {"launches": v["launches"], "avg_us": 1e6 * v["seconds"] / v["launches"],
                         "algorithmic_mb_per_launch": v["bytes"] / v["launches"] / 1e6,
                         "gbs": v["bytes"] / v["seconds"] / 1e9, "frac": v["bytes"] / v["seconds"] / 1e9 / HBM_PEAK_GBS}
            if v["flops"] > 0:
                hbm[name]["tflops"] = v["flops"] / v["seconds"] / 1e12
        short = [(shape, t) for shape, t in prof.by_shape().items() if shape[2] <= 640 and shape[0] >= 16384]
        if short:
            sb = sum(prof.shape_bytes.get(shape, 0.0) for shape, _ in short)
            ss = sum(t[1] for _, t in short)
            sl = sum(t[0] for _, t in short)
            hbm["gemm K<=640 (M>=16384)"] = {"launches": sl, "avg_us": 1e6 * ss / sl, "algorithmic_mb_per_launch": sb / sl / 1e6,
                                            "gbs": sb / ss / 1e9, "frac": sb / ss / 1e9 / HBM_PEAK_GBS,
                                            "tflops": sum(t[2] for _, t in short) / ss / 1e12}
        per_kernel = {k: {"launches": v["launches"], "avg_us": 1e6 * v["seconds"] / v["launches"],
                          "tflops": v["flops"] / v["seconds"] / 1e12,
                          # fp8 launches are priced against the dense fp8 MFMA peak (K incl. the zero padding)
                          "frac_of_peak": v["flops"] / v["seconds"] / 1e12 /
                          (PEAK_FP8_TFLOPS if "fp8" in k else PEAK_BF16_TFLOPS)} for k, v in summ.items()}
        for k, v in osyms.items():            # the non-GEMM MFMA kernels: attention, the one-launch blocks
            if v["flops"] > 0:
                per_kernel[k] = {"launches": v["launches"], "avg_us": 1e6 * v["seconds"] / v["launches"],
                                 "tflops": v["flops"] / v["seconds"] / 1e12,
                                 "frac_of_peak": v["flops"] / v["seconds"] / 1e12 / PEAK_BF16_TFLOPS, "wrapper": v["name"]}
        mf = [v for v in allk.values() if v["flops"] > 0]
        result["roofline"] = {
            "bound": bound, "kernel": dom[0], "achieved": ach, "peak": peak, "unit": unit,
            "frac": ach / peak, "frac_source": "HIP events of this run (launch stream)", "rocprof": rp,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_flop_per_launch": dom[1]["flops"] / max(dom[1]["launches"], 1),
            "algorithmic_bytes_per_launch": dom[1].get("bytes", 0.0) / max(dom[1]["launches"], 1),
            "avg_launch_us": 1e6 * dom[1]["seconds"] / dom[1]["launches"], "launches": dom[1]["launches"],
            "share_of_clip_kernel_time": clip_s[dom_sym] / clip_total,
            "selection": "most clip-weighted HIP-event time over ALL profiled kernels (vx_gemm, attention, fused blocks, "
                         "HBM-bound kernels); see `ranking`",
            "ranking": ranking,
            "family": {"kernels": fam_key + ("<...>" if "<" not in fam_key else ", ...>"), "launches": fam_n,
                       "avg_launch_us": 1e6 * fam_s / fam_n, "achieved": (fam_fl / fam_s / 1e12) if fam_fl else None,
                       "frac": (fam_fl / fam_s / 1e12 / PEAK_BF16_TFLOPS) if fam_fl else None},
            "dominant_gemm": {"kernel": gsym, "achieved": gv["flops"] / gv["seconds"] / 1e12,
                              "frac": gv["flops"] / gv["seconds"] / 1e12 / PEAK_BF16_TFLOPS,
                              "avg_launch_us": 1e6 * gv["seconds"] / gv["launches"], "launches": gv["launches"],
                              "share_of_clip_kernel_time": clip_s[gsym] / clip_total},
            "all_gemm_tflops": tot_f / tot_s / 1e12,
            "all_mfma_tflops": sum(v["flops"] for v in mf) / sum(v["seconds"] for v in mf) / 1e12,
            "per_kernel": per_kernel,
            "per_instantiation": {k: {"launches": v["launches"], "avg_us": 1e6 * v["seconds"] / v["launches"],
                                      "tflops": v["flops"] / v["seconds"] / 1e12} for k, v in allk.items()
                                  if v["flops"] > 0},
            "hbm_kernels": {"peak_gbs": HBM_PEAK_GBS, "note": "algorithmic bytes / HIP-event time of this run; "
                            "~6300 GB/s is what a streaming copy reaches on this part", "kernels": hbm},
            "whole_path": {"tflop_per_frame": fpf, "achieved": fps * fpf / world, "frac": fps * fpf / world / PEAK_BF16_TFLOPS,
                           "note": "fps x algorithmic TFLOP/frame (SURVEY.md 8d) per GPU / 2.5 PFLOP/s"},
        }
    if world > 1 and not args.no_same_clip_1gpu:
        # the same clip on ONE GPU (rank 0 alone, sharding off) so that the multi-GPU speed-up compares like with like
        if rank == 0:
            from v_express_amd.distributed import DistContext
            saved = pipe.dist
            pipe.dist = DistContext()
            if args.warmup:
                one_clip()            # the single-GPU pass has its own launch geometries / scratch buffers: warm them too
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one_clip()
            torch.cuda.synchronize()
            t1 = time.perf_counter() - t0
            pipe.dist = saved
            result["same_clip_1gpu_fps"] = F / t1
            result["same_clip_1gpu_warmed"] = bool(args.warmup)
            result["speedup_vs_1gpu_same_clip"] = fps / (F / t1)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.size, 30)
        if result["cpu_baseline"]["value"]:
            result["speedup_vs_cpu"] = fps / result["cpu_baseline"]["value"]
    if rank == 0:
        # ONE compact line for the driver (< LINE_LIMIT bytes; BENCH_r05.json could not be parsed at 23 KB); every table
        # (ranking, per-kernel, per-instantiation, HBM kernels, block paths, the CPU leg's probes) goes to the side file
        detail_path = write_detail(result, args.detail)
        print(json.dumps(compact_line(result, detail_path)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
