"""Sample GPU power and shader clock while a command runs (is the d = 40 attention at the power cap?).

    python tools/power_sample.py [--hz 20] [--tag NAME] -- <command ...>

Sources, first one that works: the amdgpu hwmon files in sysfs (power1_average / power1_input in microwatts,
freq1_input in Hz; ~0.1 ms per read), then `amd-smi metric --power --clock --json`, then `rocm-smi --showpower
--showclocks --json` (both ~100-300 ms per call, so the effective rate drops).  Prints one JSON line: samples taken
while the command ran, min / median / max power and sclk, the power cap, and the idle values before the command.
Measurement tool only - nothing in the product path imports it."""
import glob
import json
import os
import statistics
import subprocess
import sys
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def hip_pci_bus_id(device=0):
    """PCI bus id ('0000:c1:00.0') of HIP device `device` - the host's sysfs lists every GPU of the node, the process
    sees one."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, device) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def find_hwmon():
    bus = hip_pci_bus_id()
    cands = sorted(glob.glob(f"/sys/bus/pci/devices/{bus}/hwmon/hwmon*")) if bus else []
    cands += sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    for d in cands:
        if _read(os.path.join(d, "power1_average")) or _read(os.path.join(d, "power1_input")):
            return d
    return None


class Sysfs:
    name = "sysfs-hwmon"

    def __init__(self, d):
        self.d = d
        self.pfile = os.path.join(d, "power1_average" if _read(os.path.join(d, "power1_average")) else "power1_input")
        self.ffile = os.path.join(d, "freq1_input")
        cap = _read(os.path.join(d, "power1_cap"))
        self.cap = float(cap) * 1e-6 if cap else None
        self.dev = os.path.dirname(os.path.dirname(d))
        self.name = f"sysfs-hwmon {os.path.realpath(self.dev)}"

    def sample(self):
        p, f = _read(self.pfile), _read(self.ffile)
        sclk = float(f) * 1e-6 if f else None
        if sclk is None:
            cur = [ln for ln in (_read(os.path.join(self.dev, "pp_dpm_sclk")) or "").splitlines() if ln.endswith("*")]
            if cur:
                sclk = float(cur[0].split()[1].lower().replace("mhz", ""))
        return (float(p) * 1e-6 if p else None, sclk)


class Cli:
    def __init__(self, which):
        self.name = which
        self.cap = None

    def sample(self):
        try:
            if self.name == "amd-smi":
                out = subprocess.run(["amd-smi", "metric", "--power", "--clock", "--json"], capture_output=True,
                                     text=True, timeout=5).stdout
                d = json.loads(out)
                d = d[0] if isinstance(d, list) else (d.get("gpu_data") or [d])[0]
                pw = d.get("power", {})
                p = pw.get("socket_power", pw.get("current_power", {}))
                p = p.get("value") if isinstance(p, dict) else p
                clk = d.get("clock", {})
                g = clk.get("gfx_0", clk.get("gfx", {}))
                s = g.get("clk", {}) if isinstance(g, dict) else {}
                s = s.get("value") if isinstance(s, dict) else s
                return (float(p) if p not in (None, "N/A") else None, float(s) if s not in (None, "N/A") else None)
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                                 timeout=5).stdout
            d = json.loads(out)
            c = d[sorted(d)[0]]
            p = next((float(v) for k, v in c.items() if "Power (W)" in k and "Average" in k or "Socket" in k), None)
            s = next((float(v.strip("()Mhz ")) for k, v in c.items() if k.startswith("sclk clock speed")), None)
            return (p, s)
        except Exception:
            return (None, None)


def pick_source():
    d = find_hwmon()
    if d:
        return Sysfs(d)
    for w in ("amd-smi", "rocm-smi"):
        s = Cli(w)
        if any(v is not None for v in s.sample()):
            return s
    return None


def summarize(vals):
    vals = [v for v in vals if v is not None]
    if not vals:
        return None
    return dict(min=round(min(vals), 1), median=round(statistics.median(vals), 1), max=round(max(vals), 1), n=len(vals))


def main():
    a = sys.argv[1:]
    hz, tag = 20.0, ""
    while a and a[0] != "--":
        if a[0] == "--hz":
            hz = float(a[1]); a = a[2:]
        elif a[0] == "--tag":
            tag = a[1]; a = a[2:]
        else:
            raise SystemExit(__doc__)
    cmd = a[1:]
    if not cmd:
        raise SystemExit(__doc__)
    src = pick_source()
    if src is None:
        print(json.dumps(dict(tag=tag, error="no power/clock source (sysfs hwmon, amd-smi, rocm-smi) works here")))
        return subprocess.call(cmd)
    def all_cards():
        """(device, W, MHz) of every amdgpu hwmon the host exposes - shows which card actually follows the load"""
        out = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            p = _read(os.path.join(d, "power1_average")) or _read(os.path.join(d, "power1_input"))
            f = _read(os.path.join(d, "freq1_input"))
            out.append((os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(d)))),
                        round(float(p) * 1e-6, 1) if p else None, round(float(f) * 1e-6) if f else None))
        return out

    idle = [src.sample() for _ in range(5)]
    cards_idle, cards_busy = all_cards(), None
    t0 = time.time()
    proc = subprocess.Popen(cmd)
    samples = []
    while proc.poll() is None:
        samples.append((round(time.time() - t0, 3),) + src.sample())
        if cards_busy is None and time.time() - t0 > 1.5:
            cards_busy = all_cards()
        time.sleep(1.0 / hz)
    # the command's launch phase (dlopen, fills, reference checks) is not the loop: report the busiest half too
    pw = [s[1] for s in samples]
    top = sorted([p for p in pw if p is not None])[len(pw) // 2:]
    print(json.dumps(dict(tag=tag, source=src.name, hz_requested=hz, seconds=round(time.time() - t0, 2),
                          hip_device_pci=hip_pci_bus_id(), cards_idle=cards_idle, cards_at_1p5s=cards_busy,
                          power_cap_w=src.cap, idle=dict(power_w=summarize([s[0] for s in idle]),
                                                         sclk_mhz=summarize([s[1] for s in idle])),
                          power_w=summarize(pw), power_w_busiest_half=summarize(top),
                          sclk_mhz=summarize([s[2] for s in samples]),
                          sclk_mhz_when_power_above_median=summarize(
                              [s[2] for s in samples if s[1] is not None and top and s[1] >= top[0]]),
                          trace=samples[:: max(1, len(samples) // 60)])))
    return proc.returncode


if __name__ == "__main__":
    raise SystemExit(main())
