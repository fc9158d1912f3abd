"""What the GPU box's HOST can actually run (the CPU leg of bench.py sized itself from the visible CPU count and 16 pinned
8-thread workers did not get through their quarter-size warm-up in 30 s, profiles/r06b): cgroup CPU quota, topology as sysfs
shows it, and the CPU leg at 1 / 4 / 16 workers with a long budget - per-worker stage times say whether the box scales.
    python tools/cpu_box_probe.py > gpurun_out/<tag>_cpu_box_probe.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cat(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def main():
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat",
              "/sys/devices/system/cpu/cpu0/topology/thread_siblings_list", "/sys/devices/system/cpu/cpu1/topology/thread_siblings_list",
              "/sys/devices/system/cpu/cpu0/topology/core_id", "/sys/devices/system/cpu/cpu128/topology/thread_siblings_list",
              "/sys/devices/system/cpu/online", "/proc/loadavg"):
        print(p, "=", cat(p).replace("\n", " | ")[:300])
    print(subprocess.run("lscpu | grep -E 'Model name|Thread|Core|Socket|NUMA|MHz|Hypervisor|Virtualization'", shell=True,
                         capture_output=True, text=True).stdout)
    for n in (1, 4, 16):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-leg", "--size", "512", "--cpu-budget", "75",
                            "--cpu-max-workers", str(n)], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONPATH=ROOT))
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not rows:
            print(n, "workers: no output", r.stderr[-300:])
            continue
        d = json.loads(rows[-1])
        ws = d["workers"]
        print(f"{n:2d} workers x {d['threads']} threads: weights {d['weights_s']:.1f} s, leg {d['leg_process_s']:.1f} s, killed {d['killed']}; "
              "warm / unet f=2 / vae frame per worker:",
              " ".join(f"{w.get('warm_s', float('nan')):.1f}/{w.get('unet_forward_s', float('nan')):.1f}/{w.get('vae_frame_s', float('nan')):.1f}"
                       for w in ws))
        print("   cpu.stat after:", cat("/sys/fs/cgroup/cpu.stat").replace("\n", " | ")[:300])


if __name__ == "__main__":
    main()
