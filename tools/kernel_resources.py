"""Per-kernel register / spill / LDS figures of a HIP source compiled for gfx950 (no GPU needed):
    python tools/kernel_resources.py v-express_amd/csrc/vx_gemm_ring.hip [-D...]
Prints name, vgpr_count, spills, sgprs, static LDS for every kernel; used to keep the hot kernels spill-free."""
import re
import subprocess
import sys
import tempfile


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
               "-Wno-inline-asm", "-S", "--cuda-device-only", src, "-o", f.name] + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        text = open(f.name).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text, re.S):
        name, body, vg, sp = m.group(1), m.group(2), int(m.group(3)), int(m.group(4))
        sg = re.search(r"\.sgpr_count:\s+(\d+)", body)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("(vx_gemm_params)", "")
        print(f"{vg:4d} vgpr {sp:4d} spill {int(sg.group(1)) if sg else -1:4d} sgpr  {dem[:150]}")


if __name__ == "__main__":
    main()
