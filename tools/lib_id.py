"""Print the identity of the kernel sources (v_express_amd.lib.source_id) without importing torch or loading the library:
the value rocprofv3 summaries under profiles/ carry in their first line (`# lib_sha256=`)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v-express_amd", "csrc")
h = hashlib.sha256()
files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
               glob.glob(os.path.join(CSRC, "*.cpp")) + [os.path.join(CSRC, "Makefile"),
                                                          os.path.join(ROOT, "include", "vexpress_hip.h")])
for path in files:
    h.update(os.path.basename(path).encode())
    with open(path, "rb") as f:
        h.update(f.read())
print(h.hexdigest()[:16])
