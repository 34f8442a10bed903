#!/usr/bin/env python3
"""Key generation of the bootstrap leg's key set (N = 2^17, 65536 slots, level budget {4,4}: the relinearisation key + 63 rotation keys, 5-6 GB)
with the host samplers (default) and with the device samplers (FHE_HAL_DEVICE_SAMPLER=1), one bootstrap each to show the keys work.
usage (GPU box): python tools/keygen_probe.py   -> gpurun_out/r06_keygen_probe.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = f"""
import sys, json; sys.path.insert(0, {ROOT!r})
from openfhe_amd import boot_batch as bb
import os
prng = os.path.join({ROOT!r}, "tests", "hal", "_build", "libdetprng.so")
r = bb.run_rank(17, 1 << 16, 1, 1, 1, 0, prng, warmup=0, key_threads=8)
h = r.pop("handle")
st = h.member_stats()
print("RESULT", json.dumps({{"keygen_s": r["keygen_s"], "setup_s": r["setup_s"], "max_abs_error": r["max_abs_error"],
                             "device_sampler_launches": st.get("DeviceSampler", (0, 0, 0, 0))[0],
                             "keyswitchgen_device_ops": st.get("KeySwitchGenInternal", (0, 0, 0, 0))[0]}}))
h.close()
"""
out = {}
for name, env in (("host_samplers", {}), ("device_samplers", {"FHE_HAL_DEVICE_SAMPLER": "1"})):
    e = dict(os.environ, FHE_HIP_LIB=os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so"), FHE_HAL_REQUIRE_DEVICE="1", **env)
    p = subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True, timeout=1500)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    out[name] = json.loads(line[0][7:]) if line else {"error": (p.stdout + p.stderr)[-500:]}
    print(name, out[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_keygen_probe.json"), "w"), indent=1)
