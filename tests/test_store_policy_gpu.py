"""GPU: the library's result stores are inline asm (`global_store ... sc1`, agent-scope write-through) carrying their own hazard wait
states (csrc/common.hpp st16 / st8) -- correct on this toolchain, silent on the next.  `make plainstores` builds the same library with
those stores written in C++; this test runs one deterministic battery over every kernel family of the path under both builds
(tools/store_policy_battery.py, two processes, FDM_LIB_PATH) and compares every output bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "face-diffusion-model_amd", "csrc")
PLAIN = os.path.join(ROOT, "face-diffusion-model_amd", "fdm_amd", "libfdm_hip_plain.so")


def _battery(lib_path):
    env = dict(os.environ)
    env.pop("FDM_LIB_PATH", None)
    if lib_path:
        env["FDM_LIB_PATH"] = lib_path
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_policy_battery.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.strip()]


def test_write_through_asm_stores_equal_plain_stores_over_the_whole_path():
    # __graft_entry__.build() builds both libraries; built here (the box has hipcc) when the plain-store one is missing or older than a source
    import glob
    newest_src = max(os.path.getmtime(f) for pat in ("*.hip", "*.hpp", "Makefile") for f in glob.glob(os.path.join(CSRC, pat)))
    if not os.path.exists(PLAIN) or os.path.getmtime(PLAIN) + 2.0 < newest_src:
        subprocess.check_call(["make", "-j16", "-C", CSRC, "plainstores"])
    a, b = _battery(None), _battery(PLAIN)
    assert len(a) == len(b) and len(a) >= 60, (len(a), len(b))
    diff = [(x, y) for x, y in zip(a, b) if x != y]
    assert not diff, f"{len(diff)} of {len(a)} outputs differ between the sc1 asm stores and plain stores, first: {diff[0]}"
    print(f"[store policy] {len(a)} outputs bit-identical between the write-through asm stores and the plain-store build")
