"""Host-side behaviour of the multi-GPU entry that needs no GPU: rn_group_create resolves librccl with dlopen at its first
call, BEFORE it touches a device -- a host without the library must get RN_E_STATE and the documented message, not a
crash (the error string was built from two dlerror() calls, the second of which returns NULL)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
from roomnet_amd import _capi
from roomnet_amd.graph import build_graph
from roomnet_amd.tf_bundle import BundleReader
weights = BundleReader(os.path.join(%(root)r, "roomnet_amd", "final_model", "roomnet")).load_all()
try:
    _capi.Group(build_graph(6, 224), weights, devices=[0], dtype="bf16", max_batch_per_device=2)
except Exception as e:                      # RuntimeError from _check: the library reported, it did not crash
    print("RAISED", type(e).__name__, str(e))
else:
    print("CREATED")
"""


def test_group_create_without_librccl_reports_an_error_instead_of_crashing():
    env = dict(os.environ, ROOMNET_RCCL_LIB="/nonexistent/librccl_bogus.so.1")
    out = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, "child died (rc %d): %s" % (out.returncode, out.stderr[-2000:])
    line = [l for l in out.stdout.splitlines() if l.startswith(("RAISED", "CREATED"))][-1]
    assert line.startswith("RAISED"), line
    assert "dlopen(librccl) failed" in line and "librccl_bogus" in line, line
