"""Static hazard audit of the generated gfx950 code of the stage kernels (CPU test: hipcc -S cross-compiles).

Two failure classes that passed every functional test until a particular schedule exposed them (NOTES.md, rounds 1-2 section 4):
 * a VALU instruction overwriting the data registers of a 128-bit store in the very next issue slot;
 * an inline-asm LDS read whose destination registers are touched before the s_waitcnt that retires it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    src = os.path.join(ROOT, "roomnet_amd", "csrc", "rn_stage_rw.hip")
    deps = [src] + [os.path.join(ROOT, "roomnet_amd", "csrc", h) for h in ("rn_stage.h", "rn_fused.h", "rn_internal.h")]
    cache = os.path.join(ROOT, "build", "asm", "rw_audit.s")
    if os.path.exists(cache) and all(os.path.getmtime(cache) >= os.path.getmtime(d) for d in deps):
        return cache                                         # the listing takes ~2.5 minutes to generate
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    out = cache
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "roomnet_amd", "csrc"), "-DRN_BUILDING", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S",
           "--cuda-device-only", src, "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return str(out)


def _run(tool, listing, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), listing, "stage_rw_kernel", *args], capture_output=True,
                       text=True, check=True)
    return r.stdout


def test_no_valu_write_right_behind_a_wide_store(listing):
    out = _run("asm_store_audit.py", listing, "1")
    summary = [l for l in out.splitlines() if l.endswith("VALU overwrites flagged")]
    assert len(summary) >= 16, out[-2000:]                  # every instantiation (2 dtypes x 9 variants) was scanned
    assert all(l.split()[-4] == "0" for l in summary), "\n".join(summary)


def test_no_lds_read_result_touched_before_its_wait(listing):
    out = _run("asm_lds_audit.py", listing)
    summary = [l for l in out.splitlines() if l.endswith("flagged")]
    assert len(summary) >= 16
    assert all(l.rsplit(",", 1)[1].strip() == "0 flagged" for l in summary), "\n".join(summary)
