"""The reference's unmodified instant_nsr modules over the drop-in shims (see
tests/dropin_seam_check.py).  Runs where /root/reference exists (the build container)."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference/2_charactor_reconstructor/instant_nsr"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference snapshot is only in the build container")
def test_reference_modules_over_shims_reproduce_reference_fixture():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "dropin_seam_check.py")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "drop-in seam: reference modules over the shims reproduce" in r.stdout
