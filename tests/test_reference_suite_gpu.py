"""The reference's OWN tests, unmodified, on top of the MI355X library through the one-file shim of INTEGRATION.md
section 2 (tests/golden/ref/, staged by tests/golden/stage_reference_tests.py): its gather/scatter tests, its
barycentric-gradient tests, and the whole tetrahedra tracer suite -- the on-ray property on its bottle mesh
(tests/test_tetrahedra_tracer.py:62-209, triangles variant :50-130 of the other file), the find_tetrahedra known
answers (:270-344) and the einsum definition of interpolate_values forward + backward (:346-456)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
REF = ROOT / "tests" / "golden" / "ref"


def test_reference_suite_runs_unmodified_on_the_shim():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([str(REF), str(ROOT), env.get("PYTHONPATH", "")])
    files = ["tests/test_uint32.py", "tests/test_barycentrics.py", "tests/test_tetrahedra_tracer.py",
             "tests/test_tetrahedra_tracer_triangles.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", *files], cwd=REF, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-1500:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
