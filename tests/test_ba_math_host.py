"""ba_oplus_fast (the series form of the SE3 update on the serial path of the pose-optimisation kernel) against ba_oplus (the closed forms of g2o's SE3Quat::exp,
types/se3quat.h:217-285), both compiled for the HOST from the very header the kernels include."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_series_form_of_the_se3_update_matches_the_closed_form(tmp_path):
    hdr = open(os.path.join(ROOT, "ccm_slam_amd", "csrc", "ba_math.h")).read()
    hdr = hdr.replace("#include <hip/hip_runtime.h>", "").replace("#define BA_HD __host__ __device__ __forceinline__", "#define BA_HD static inline")
    (tmp_path / "ba_math_nohip.h").write_text(hdr)
    exe = tmp_path / "oplus_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-D_GNU_SOURCE", "-I", str(tmp_path), "-o", str(exe), os.path.join(HERE, "host", "oplus_check.cpp"), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"quaternion component difference ([0-9.e+-]+), worst relative translation difference ([0-9.e+-]+)", out)
    assert m, out
    dq, dt = float(m.group(1)), float(m.group(2))
    # rotation: 1 ulp.  translation: the closed forms (1 - cos t)/t^2 and (t - sin t)/t^3 of the reference cancel for small angles (relative error ~1e-16 / t^2 resp. / t^3 in the
    # coefficient, times t resp. t^2 in the term), the series does not: the two differ by up to ~1e-11 of the translation, four orders below the pose tolerance of the parity tests
    assert dq < 1e-15 and dt < 1e-10, out
