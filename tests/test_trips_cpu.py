"""vk::count_trips (vokselis_amd/csrc/vk_trips.hpp) -- the exact trip count of `for (t = t0; t < t1; t += dt)`
(raycast_naive.wgsl:101) that every march kernel carries instead of t -- host build against the loop itself."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_count_trips_matches_the_loop(tmp_path):
    exe = str(tmp_path / "trips_fuzz")
    # -ffp-contract=off as the kernels: the function's additions must stay additions
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "vokselis_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "trips_fuzz.cpp")], check=True)
    for seed in ("0x9E3779B97F4A7C15", "88172645463325252"):
        r = subprocess.run([exe, "1000000", seed], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:]
        assert r.stdout.strip().endswith("bad 0 of 1000000")
