"""The host-side index arithmetic of the partition -- vokselis_amd/csrc/vk_hostmath.hpp: deal of tile positions over ranks, block -> pixel map,
compact records, the root's un-tile map, cull rectangle / silhouette hull / heaviest-first order -- under AddressSanitizer and
UndefinedBehaviorSanitizer on the CPU (tests/hostmath_fuzz.cpp replays whole batches: launch -> gather -> un-tile, 1..8 ranks)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostmath") / "hostmath_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", os.path.join(ROOT, "vokselis_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tests", "hostmath_fuzz.cpp")], check=True)
    return exe


@pytest.mark.parametrize("seed", ["88172645463325252", "0x9E3779B97F4A7C15", "20261004"])
def test_hostmath_under_sanitizers(fuzz_exe, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([fuzz_exe, "400", seed], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.strip().endswith("OK (400 cases)") and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_hostmath_header_is_what_the_library_compiles():
    """The kernels and the host take these maps from the header the fuzz includes -- not from copies of their own."""
    csrc = os.path.join(ROOT, "vokselis_amd", "csrc")
    common = open(os.path.join(csrc, "vk_common.hpp")).read()
    assert '#include "vk_hostmath.hpp"' in common and "deal_pos(uint32_t rank" not in common
    assert "batch_block_split(" in common and "tile_pixel(" in common and "compact_pixel_index(" in common
    post = open(os.path.join(csrc, "vk_post.hpp")).read()
    assert "untile_item(" in post and "gathered_pixel_index(" in post
    order = open(os.path.join(csrc, "vk_order.hip")).read()
    assert "vk::tile_order(" in order and "hull_separates" not in order.replace("vk_hostmath", "")
    batch = open(os.path.join(csrc, "vk_batch.hip")).read()
    assert "batch_table_bytes(" in batch and "batch_order_offset(" in batch
