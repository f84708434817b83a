"""CPU: the sanitizer driver (tools/fuzz_host.cpp) over the host layer, the image / glTF readers and the
commit-time code of the backend (raytracer_amd/csrc/rt_commit.cpp), built by g++ with
AddressSanitizer + UndefinedBehaviorSanitizer.  A bounded pass: the self-test (every named scene through
submit + commit in all tree modes) and a few thousand mutated scene streams, PNGs and JPEGs.  The longer
passes and the ThreadSanitizer build are `make sanitize`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "fuzz_host_asan")
SEEDS = os.path.join(ROOT, "build", "fuzz_seeds")
STARTUP_TROUBLE = ("Shadow memory", "ASan runtime does not come first", "failed to allocate", "ReserveShadowMemoryRange")


@pytest.fixture(scope="module")
def driver():
    made = subprocess.run(["make", "-C", ROOT, "asan"], capture_output=True, text=True)
    if made.returncode != 0 or not os.path.exists(BIN):
        pytest.skip("the sanitizer build is not available here: " + made.stderr[-300:])
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_fuzz_seeds.py"), SEEDS], check=True, capture_output=True)
    return BIN


def run(driver, *args):
    r = subprocess.run([driver, *map(str, args)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    if r.returncode != 0 and any(s in r.stderr for s in STARTUP_TROUBLE):
        pytest.skip("the sanitizer runtime cannot start in this environment")
    assert r.returncode == 0, (r.stdout[-600:], r.stderr[-1500:])
    assert r.stdout.strip().endswith("ok"), r.stdout[-400:]
    return r.stdout


def seeds(ext):
    return sorted(os.path.join(SEEDS, f) for f in os.listdir(SEEDS) if f.endswith(ext))


def test_selftest_every_scene_every_tree_mode(driver):
    assert "14 scenes x 5 tree modes" in run(driver, "selftest")


def test_mutated_scene_streams_are_refused_or_committed(driver):
    run(driver, "streams", 3000, 123)


def test_mutated_images_are_refused_or_decoded(driver):
    run(driver, "png", 3000, 5, *seeds(".png"))
    run(driver, "jpeg", 3000, 6, *seeds(".jpg"))


def test_mutated_glb_files(driver):
    run(driver, "glb", 300, 7, *seeds(".glb"), os.path.join(ROOT, "tests", "golden", "room.glb"))
