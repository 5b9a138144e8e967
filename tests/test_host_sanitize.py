"""CPU: the host mirror (nodey-audio-editor_amd/host: fibers, bounded streams, runner, processors' host logic, cadence rules) and its self-test
under AddressSanitizer + UndefinedBehaviorSanitizer, `selftest cpu` mode (no device call is made; GPU sanitizers are not available on the
pool).  The fibers switch stacks with swapcontext, which ASan only warns about; any finding or leak fails the run."""
import concurrent.futures
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nodey-audio-editor_amd", "host")
SRC = ["infra/fiber.cpp", "infra/runner.cpp", "processor/audio-stream.cpp", "processor/audio-vol.cpp", "processor/audio-mix.cpp",
       "processor/audio-velocity.cpp", "processor/draw-headless.cpp", "register.cpp"]
SAN = ["-O1", "-g", "-std=c++20", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-ffp-contract=off"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_mirror_is_clean_under_asan_ubsan(tmp_path):
    for d in (os.path.join(ROOT, "nodey-audio-editor_amd"), os.path.join(ROOT, "oracle")):      # the libraries the self-test links (not instrumented)
        r = subprocess.run(["make", "-C", d, "-j4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    inc = ["-I" + HOST, "-I" + os.path.join(ROOT, "include")]

    def cc(src):
        obj = str(tmp_path / (os.path.basename(src)[:-4] + ".o"))
        r = subprocess.run(["g++", *SAN, "-fPIC", *inc, "-c", os.path.join(HOST, src), "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return obj

    with concurrent.futures.ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, SRC))
    exe = str(tmp_path / "selftest_san")
    r = subprocess.run(["g++", *SAN, "-pthread", *inc, os.path.join(ROOT, "tests", "host", "selftest.cpp"), "-o", exe, *objs,
                        "-L" + os.path.join(ROOT, "nodey-audio-editor_amd"), "-lnae_gpu", "-L" + os.path.join(ROOT, "oracle"), "-lnae_oracle",
                        "-Wl,-rpath," + os.path.join(ROOT, "nodey-audio-editor_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                        "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=280, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "SELFTEST OK cpu" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
