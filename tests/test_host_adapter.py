"""The C++ host mirror of the reference's plugin interface (nodey-audio-editor_amd/host): infra::Processor,
Audio_stream, a fiber runner and the GPU processors under the reference's identifiers.  The C++ self-test
(tests/host/selftest.cpp) drives real graphs; this file builds and runs it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_TEST = os.path.join(ROOT, "tests", "host")


def build():
    for d in (os.path.join(ROOT, "nodey-audio-editor_amd"), os.path.join(ROOT, "oracle"),
              os.path.join(ROOT, "nodey-audio-editor_amd", "host"), HOST_TEST):
        r = subprocess.run(["make", "-C", d, "-j4"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def run(mode):
    build()
    r = subprocess.run([os.path.join(HOST_TEST, "selftest"), mode], capture_output=True, text=True, timeout=300)
    print(r.stdout[-4000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-4000:]
    assert f"SELFTEST OK {mode}" in r.stdout


def test_host_mirror_scheduler_registry_json():
    """no GPU: bounded streams + back-pressure, fiber round-robin, registry identifiers, JSON keys"""
    run("cpu")


@pytest.mark.gpu
def test_host_mirror_gpu_graphs():
    """GPU: volume / amix / pitch->spectrum with fan-out / velocity / bimix_v2 graphs vs the oracle; error capture"""
    run("gpu")
