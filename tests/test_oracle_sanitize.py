"""CPU: the oracle sources under AddressSanitizer + UndefinedBehaviorSanitizer (the FIFO-based WSOLA chain in block and
streaming use with interleaved receives, and the vocoder oracle), over every stage order and a few odd sizes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracles_are_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "orc_sanitize")
    src = [os.path.join(ROOT, "tests", "sanitize", "orc_sanitize.c")] + [os.path.join(ROOT, "oracle", f) for f in ("orc_wsola.c", "orc_stft.c", "orc_nodes.c")]
    r = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                        "-I" + os.path.join(ROOT, "oracle"), "-I" + os.path.join(ROOT, "include"), *src, "-lm", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("sr ")]
    assert len(lines) == 30
    for l in lines:                                   # block and streaming deliver the same number of frames
        a, b = l.split("block ")[1].split(" stream ")
        assert int(a) == int(b), l
