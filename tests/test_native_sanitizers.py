"""Host-only half of the C ABI (word collator, FLAC decoder, beam-search bookkeeping) under AddressSanitizer +
UndefinedBehaviorSanitizer: tests/native/fuzz_host.cpp is compiled together with csrc/collate.cpp, csrc/flac.cpp and
csrc/beamhost.cpp (plain g++, no HIP) and fed mutated FLAC streams, noise, random token streams and random / hostile beam
candidates.  SURVEY.md section 5 lists a sanitizer build of the C ABI among the reference-side
auxiliaries; the container parsers are the part of this library that reads untrusted bytes."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _valid_flac_streams():
    from tests import flac_writer as FW
    rng = np.random.default_rng(5)
    out = []
    for bps, nch, stereo in ((16, 2, "mid_side"), (24, 1, "independent"), (8, 2, "left_side")):
        n_blocks = [576, 192, 1024]
        total = sum(n_blocks)
        t = np.arange(total)
        chans = [np.round((2 ** (bps - 2)) * np.sin(0.01 * (c + 1) * t) + rng.integers(-3, 4, total)).astype(np.int64) for c in range(nch)]
        frames, pos = [], 0
        kinds = [dict(kind="fixed", order=2, method=0, porder=2), dict(kind="verbatim"), dict(kind="fixed", order=1, method=1, porder=1)]
        for nb, kw in zip(n_blocks, kinds):
            frames.append(dict(n=nb, stereo=stereo if nch == 2 else "independent", plans=[dict(kw) for _ in range(nch)]))
            pos += nb
        out.append(FW.write_stream(chans, bps, 44100, frames))
    return out


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_c_abi_under_asan_ubsan(tmp_path):
    exe = tmp_path / "fuzz_host"
    src = [os.path.join(ROOT, "tests", "native", "fuzz_host.cpp"), os.path.join(ROOT, "crisperwhisper_amd", "csrc", "collate.cpp"),
           os.path.join(ROOT, "crisperwhisper_amd", "csrc", "flac.cpp"), os.path.join(ROOT, "crisperwhisper_amd", "csrc", "beamhost.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           *src, "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("toolchain without sanitizer runtimes: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    seeds = []
    for i, data in enumerate(_valid_flac_streams()):
        p = tmp_path / f"seed{i}.flac"
        p.write_bytes(data)
        seeds.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed in (1, 2):
        r = subprocess.run([str(exe), "4000", str(seed), *seeds], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
        assert "no sanitizer report" in r.stdout
