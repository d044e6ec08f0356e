"""World-size-2 test of the multi-GPU path on CPU (gloo): chunks are sharded in contiguous blocks, each
rank produces its records, one all-gather, every rank reconstructs the same word list as a single
process.  Device arithmetic is stood in by the oracle-backed engine double (host logic under test:
dist.shard_bounds / pack / all_gather_records / collate)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as td
from crisperwhisper_amd import audio, collate, dist, generation, synthetic as syn
from tests import helpers as Hh
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
td.init_process_group("gloo", rank=rank, world_size=world)
g, v, W, spec = Hh.tiny_setup()
x = syn.synth_audio(0, 70 * 16000, "mixed")
windows = audio.chunk_windows(len(x), 480000, 80000, 80000)
lo, hi = dist.shard_bounds(len(windows), world)[rank]
eng = Hh.OracleBackedEngine(g, v, W, spec)
recs = []
for i in range(lo, hi):
    s, n, st, _ = windows[i]
    _, nf = eng.mel([x[s:s + n]])
    out = generation.generate(eng, 1, nf, language="<|en|>", task="transcribe", max_new_tokens=40)
    k = len(out["token_timestamps"][0])
    recs.append(dist.pack_record(i, out["sequences"][0][:k], out["token_timestamps"][0], tuple(t / 16000 for t in st)))
recs = np.stack(recs) if recs else np.zeros((0, dist.REC_WORDS), np.int32)
shard = dist.Shard(rank, world)
allr = shard.all_gather_records(recs, max(h - l for l, h in dist.shard_bounds(len(windows), world)))
outs = []
for r in allr:
    _, toks, ts, stride = dist.unpack_record(r)
    outs.append({"tokens": toks, "token_timestamps": ts, "stride": stride})
text, words = collate.decode_asr(collate.Vocabulary.from_synthetic(v), outs)
json.dump({"text": text, "chunks": [{"text": w["text"], "timestamp": list(w["timestamp"])} for w in words]},
          open(os.path.join(sys.argv[2], f"rank{rank}.json"), "w"))
td.destroy_process_group()
'''


def test_two_rank_shard_and_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(WORKER)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen([sys.executable, script, ROOT, tmp], env=env))
        for p in procs:
            assert p.wait(timeout=600) == 0
        res = [json.load(open(os.path.join(tmp, f"rank{r}.json"))) for r in range(2)]
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_golden.json")))["mixed70_b2_n40"]
    assert res[0] == res[1]
    # chunks are independent: sharding 3 chunks as (2, 1) with batch 1 gives the reference words
    assert res[0]["text"] == gold["text"]
    assert [c["text"] for c in res[0]["chunks"]] == [c["text"] for c in gold["chunks"]]
    for a, b in zip(res[0]["chunks"], gold["chunks"]):
        assert np.allclose(a["timestamp"], b["timestamp"], atol=0.02 + 1e-9)


def test_bench_gpus_flag_builds_a_launcher_command():
    """`python bench.py --gpus N` outside a launcher re-runs itself as N ranks (bench.spawn_command): one process per GPU,
    rendezvous on 127.0.0.1, every other argument passed through."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    cmd = bench.spawn_command(argparse.Namespace(gpus=4), argv, 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv


@pytest.mark.gpu
def test_bench_gpus2_spawns_two_ranks_and_reports_them():
    """bench.py --gpus 2 with no launcher around it: two ranks (here sharing the one GPU of the box, collectives over gloo:
    two RCCL ranks cannot share a device) -- the line must say n_gpus == 2 and the communicator must have seen 2 ranks."""
    env = dict(os.environ, CW_DIST_BACKEND="gloo", PYTHONUNBUFFERED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--geometry", "tiny", "--batch", "2",
                        "--tokens", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--kernel-iters", "3"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["collective"]["ranks_seen"] == 2
    assert rec["collective"]["chunks_per_rank"] == [2, 2] and rec["config"]["parallelism"] == "chunk-dp2"
    assert rec["value"] > 0


@pytest.mark.gpu
def test_bench_gpus8_large_geometry_shards_the_long_recording_like_one_rank():
    """What the driver's 8-GPU run does, on the one device of the box: `bench.py --gpus 8` at the large-v3 geometry spawns eight
    ranks (sharing the GPU: 8 x 3.1 GB of weights; collectives over gloo because eight RCCL ranks cannot share a device).  The
    line must report 8 ranks seen by the communicator, the BASELINE configs[2] recording (600 s -> 30 chunks) sharded
    (4,4,4,4,4,4,3,3), and the merged word list -- every word and timestamp, by digest -- must be the one a single rank
    produces from the same recording (chunks are independent)."""
    env = dict(os.environ, CW_DIST_BACKEND="gloo", PYTHONUNBUFFERED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--geometry", "large-v3", "--batch", "4", "--tokens", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
              "--no-config3", "--kernel-iters", "3"]
    recs = {}
    for n in (1, 8):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common + (["--no-rccl"] if n == 1 else []),
                           env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        recs[n] = json.loads(lines[0])
    r8, r1 = recs[8], recs[1]
    assert r8["n_gpus"] == 8 and r8["collective"]["ranks_seen"] == 8 and r8["config"]["parallelism"] == "chunk-dp8"
    assert r8["collective"]["chunks_per_rank"] == [4] * 8
    assert r8["longform"]["chunk_shards"] == [4, 4, 4, 4, 4, 4, 3, 3] and r1["longform"]["chunk_shards"] == [30]
    assert r8["longform"]["words"] == r1["longform"]["words"] > 0
    assert r8["longform"]["output_sha1"] == r1["longform"]["output_sha1"]
