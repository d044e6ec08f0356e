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
