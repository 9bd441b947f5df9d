"""CPU: the N > 1 plumbing of bench.py (contiguous clip sharding + the one collective, an
all_gather of fixed-stride id records) rehearsed with world_size 2 and 3 on gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_shard_range_partitions_exactly():
    for world in (1, 2, 3, 4, 8):
        for total in (8, 32, 33, 256):
            spans = [bench.shard_range(r, world, total) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_range(3, 8, 256) == (96, 128)  # config 5: 256 clips -> 32 per GPU


def test_synthetic_mel_is_shard_invariant():
    a = bench.synthetic_mel(0, 4, (2, 8))
    b = np.concatenate([bench.synthetic_mel(0, 2, (2, 8)), bench.synthetic_mel(2, 4, (2, 8))])
    assert np.array_equal(a, b) and a.min() >= -1.0 and a.max() <= 1.5


def test_pack_records_layout():
    ids = np.arange(64, dtype=np.int64).reshape(2, 32)
    rec = bench.pack_records(ids, np.array([31, 7], np.int32))
    assert rec.shape == (2, 33) and rec[1, 32] == 7 and rec[0, 31] == 31 and rec.dtype == np.int64


@pytest.mark.parametrize("world", [2, 3])
def test_gather_over_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + os.getpid() % 400),
           os.path.join(ROOT, "bench.py"), "--dry-run-gloo", "--batch", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"dry_run": True, "world": world, "clips": 5 * world, "ok": True}
