"""CPU, world_size 2 over gloo: the N>1 path of bench.py / the sharding helpers.
Each rank decodes ITS block range (with the oracle standing in for the GPU -- there is no
GPU here), nothing is exchanged on the data path, and the gathered result equals the
single-process decode.  Also exercises the barrier + max-over-ranks timing reduction
bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_blocks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from datagen import values
        from fastlanes_amd.sharding import block_range, shard_mixed
        from oracle_lib import load_oracle
        o = load_oracle()
        # uniform width: u32 W=7, every rank regenerates only its slice of the packed column
        full = values("u32", n_blocks * 224, 11)
        start, count = block_range(n_blocks, world, rank)
        mine = o.batch("unpack", "u32", 7, full[start * 224:(start + count) * 224])
        # mixed widths 1..32 (BASELINE config 5 shape): width[b] = 1 + b % 32
        widths = (1 + np.arange(n_blocks) % 32).astype(np.uint8)
        s2, c2, byte0, nbytes = shard_mixed(widths, world, rank)
        assert (s2, c2) == (start, count)
        total_words = int(widths.astype(np.int64).sum()) * 32
        col = values("u32", total_words, 12)
        sl = col[byte0 // 4:(byte0 + nbytes) // 4]
        out2 = np.zeros(count * 1024, dtype=np.uint32)
        pos = 0
        for i in range(count):
            w = int(widths[start + i])
            out2[i * 1024:(i + 1) * 1024] = o.unpack("u32", w, sl[pos:pos + 32 * w])
            pos += 32 * w
        assert pos == sl.size
        # bookkeeping collectives only: counts, and bench.py's barrier + MAX-over-ranks time
        cnt = torch.tensor([count], dtype=torch.int64)
        dist.all_reduce(cnt)
        assert int(cnt) == n_blocks
        dist.barrier()
        t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert abs(float(t) - 0.010 * world) < 1e-12
        q.put((rank, start, count, mine, out2))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_block_range_sharding():
    from datagen import values
    from oracle_lib import load_oracle
    world, n_blocks = 2, 37  # odd: rank 0 holds the remainder block
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    o = load_oracle()
    assert [g[1:3] for g in got] == [(0, 19), (19, 18)]
    want = o.batch("unpack", "u32", 7, values("u32", n_blocks * 224, 11))
    assert np.array_equal(np.concatenate([g[3] for g in got]), want)
    widths = (1 + np.arange(n_blocks) % 32).astype(np.uint8)
    col = values("u32", int(widths.astype(np.int64).sum()) * 32, 12)
    want2, pos = [], 0
    for w in widths:
        want2.append(o.unpack("u32", int(w), col[pos:pos + 32 * int(w)]))
        pos += 32 * int(w)
    assert np.array_equal(np.concatenate([g[4] for g in got]), np.concatenate(want2))


def test_block_range_properties():
    from fastlanes_amd.sharding import block_range, packed_offsets, shard_mixed
    # BASELINE config 5: 10 B integers over 8 GPUs
    n = 9_765_625
    rs = [block_range(n, 8, r) for r in range(8)]
    assert rs[0] == (0, 1_220_704) and all(c == 1_220_703 for _, c in rs[1:])
    assert sum(c for _, c in rs) == n and all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(7))
    for n, ws in ((0, 3), (1, 8), (7, 8), (8, 8), (1000, 7)):
        rs = [block_range(n, ws, r) for r in range(ws)]
        assert sum(c for _, c in rs) == n and rs[0][0] == 0
        assert max(c for _, c in rs) - min(c for _, c in rs) <= 1
    widths = (1 + np.arange(9_765_625) % 32).astype(np.uint8)
    off, total = packed_offsets(widths)
    assert total == 20_624_988_800
    assert off[1] == 128 and off[32] == 128 * sum(range(1, 33))
    parts = [shard_mixed(widths, 8, r) for r in range(8)]
    assert sum(p[3] for p in parts) == total and parts[0][2] == 0
    assert all(parts[i][2] + parts[i][3] == parts[i + 1][2] for i in range(7))
