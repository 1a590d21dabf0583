"""The N>1 path on CPU: world_size-2 gloo run of the detections all-gather (bench.py uses the same
function over RCCL), plus batch sharding.  Rendezvous on 127.0.0.1."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from masklab_hip import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        images = torch.arange(4 * 2 * 2 * 3, dtype=torch.float32).reshape(4, 2, 2, 3)
        mine = parallel.shard_batch(images, rank, world)
        assert mine.shape[0] == 2 and torch.equal(mine, images[rank * 2:(rank + 1) * 2])
        cap = 5
        prop = torch.full((2, cap, 6), -1.0)
        counts = torch.tensor([rank + 1, 2 * rank], dtype=torch.int32)
        for b in range(2):
            for i in range(int(counts[b])):
                prop[b, i] = torch.tensor([rank, b, i, 1.0, float(i % 5), 0.9 - 0.1 * i])
        allp, allc = parallel.all_gather_detections(prop, counts)
        # the hot path hands over the record the detection kernel wrote (one [B, cap*6+1] tensor) and gets views back
        payload = parallel.pack_payload(prop, counts)
        g = parallel.AsyncDetectionGather("cpu")
        vp, vc = g.wait(g.launch(payload, cap))
        assert torch.equal(vp, allp) and torch.equal(vc, allc) and vc.dtype == torch.int32
        p3, c3 = parallel.all_gather_detections(prop, counts, payload=payload)
        assert torch.equal(p3, allp) and torch.equal(c3, allc)
        # bench.py's post-run assertion: B x world images, this rank's rows at offset rank (else a non-zero exit)
        assert parallel.check_merged(vp, vc, prop, counts, rank, world) is None
        assert "offset" in parallel.check_merged(vp, vc, prop, counts, 1 - rank, world)        # someone else's slot
        assert "expected" in parallel.check_merged(vp[:2], vc[:2], prop, counts, rank, world)  # unmerged record
        seg = torch.full((2, 3, 3, 3), float(rank))
        (allseg,) = parallel.all_gather_outputs([seg])
        q.put((rank, allp.numpy(), allc.numpy(), allseg.numpy()))
    finally:
        dist.destroy_process_group()


def test_all_gather_detections_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, allp, allc, allseg = q.get(timeout=120)
        results[rank] = (allp, allc, allseg)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    p0, c0, s0 = results[0]
    p1, c1, s1 = results[1]
    np.testing.assert_array_equal(p0, p1); np.testing.assert_array_equal(c0, c1); np.testing.assert_array_equal(s0, s1)
    assert p0.shape == (4, 5, 6)
    np.testing.assert_array_equal(c0, [1, 0, 2, 2])                 # rank r's images at offset r (Concatenate axis 0)
    assert p0[0, 0, 0] == 0 and p0[2, 0, 0] == 1 and np.all(p0[1] == -1)
    np.testing.assert_array_equal(s0[:, 0, 0, 0], [0, 0, 1, 1])


def test_single_process_passthrough_and_shard_errors():
    from masklab_hip import parallel
    p, c = torch.zeros(2, 3, 6), torch.zeros(2, dtype=torch.int32)
    a, b = parallel.all_gather_detections(p, c)
    assert a is p and b is c
    pay = parallel.pack_payload(torch.arange(36.0).reshape(2, 3, 6), torch.tensor([3, 1], dtype=torch.int32))
    p2, c2 = parallel.unpack_payload(pay, 3)
    assert pay.shape == (2, 19) and p2.data_ptr() == pay.data_ptr() and c2.tolist() == [3, 1]
    assert torch.equal(p2, torch.arange(36.0).reshape(2, 3, 6))
    with pytest.raises(ValueError):
        parallel.shard_batch(torch.zeros(5, 2, 2, 3), 0, 2)
    with pytest.raises(ValueError):
        parallel.shard_batch(torch.zeros(66, 1, 1, 3), 0, 2)         # 33 per GPU > MoldBatch limit


def test_bench_sets_dmabuf_ipc_in_every_rank():
    """bench.py must export HSA_ENABLE_IPC_MODE_LEGACY=0 in the rank process itself (the driver launches the ranks with
    torch.distributed.run, not through bench.py's own spawner): main() sets it before anything imports torch.cuda."""
    import ast
    import pathlib
    src = pathlib.Path(__file__).resolve().parents[1].joinpath("bench.py").read_text()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    first_calls = [ast.unparse(n) for n in main.body[:4]]
    assert any("HSA_ENABLE_IPC_MODE_LEGACY" in c and "setdefault" in c for c in first_calls), first_calls
    spawn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "spawn_ranks")
    assert "HSA_ENABLE_IPC_MODE_LEGACY" not in ast.unparse(spawn)        # one place only: no launch-dependent difference
