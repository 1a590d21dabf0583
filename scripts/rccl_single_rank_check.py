#!/usr/bin/env python3
"""The detection all-gather through the REAL collective backend (RCCL, torch 'nccl') in the only form a one-GPU box
allows: a one-rank communicator (two ranks on one device are refused by RCCL).  Exercises what the gloo rehearsal
cannot: communicator creation with `device_id`, `all_gather_into_tensor` issued asynchronously on the dedicated
stream of `AsyncDetectionGather`, `work.wait()` on the compute stream, the allocator hand-over (`record_stream`).
Run by tests/test_gpu_bench.py::test_rccl_single_rank_async_gather; exits non-zero on any mismatch."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-segmentation-road-project_amd"))


def main():
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from masklab_hip import parallel
    cap, B = 100, 8
    gen = torch.Generator(device="cpu").manual_seed(7)
    proposed = torch.rand((B, cap, 6), generator=gen).to(dev)
    counts = torch.randint(0, cap + 1, (B,), generator=gen, dtype=torch.int32).to(dev)
    payload = parallel.pack_payload(proposed, counts)
    gather = parallel.AsyncDetectionGather(dev)
    handles = [gather.launch(payload, cap) for _ in range(3)]          # several collectives in flight, like the bench loop
    ok = True
    for h in handles:
        got_p, got_c = gather.wait(h)
        torch.cuda.synchronize(dev)
        ok &= bool(torch.equal(got_p, proposed)) and bool(torch.equal(got_c, counts))
    same_p, same_c = parallel.all_gather_detections(proposed, counts, payload=payload)   # world 1: returned as is
    ok &= same_p is proposed and same_c is counts
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single-rank gather:", "OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
