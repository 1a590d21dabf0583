"""Multi-GPU data parallelism for the inference path: one process per GPU, batch sharded by
rank, weights replicated, no activation exchange.  The single collective is an RCCL all-gather
(over xGMI) of the FIXED-CAPACITY per-GPU detections, so that rank r's images land at offset r --
the same ordering as the reference's in-graph DP merge `Concatenate(axis=0)`
(reference engine/parallel.py:64-66 split, :92-107 merge).  Payload per GPU for B_local=8:
8*100*6*4 B = 19 KB + 32 B of counts: latency-bound, one call per batch."""
import torch


def shard_batch(images, rank, world_size):
    """Contiguous split by rank (tf.split in reference parallel.py:64-66)."""
    B = images.shape[0]
    if B % world_size:
        raise ValueError(f"global batch {B} is not divisible by {world_size} ranks")
    per = B // world_size
    if per > 32:
        raise ValueError("at most 32 images per GPU per forward (MoldBatch, reference misc.py:275)")
    return images[rank * per:(rank + 1) * per]


def all_gather_detections(proposed, counts, group=None):
    """proposed [B_local,cap,6] f32 (-1 padded), counts [B_local] i32 -> ([B_global,cap,6], [B_global]).
    Works with backend 'nccl' (= RCCL on ROCm) on GPU tensors and 'gloo' on CPU tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return proposed, counts
    world = dist.get_world_size(group)
    out_p = torch.empty((world * proposed.shape[0],) + tuple(proposed.shape[1:]), dtype=proposed.dtype,
                        device=proposed.device)
    out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out_p, proposed.contiguous(), group=group)
    dist.all_gather_into_tensor(out_c, counts.contiguous(), group=group)
    return out_p, out_c


def all_gather_outputs(tensors, group=None):
    """Concatenate(axis=0) of arbitrary fixed-shape per-rank outputs (e.g. seg_pred)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    world = dist.get_world_size(group)
    outs = []
    for t in tensors:
        o = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(o, t.contiguous(), group=group)
        outs.append(o)
    return outs
