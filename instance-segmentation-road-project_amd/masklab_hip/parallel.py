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


def _gather0(t, group):
    """all_gather_into_tensor along axis 0; a GPU tensor on the 'gloo' backend (CPU rehearsal of the N>1
    path on a one-GPU box) is staged through host memory, RCCL ('nccl') takes it as is."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    stage = t.is_cuda and dist.get_backend(group) == "gloo"
    src = t.contiguous().cpu() if stage else t.contiguous()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(t.device) if stage else out


def all_gather_detections(proposed, counts, group=None):
    """proposed [B_local,cap,6] f32 (-1 padded), counts [B_local] i32 -> ([B_global,cap,6], [B_global]).
    ONE collective per batch: the int32 counts ride as one extra (bit-cast) float column of the payload.
    Works with backend 'nccl' (= RCCL on ROCm) on GPU tensors and 'gloo' on CPU tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return proposed, counts
    B, cap, f = proposed.shape
    payload = torch.cat([proposed.reshape(B, cap * f),
                         counts.to(torch.int32).contiguous().view(torch.float32).reshape(B, 1)], dim=1)
    out = _gather0(payload, group)
    out_p = out[:, :cap * f].reshape(-1, cap, f).contiguous()
    out_c = out[:, cap * f].contiguous().view(torch.int32).to(counts.dtype)
    return out_p, out_c


def all_gather_outputs(tensors, group=None):
    """Concatenate(axis=0) of arbitrary fixed-shape per-rank outputs (e.g. seg_pred)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    return [_gather0(t, group) for t in tensors]
