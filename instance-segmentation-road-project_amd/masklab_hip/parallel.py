"""Multi-GPU data parallelism for the inference path: one process per GPU, batch sharded by
rank, weights replicated, no activation exchange.  The single collective is an RCCL all-gather
(over xGMI) of the FIXED-CAPACITY per-GPU detections, so that rank r's images land at offset r --
the same ordering as the reference's in-graph DP merge `Concatenate(axis=0)`
(reference engine/parallel.py:64-66 split, :92-107 merge).  Payload per GPU for B_local=8:
8 * (100*6 + 1) * 4 B = 19 KB: latency-bound, one call per batch.

The record that travels is written by the detection kernel itself (`ml_detection_proposal_f32`'s
gather_payload: per image the 6*cap floats of `proposed` followed by the count bit-cast to float), so the
hot path runs no torch arithmetic to build it; the gathered tensor is handed back as views.
`AsyncDetectionGather` issues the collective on its own HIP stream so that the next batch's backbone
overlaps it."""
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


def _gather0(t, group, async_op=False):
    """all_gather_into_tensor along axis 0 -> (out, work or None); a GPU tensor on the 'gloo' backend (CPU
    rehearsal of the N>1 path on a one-GPU box) is staged through host memory, RCCL ('nccl') takes it as is."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    stage = t.is_cuda and dist.get_backend(group) == "gloo"
    src = t.contiguous().cpu() if stage else t.contiguous()
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    work = dist.all_gather_into_tensor(out, src, group=group, async_op=async_op and not stage)
    return (out.to(t.device) if stage else out), (work if async_op and not stage else None)


def pack_payload(proposed, counts):
    """[B,cap,6] f32 + [B] i32 -> the [B, cap*6+1] record (for callers that do not have the kernel-written
    one: CPU tests, hand-made detections).  Plain data movement, not on the model's hot path."""
    B, cap, f = proposed.shape
    return torch.cat([proposed.reshape(B, cap * f),
                      counts.to(torch.int32).contiguous().view(torch.float32).reshape(B, 1)], dim=1)


def unpack_payload(payload, cap):
    """[B, cap*6+1] -> (proposed [B,cap,6], counts [B] int32) as VIEWS of `payload` (no copy, no kernel)."""
    proposed = payload[:, :cap * 6].view(-1, cap, 6)
    counts = payload[:, cap * 6].view(torch.int32)
    return proposed, counts


def all_gather_detections(proposed, counts, group=None, payload=None):
    """proposed [B_local,cap,6] f32 (-1 padded), counts [B_local] i32 -> ([B_global,cap,6], [B_global]).
    ONE collective per batch: the int32 counts ride as one extra (bit-cast) float column of the payload.
    `payload` = the record the detection kernel wrote (InferenceModel.last_detections['payload']); without it
    the record is assembled here.  Works with backend 'nccl' (= RCCL on ROCm) on GPU tensors and 'gloo' on
    CPU tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return proposed, counts
    cap = proposed.shape[1]
    if payload is None:
        payload = pack_payload(proposed, counts)
    out, _ = _gather0(payload, group)
    out_p, out_c = unpack_payload(out, cap)
    return out_p, out_c.to(counts.dtype) if out_c.dtype != counts.dtype else out_c


class AsyncDetectionGather:
    """The per-batch all-gather on a dedicated HIP stream: `launch()` after a forward has been enqueued makes
    the communication stream wait for that forward only, so the NEXT forward (enqueued on the compute stream
    right away) overlaps the collective; `wait(handle)` makes the calling stream wait for the gathered data
    and returns (proposed [B_global,cap,6], counts [B_global]) views."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.stream = None

    def launch(self, payload, cap, snapshot=False):
        """snapshot=True: `payload` is a buffer the NEXT forward rewrites in place (the outputs of a replayed hipGraph
        are graph-owned: InferenceModel.outputs_graph_owned) -- the record is copied on the compute stream first, so
        the collective never reads a buffer the following replay is already writing."""
        import torch.distributed as dist
        if dist.get_backend(self.group) == "gloo":          # CPU tests / one-GPU rehearsal: host-staged, synchronous
            out, _ = _gather0(payload, self.group)
            return (out, None, cap)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        if snapshot:
            payload = payload.clone()                       # 19 KB device copy, ordered before the next replay
        self.stream.wait_stream(main)                       # the forward that produced `payload`
        with torch.cuda.stream(self.stream):
            out, work = _gather0(payload, self.group, async_op=True)
        payload.record_stream(self.stream)                  # allocator: the comm stream still reads it
        return (out, work, cap)

    def wait(self, handle):
        out, work, cap = handle
        if work is not None:
            work.wait()                                     # current stream waits for the collective (no host block)
            out.record_stream(torch.cuda.current_stream(self.device))
        return unpack_payload(out, cap)


def check_merged(merged_rows, merged_counts, own_rows, own_counts, rank, world):
    """The merged record must be Concatenate(axis=0) of the ranks' records (reference engine/parallel.py:92-107):
    `world` x B_local images with this rank's rows at image offset rank * B_local.  -> None, or what is wrong."""
    B = int(own_rows.shape[0])
    if int(merged_rows.shape[0]) != B * world or int(merged_counts.shape[0]) != B * world:
        return f"merged batch holds {int(merged_rows.shape[0])} images, expected {B} x {world}"
    sl = slice(rank * B, (rank + 1) * B)
    if not torch.equal(merged_counts[sl].to(own_counts.device, own_counts.dtype), own_counts):
        return f"counts at offset {rank * B} are not rank {rank}'s"
    mine, got = own_rows, merged_rows[sl].to(own_rows.device)
    if not torch.equal(got.view(torch.int32), mine.contiguous().view(torch.int32)):      # bit patterns (-1 padding included)
        return f"rows at offset {rank * B} are not rank {rank}'s"
    return None


def all_gather_outputs(tensors, group=None):
    """Concatenate(axis=0) of arbitrary fixed-shape per-rank outputs (e.g. seg_pred)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(tensors)
    return [_gather0(t, group)[0] for t in tensors]
