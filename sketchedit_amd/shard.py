"""Batch sharding of the inference path across the GPUs of one node (one process per GPU).

Images are independent (no batch statistics; attention, K-normalisation and pooling are per sample --
SURVEY.md 8e), so rank r runs rows [lo, hi) of the global batch with replicated weights and no exchange
during the forward.  The only collective is ONE all-gather of the packed outputs (B/n, 4, H, W) -- planes 0-2 the
composite, plane 3 the soft mask (SE_FLAG_PACKED_OUT lets the last kernel write that layout directly); on the GPU box
the backend is nccl = RCCL over xGMI, in the CPU tests it is gloo.
"""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Contiguous, balanced shard [lo, hi) of n items for `rank` of `world`."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_batch(local, group=None):
    """All-gather equally sized batch shards (dim 0) from every rank, in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    else:
        dist.all_gather(list(out.chunk(world, 0)), local.contiguous(), group=group)
    return out


def global_mode(n, H, W):
    """The library's execution mode for a GLOBAL batch of n images: every rank runs its shard in the mode the unsharded
    call would run in, so an image's result does not depend on the world size (the library guarantees bit-identical
    results across batch positions and ranks only within one mode, include/sketchedit_hip.h)."""
    from ._lib import Engine
    return Engine.is_low_latency(n, H, W)


def _accepts(forward, name):
    import inspect
    try:
        return name in inspect.signature(forward).parameters
    except (TypeError, ValueError):
        return False


def sharded_inference(forward, image, sketch, group=None):
    """Run `forward(image_shard, sketch_shard[, low_latency=...])` on this rank's rows of the global batch and return
    the gathered global (composed, mask).  `forward` returns either the packed (b,4,H,W) tensor
    (Engine.inference_packed) or a (composed, mask) pair, which is packed here; one collective either way.  When
    `forward` has a `low_latency` parameter it receives the mode of the GLOBAL batch (`global_mode`), not the one its
    shard's size would select.  Requires batch % world == 0."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = image.shape[0]
    if n % world:
        raise ValueError("global batch %d is not a multiple of the world size %d" % (n, world))
    lo, hi = shard_range(n, world, rank)
    kw = {"low_latency": global_mode(n, image.shape[2], image.shape[3])} if _accepts(forward, "low_latency") else {}
    out = forward(image[lo:hi].contiguous(), sketch[lo:hi].contiguous(), **kw)
    packed = out if isinstance(out, torch.Tensor) else torch.cat([out[0], out[1]], 1)
    full = gather_batch(packed, group)
    c = full.shape[1] - 1
    return full[:, :c], full[:, c:]
