"""Multi-GPU partitioning of one image: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPUs).

The reference is single-GPU.  Every pixel sample is independent given the scene
(shader/src/lib.rs:57-69 touches only its own accumulator element), so the work
splits two ways and the scene is simply replicated on every GPU:

  tile mode    the canvas is dealt to the ranks in interleaved strips of rows
               (rt_set_rows).  No collective in the data path; one gather of
               disjoint rows at the end.  Seeds depend on absolute (x, y), so the
               gathered image is byte-identical to a single-GPU render.
  sample mode  every rank renders the whole canvas for its own subset of the
               frames (rank g: frames g, g + G, ...), then ONE sum all-reduce of
               the accumulation buffer.  Needs the independent-seed variant
               (RT_OPT_SEED_MODE = 1) because the reference's seeds chain a
               pixel's samples through its accumulator.
"""
import numpy as np


def strip_rows(height, strip_h, rank, world):
    """Canvas rows owned by `rank`: strips of `strip_h` rows dealt round-robin."""
    if world <= 1:
        return np.arange(height)
    rows = []
    s = rank
    while s * strip_h < height:
        rows.extend(range(s * strip_h, min((s + 1) * strip_h, height)))
        s += world
    return np.asarray(rows, dtype=np.int64)


def gather_strips(local_image, strip_h, group=None, dst=0):
    """local_image: (H, W, C) tensor in which only this rank's rows are meaningful.
    Returns the assembled (H, W, C) image on rank `dst` (None elsewhere).  The
    exchange is one gather of each rank's own rows, padded to a common length: only
    `dst` receives anything (1/world of the image from every other rank)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local_image.clone()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    h = local_image.shape[0]
    if world == 1:
        return local_image.clone()
    mine = torch.as_tensor(strip_rows(h, strip_h, rank, world), device=local_image.device)
    n_max = max(len(strip_rows(h, strip_h, r, world)) for r in range(world))
    send = torch.zeros((n_max,) + tuple(local_image.shape[1:]), dtype=local_image.dtype, device=local_image.device)
    send[: len(mine)] = local_image[mine]
    out = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    if send.is_cuda and dist.get_backend(group) == "gloo":
        torch.cuda.current_stream(send.device).synchronize()  # gloo copies from a thread of its own
    dist.gather(send, out, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty_like(local_image)
    for r in range(world):
        rows = torch.as_tensor(strip_rows(h, strip_h, r, world), device=local_image.device)
        full[rows] = out[r][: len(rows)]
    return full


def reduce_samples(acc, group=None, dst=None):
    """Sum the per-rank accumulation buffers in place (sample mode): one collective over
    4*W*H floats (1080p: 33 MB, 4K: 133 MB) -- a reduce to rank `dst` when only that rank
    needs the image (the others' buffers are then left as they were), an all-reduce when
    `dst` is None.  The collective is ordered after the
    work on torch's current stream of the tensor's device: render on that stream (App.create(...,
    stream=torch.cuda.current_stream().cuda_stream)) or synchronize the context first."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if acc.is_cuda and dist.get_backend(group) == "gloo":
            # gloo stages device tensors through the host from a thread of its own; with the render
            # kernel still in flight on the stream (two processes rehearsing on one GPU) that never
            # returned.  RCCL needs no such wait: its kernel is ordered after the stream's work.
            import torch
            torch.cuda.current_stream(acc.device).synchronize()
        if dst is None:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc


def sample_split(total_frames, rank, world):
    """Frames rendered by `rank` in sample mode: (first_frame, stride, count)."""
    count = len(range(rank, total_frames, world))
    return rank, world, count
