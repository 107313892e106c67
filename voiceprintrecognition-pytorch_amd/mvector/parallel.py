"""Multi-GPU layout of the embedding path: one process per GPU, batch rows sharded across ranks, and ONE
exchange step -- an all-gather of the ``[B_local, D]`` embedding shards (RCCL over xGMI on MI355X, i.e. the
``nccl`` backend of torch.distributed) -- before each rank scores its rows against all rows.

The reference has no counterpart (its inference is single-process: mvector/predict.py, trainer.py:403-485); the
only collective it ever issues is DDP's gradient all-reduce in training (trainer.py:356).
"""
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank=None, world=None):
    """Contiguous row range [lo, hi) of this rank (first ``n_rows % world`` ranks get one extra row)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_embeddings(emb_local, out=None):
    """[B_local, D] per rank (equal B_local) -> [world * B_local, D] on every rank, rank-major order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return emb_local
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * emb_local.shape[0], emb_local.shape[1]), dtype=emb_local.dtype, device=emb_local.device)
    dist.all_gather_into_tensor(out, emb_local.contiguous())
    return out


def cosine_block(emb_local, emb_all):
    """Rows of this rank against all rows: HIP kernel on CUDA tensors, torch on CPU tensors (gloo tests)."""
    if emb_local.is_cuda:
        from mvector import _hip
        return _hip.cosine(emb_local, emb_all)
    a = emb_local / emb_local.norm(dim=1, keepdim=True)
    b = emb_all / emb_all.norm(dim=1, keepdim=True)
    return a @ b.t()


@torch.no_grad()
def embed_and_score(featurizer, model, wav_local, lens_ratio=None):
    """One step of the sharded path: featurise + embed the local rows, all-gather, score local rows vs all."""
    emb = model(featurizer(wav_local, lens_ratio))
    emb_all = all_gather_embeddings(emb)
    return emb, emb_all, cosine_block(emb, emb_all)
