"""Multi-GPU layout of the embedding path: one process per GPU, batch rows sharded across ranks, and ONE
exchange step -- an all-gather of the ``[B_local, D]`` embedding shards (RCCL over xGMI on MI355X, i.e. the
``nccl`` backend of torch.distributed) -- before each rank scores its rows against all rows.

The reference has no counterpart (its inference is single-process: mvector/predict.py, trainer.py:403-485); the
only collective it ever issues is DDP's gradient all-reduce in training (trainer.py:356).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F


def shard_rows(n_rows, rank=None, world=None):
    """Contiguous row range [lo, hi) of this rank (first ``n_rows % world`` ranks get one extra row)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_embeddings(emb_local, out=None, always=False):
    """[B_local, D] per rank (equal B_local) -> [world * B_local, D] on every rank, rank-major order.
    ``always``: issue the collective even in a one-rank group (a 1-GPU box can then exercise the RCCL call itself)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always):
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


# ---------------------------------------------------------------------------------------------------------------------
# Variable-length batches (BASELINE config 4: 1-10 s utterances "with bucketing").  The reference pads a batch to its longest
# utterance and its CMN then runs over the padded frames (predict.py:244-255, featurizer.py:79: quirk Q2), so an
# utterance's embedding depends on what it is batched with; bucketing by length bounds both the wasted frames and that
# dependence.  Inside a bucket the semantics are exactly predict_batch's.

def length_buckets(num_samples, max_buckets=8):
    """Group utterance indices into <= max_buckets buckets of similar length: equal-width length ranges between the shortest
    and the longest utterance (padding waste <= 1/max_buckets of that range), empty ranges dropped, input order kept inside
    a bucket.  Returns a list of index lists, shortest bucket first."""
    n = [int(v) for v in num_samples]
    if not n:
        return []
    lo, hi = min(n), max(n)
    k = max(1, int(max_buckets))
    width = (hi - lo) // k + 1
    buckets = [[] for _ in range(k)]
    for i, v in enumerate(n):
        buckets[(v - lo) // width].append(i)
    return [b for b in buckets if b]


@torch.no_grad()
def embed_bucketed(featurizer, model, waveforms, max_buckets=8, device=None):
    """Embeddings [N, D] (input order, on every rank) of N variable-length waveforms (1-D float tensors).

    Utterances are bucketed by length (``length_buckets``); every bucket is sharded over the ranks like a fixed-length
    batch (``shard_rows``), zero-padded to the bucket's longest utterance, featurised with the length ratios and embedded;
    the shards travel through one all-gather per bucket (padded to equal row counts, RCCL on GPUs)."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    if device is None:
        device = next(model.parameters()).device
    lens = [int(w.numel()) for w in waveforms]
    out = None
    for idx in length_buckets(lens, max_buckets):
        longest = max(lens[i] for i in idx)
        lo, hi = shard_rows(len(idx), rank, world)
        mine = idx[lo:hi]
        per = -(-len(idx) // world)  # rows every rank contributes to the all-gather (padded)
        emb_local = None
        if mine:
            wav = torch.stack([F.pad(waveforms[i].to(device=device, dtype=torch.float32), (0, longest - lens[i])) for i in mine])
            ratio = torch.tensor([lens[i] / longest for i in mine], dtype=torch.float32, device=device)
            emb_local = model(featurizer(wav, ratio))
        if out is None:
            dim = emb_local.shape[1] if emb_local is not None else model.embd_dim
            out = torch.zeros((len(waveforms), dim), dtype=torch.float32, device=device)
        if not distributed:
            out[torch.tensor(idx, device=device)] = emb_local
            continue
        block = torch.zeros((per, out.shape[1]), dtype=torch.float32, device=device)
        if emb_local is not None:
            block[:emb_local.shape[0]] = emb_local
        gathered = all_gather_embeddings(block)
        for r in range(world):
            rlo, rhi = shard_rows(len(idx), r, world)
            if rhi > rlo:
                out[torch.tensor(idx[rlo:rhi], device=device)] = gathered[r * per:r * per + (rhi - rlo)]
    return out
