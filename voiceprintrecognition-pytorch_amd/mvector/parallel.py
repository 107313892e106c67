"""Multi-GPU layout of the embedding path: one process per GPU, batch rows sharded across ranks, and ONE
exchange step -- an all-gather of the ``[B_local, D]`` embedding shards (RCCL over xGMI on MI355X, i.e. the
``nccl`` backend of torch.distributed) -- before each rank scores its rows against all rows.

The reference has no counterpart (its inference is single-process: mvector/predict.py, trainer.py:403-485); the
only collective it ever issues is DDP's gradient all-reduce in training (trainer.py:356).
"""
import contextlib

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


def sync_native_choices(model, device=None, group=None, src=0):
    """COLLECTIVE, once per job after the weights are loaded (every rank calls it): choices a native handle makes from the checkpoint are taken from
    rank ``src`` and pinned on every rank -- today the CAM++ FCM head precision (``CAMPPlus.sync_native_head``).  The forwards themselves never issue a
    collective, so ranks may run different numbers of forwards (empty shards, rank-0-only evaluation).  Returns {module name: pinned value}."""
    out = {}
    for name, m in model.named_modules():
        if hasattr(m, 'sync_native_head'):
            out[name] = m.sync_native_head(device=device, group=group, src=src)
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


_bucket_streams = {}


def _side_streams(device, n):
    """n HIP streams of ``device`` for embed_bucketed, created once per device"""
    key = (device.index if device.index is not None else torch.cuda.current_device(), n)
    if key not in _bucket_streams:
        _bucket_streams[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return _bucket_streams[key]


@torch.no_grad()
def embed_bucketed(featurizer, model, waveforms, max_buckets=8, device=None, streams=2):
    """Embeddings [N, D] (input order, on every rank) of N variable-length waveforms (1-D float tensors).

    Utterances are bucketed by length (``length_buckets``); every bucket is sharded over the ranks like a fixed-length
    batch (``shard_rows``), zero-padded to the bucket's longest utterance, featurised with the length ratios and embedded;
    the shards travel through one all-gather per bucket (padded to equal row counts, RCCL on GPUs).

    On a CUDA device the buckets are embedded on ``streams`` HIP streams in turn (round 6): a bucket is a handful of utterances, and the late
    stages of a 2-D backbone on a handful of utterances are launches of a few dozen workgroups (ERes2NetV2 54.9 M: 1 256 launches per pass, most of
    them on a fraction of the chip) -- two buckets in flight fill each other's gaps.  One native handle serves both streams (every forward brings the
    workspace of its own stream: the C ABI's contract); a row's bits do not depend on it.  Collectives are issued afterwards, in bucket order, on
    the caller's stream.  ``streams=1``: everything on the caller's stream."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    lens = [int(w.numel()) for w in waveforms]
    out = None
    buckets = length_buckets(lens, max_buckets)
    side = _side_streams(device, streams) if device.type == 'cuda' and streams > 1 and len(buckets) > 1 else None
    main = torch.cuda.current_stream(device) if side else None
    if side:
        for s in side:
            s.wait_stream(main)   # the waveforms (and the weights) were produced on the caller's stream
    local = []
    for k, idx in enumerate(buckets):
        longest = max(lens[i] for i in idx)
        lo, hi = shard_rows(len(idx), rank, world)
        mine = idx[lo:hi]
        emb_local = None
        if mine:
            with (torch.cuda.stream(side[k % len(side)]) if side else contextlib.nullcontext()):
                wav = torch.stack([F.pad(waveforms[i].to(device=device, dtype=torch.float32), (0, longest - lens[i])) for i in mine])
                ratio = torch.tensor([lens[i] / longest for i in mine], dtype=torch.float32, device=device)
                emb_local = model(featurizer(wav, ratio))
                if side:
                    emb_local.record_stream(main)   # (allocated on the side stream, consumed on the caller's)
        local.append(emb_local)
    if side:
        for s in side:
            main.wait_stream(s)
    for idx, emb_local in zip(buckets, local):
        lo, hi = shard_rows(len(idx), rank, world)
        per = -(-len(idx) // world)  # rows every rank contributes to the all-gather (padded)
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


# ---------------------------------------------------------------------------------------------------------------------
# Host <-> device pipeline for a stream of batches (pinned-memory upload of 16-bit PCM, conversion and dB
# normalisation on the device, asynchronous download of the embeddings).  One batch takes 0.5 ms to upload as int16 (24.6 MB for
# 256 x 3 s) against ~3.5 ms of compute: run back to back on one stream -- what a loop over predict_batch does -- the upload, the
# download and the host synchronisation cost 15-20 % (bench.py h2d_inclusive); on a copy stream they disappear behind the compute.

@torch.no_grad()
def embed_stream(featurizer, model, batches, device=None, target_db=None, depth=2):
    """Generator: embeddings (CPU tensors ``[B_i, D]`` in pinned memory; a result stays valid until the next-but-one result
    has been requested -- copy it to keep it) of an iterable of waveform batches, in order.

    ``batches`` yields CPU tensors ``[B_i, L_i]``: int16 PCM (converted to float32 / 32768, with the reference's dB
    normalisation when ``target_db`` is given, mvector/predict.py:185-212) or float32 waveforms, ideally in pinned memory.
    On a CUDA device the upload of batch i+1 and the download of batch i-1 run on a copy stream while batch i computes on the
    caller's stream.  Device memory is bounded: ``depth`` input slots and ``depth + 1`` pinned result slots, each a flat buffer
    that only grows to the largest batch seen (a length-sorted evaluation list with a different ``L`` per batch reuses the
    same slots; the first version kept one ring per distinct shape and grew without bound).

    Contract for the producer: a yielded host batch is read by an ASYNCHRONOUS copy -- it must stay untouched until the result
    of that batch has been returned (a producer that refills one pinned staging buffer needs ``depth + 1`` of them).
    Results are identical to calling the featurizer and the model batch by batch, on the GPU and on the CPU."""
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)

    def compute(dev_batch):
        if dev_batch.dtype == torch.int16:
            from mvector import _hip
            dev_batch, _ = _hip.wave_prepare(dev_batch, target_db=target_db)
        return model(featurizer(dev_batch))

    if device.type != 'cuda':
        for host in batches:
            if host.dtype == torch.int16:
                w = host.to(torch.float32) / 32768.0
                if target_db is not None:  # AudioSegment.normalize (predict.py:210-211): gain from the mean square over the row
                    rms_db = 10.0 * torch.log10(w.pow(2).mean(dim=1, keepdim=True).clamp_min(1e-30))
                    gain = target_db - rms_db
                    w = torch.where(gain <= 300.0, w * torch.pow(10.0, gain / 20.0), w)  # silence stays unscaled, as on the device
            else:
                w = host
            yield model(featurizer(w))
        return

    import collections
    main = torch.cuda.current_stream(device)
    copy = torch.cuda.Stream(device)
    inflight = collections.deque()   # (pinned result, download event), oldest first

    class Slots:
        """ring of flat buffers that grow on demand; ``free[k]`` = event after which slot k may be overwritten"""

        def __init__(self, n, make):
            self.n, self.make, self.bufs, self.free, self.count = n, make, [None] * n, [None] * n, 0

        def take(self, shape, dtype, wait):
            k = self.count % self.n
            self.count += 1
            if self.free[k] is not None:
                wait(self.free[k])
            numel = 1
            for d in shape:
                numel *= int(d)
            nbytes = numel * torch.empty((), dtype=dtype).element_size()
            if self.bufs[k] is None or self.bufs[k].numel() < nbytes:
                if self.free[k] is not None:
                    self.free[k].synchronize()   # the old buffer is released: its last reader must be done
                self.bufs[k] = self.make(nbytes)
            return k, self.bufs[k][:nbytes].view(dtype).view(tuple(shape))

    dev_slots = Slots(depth, lambda n: torch.empty(n, dtype=torch.uint8, device=device))
    out_slots = Slots(depth + 1, lambda n: torch.empty(n, dtype=torch.uint8).pin_memory())

    for host in batches:
        with torch.cuda.stream(copy):
            k_in, dev_in = dev_slots.take(host.shape, host.dtype, copy.wait_event)  # the batch `depth` uploads ago has been consumed
            dev_in.copy_(host, non_blocking=True)
            up = torch.cuda.Event()
            up.record(copy)
        main.wait_event(up)
        emb = compute(dev_in)
        done = torch.cuda.Event()
        done.record(main)
        dev_slots.free[k_in] = done
        k_out, host_out = out_slots.take(emb.shape, emb.dtype, lambda ev: ev.synchronize())  # (its result was handed out long ago)
        with torch.cuda.stream(copy):
            copy.wait_event(done)
            host_out.copy_(emb, non_blocking=True)
            down = torch.cuda.Event()
            down.record(copy)
        emb.record_stream(copy)              # the allocator must not hand `emb` out again before the download has read it
        out_slots.free[k_out] = down
        inflight.append((host_out, down))
        while len(inflight) >= depth:        # keep `depth - 1` batches in flight behind the one being submitted
            res, ev = inflight.popleft()
            ev.synchronize()
            yield res
    while inflight:
        res, ev = inflight.popleft()
        ev.synchronize()
        yield res
