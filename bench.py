"""Benchmark of the embedding-extraction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic waveforms already resident in HBM:
    waveforms [B, 48000] fp32 -> HIP Fbank-80 + CMN -> native backbone forward -> embeddings [B, 192]
    -> (N>1: RCCL all-gather of the embedding shards) -> HIP cosine block of this rank's rows vs all rows.
The batch dimension shards across ranks with no other exchange (weak scaling: B utterances per GPU).

Output: ONE JSON line on rank 0 (metric = utterances/sec embedded, BASELINE.json) with
  * per-stage times from events recorded on the launch stream inside the timed region,
  * ``roofline``: the dominant kernel class (the implicit-GEMM conv1d launches, ~60 % of the GPU time) against the dense
    fp16 MFMA peak -- algorithmic FLOPs 2*B*T*cin*cout*k per launch over the launch durations measured with HIP events the
    library records on the launch stream around every conv launch of every 4th timed step (mv_profile_enable);
    ``roofline_fbank``: the Fbank kernel against the HBM roofline the same way (algorithmic bytes 287 360 B/utt,
    BASELINE.md section 4); ``roofline_backbone``: the whole backbone stage (all kernels) against the MFMA peak;
    ``traffic`` = HBM bytes per launch from the committed PMC pass (profiles/pmc_traffic.json), null without one,
  * ``cpu_baseline``: the oracle (torch CPU fp32 port of the reference path) timed on a bounded sample on this
    host's cores -- N=1 only, outside the timed region,
  * ``parity``: max (1 - cos) between GPU and oracle embeddings on the first utterances of the batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (the ERes2Net family computes on fp32 operands, csrc/conv2d.hip)
SAMPLES = 48000              # 3 s @ 16 kHz

MODELS = {
    # name: (class, kwargs, feature method, method args, GFLOP/utt, BASELINE config label)
    'ecapa1024': ('EcapaTdnn', dict(channels=[1024, 1024, 1024, 1024, 3072]), 'Fbank',
                  dict(sample_frequency=16000, num_mel_bins=80), 11.175,
                  'EcapaTdnn (c=1024) + Fbank-80, bs=256, 3 s@16 kHz synthetic'),
    'ecapa512': ('EcapaTdnn', dict(), 'Fbank', dict(sample_frequency=16000, num_mel_bins=80), 3.090,
                 'EcapaTdnn (c=512) + Fbank-80, bs=256, 3 s@16 kHz synthetic'),
    'ecapa512_mel': ('EcapaTdnn', dict(), 'MelSpectrogram', dict(), 2.559,
                     'EcapaTdnn (c=512) + MelSpectrogram-128, bs=256, 3 s@16 kHz synthetic (BASELINE config 4, per-GPU share)'),
    'eres2net': ('ERes2Net', dict(m_channels=32), 'Fbank', dict(sample_frequency=16000, num_mel_bins=80), 10.083,
                 'ERes2Net (m_channels=32, configs/eres2net.yml) + Fbank-80, 3 s@16 kHz synthetic'),
    'eres2netv2': ('ERes2NetV2', dict(m_channels=32), 'Fbank', dict(sample_frequency=16000, num_mel_bins=80), 6.242,
                   'ERes2NetV2 (m_channels=32) + Fbank-80, 3 s@16 kHz synthetic'),
    'campp': ('CAMPPlus', dict(embd_dim=192), 'Fbank', dict(sample_frequency=16000, num_mel_bins=80), 3.355,
              'CAM++ + Fbank-80, bs=256, 3 s@16 kHz synthetic'),
}


def randomise_bn(model, seed=1):
    """randomised BatchNorm statistics/affines (default BN is the identity)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            n = m.num_features
            m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            if m.affine:
                m.weight.data.copy_(torch.rand(n, generator=g) * 0.4 + 0.8)
                m.bias.data.copy_(torch.randn(n, generator=g) * 0.1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='utterances per GPU')
    ap.add_argument('--model', default='ecapa1024', choices=sorted(MODELS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=256, help='utterances timed on the CPU oracle (~10-20 s of CPU work)')
    ap.add_argument('--cpu-threads', type=int, default=32, help='torch CPU threads for the oracle baseline')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    from mvector import _hip
    from mvector.data_utils.featurizer import AudioFeaturizer
    import mvector.models as models
    _hip.lib()  # fail loudly if the HIP library is missing

    cls, kwargs, method, margs, gflop_per_utt, label = MODELS[args.model]
    torch.manual_seed(0)
    featurizer = AudioFeaturizer(method, method_args=margs)
    model = getattr(models, cls)(input_size=featurizer.feature_dim, **kwargs)
    randomise_bn(model)
    model.eval()
    state_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev)

    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn([B, SAMPLES], generator=g)).clamp(-1, 1).to(dev)
    gathered = torch.empty((world * B, model.embd_dim), dtype=torch.float32, device=dev)

    def step(events=None):
        if events is not None:
            events[0].record()
        feats = featurizer(wav)
        if events is not None:
            events[1].record()
        emb = model(feats)
        if events is not None:
            events[2].record()
        if world > 1:
            dist.all_gather_into_tensor(gathered, emb)
            allemb = gathered
        else:
            allemb = emb
        if events is not None:
            events[3].record()
        scores = _hip.cosine(emb, allemb)
        if events is not None:
            events[4].record()
        return emb, scores

    cdll = _hip.lib()
    import ctypes

    def prof_read(cls):
        n, ms, work = ctypes.c_int32(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        _hip.check(cdll.mv_profile_read(cls, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work), 0), cdll)
        return n.value, ms.value, work.value

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # HIP events around every conv1d / fbank launch of every 4th step of the timed region (two event records per
            # launch are not free on the host: CAM++ issues 108 conv launches per step)
            cdll.mv_profile_enable(1 if i % 4 == 0 else 0)
            emb, scores = step(evs[i])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0

    cdll.mv_profile_enable(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    stage_ms = [sum(evs[i][j].elapsed_time(evs[i][j + 1]) for i in range(args.steps)) / args.steps for j in range(4)]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        fb_ms, bb_ms = stage_ms[0], stage_ms[1]
        bb_tflops = B * gflop_per_utt / (bb_ms * 1e-3) / 1e3
        # per-launch legs: algorithmic work / launch durations from the library's HIP events (launch stream, timed region)
        traffic = {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
                traffic = json.load(fh).get('kernels', {})
        except OSError:
            pass

        def pmc_bytes(prefix):
            # the PMC pass was taken on the headline workload: no number for the other models
            hits = [v for k, v in traffic.items() if k.startswith(prefix)] if args.model == 'ecapa1024' else []
            return max(hits, key=lambda v: v['launches_sampled'])['hbm_bytes_per_launch'] if hits else None

        f32_family = cls.startswith('ERes2Net')
        n_conv, ms_conv, flop_conv = prof_read(2 if f32_family else 0)
        n_fb, ms_fb, byte_fb = prof_read(1)
        conv_tflops = flop_conv / (ms_conv * 1e-3) / 1e12 if ms_conv > 0 else 0.0
        fb_gbs = byte_fb / (ms_fb * 1e-3) / 1e9 if ms_fb > 0 else 0.0
        mfma_peak = MFMA_F32_PEAK_TFLOPS if f32_family else MFMA_F16_PEAK_TFLOPS
        roof_conv = {'kernel': 'conv2d_kernel (fp32 implicit GEMM on v_mfma_f32_16x16x4_f32), all launches, padded channel counts'
                     if f32_family else
                     'conv1d (implicit GEMM on MFMA: conv1d_glds_persistent_kernel + conv1d_glds_kernel), all launches',
                     'bound': 'mfma', 'achieved': round(conv_tflops, 1), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                     'frac': round(conv_tflops / mfma_peak, 4), 'traffic': pmc_bytes('mv::conv1d_glds_persistent_kernel'),
                     'launches': n_conv, 'avg_launch_us': round(ms_conv / max(n_conv, 1) * 1e3, 2),
                     'algorithmic_gflop_per_launch': round(flop_conv / max(n_conv, 1) / 1e9, 3),
                     'share_of_step': round(ms_conv / len(range(0, args.steps, 4)) / ms_per_step, 3)}
        roof_fbank = {'kernel': 'fbank_kernel', 'bound': 'hbm', 'achieved': round(fb_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': round(fb_gbs / HBM_PEAK_GBS, 4), 'traffic': pmc_bytes('mv::fbank_kernel'), 'launches': n_fb,
                      'avg_launch_us': round(ms_fb / max(n_fb, 1) * 1e3, 2),
                      'algorithmic_bytes_per_launch': int(byte_fb / max(n_fb, 1))}
        out = {
            'metric': f'utterances/sec embedded (3 s@16 kHz, Fbank-80, {cls}, bs={B})' if f32_family else
            'utterances/sec embedded (3 s@16 kHz, Fbank-80, EcapaTdnn, bs=256)',
            'value': round(value, 1), 'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if f32_family else 'f16', 'data': 'synthetic',
            'config': {'workload': label, 'batch_per_gpu': B, 'global_batch': world * B, 'samples_per_utt': SAMPLES,
                       'frames': 298, 'parallelism': f'batch-sharded x{world}' + (' + RCCL all-gather of embeddings' if world > 1 else '')},
            'stage_ms': {'fbank_cmn': round(stage_ms[0], 4), 'backbone': round(stage_ms[1], 4),
                         'all_gather': round(stage_ms[2], 4), 'cosine': round(stage_ms[3], 4)},
            'roofline': roof_conv,
            'roofline_fbank': roof_fbank,
            'roofline_backbone': {'kernel': 'whole backbone stage (conv1d + res2 chain + SE + ASP + fc)', 'bound': 'mfma',
                                  'achieved': round(bb_tflops, 1), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                                  'frac': round(bb_tflops / mfma_peak, 4),
                                  'algorithmic_gflop_per_utt': gflop_per_utt},
        }
        if world == 1:
            # ---- parity gate + CPU baseline (oracle = test infrastructure; outside the timed region) ----
            from oracle import frontend as ofe, models as om
            n_par = min(4, B)
            wav_cpu = wav[:n_par].cpu()
            with torch.no_grad():
                ref = om.FORWARDS[cls](state_cpu, ofe.audio_featurizer(wav_cpu, None, method, margs))
            cosd = (1 - torch.nn.functional.cosine_similarity(emb[:n_par].cpu().double(), ref.double(), dim=1)).max().item()
            out['parity'] = {'max_one_minus_cos': cosd, 'utterances': n_par, 'tolerance': 1e-4}
            if not args.no_cpu_baseline:
                # all host cores oversubscribe badly on the 256-thread GPU host (0.3 utt/s): cap the pool, report the count
                torch.set_num_threads(min(os.cpu_count(), args.cpu_threads))
                n_cpu = max(1, min(args.cpu_sample, B))
                sample = wav[:n_cpu].cpu()
                with torch.no_grad():
                    om.FORWARDS[cls](state_cpu, ofe.audio_featurizer(sample[:2], None, method, margs))  # warm-up
                    tc = time.perf_counter()
                    for i in range(0, n_cpu, 32):  # chunks of 32 as mvector/predict.py:261
                        om.FORWARDS[cls](state_cpu, ofe.audio_featurizer(sample[i:i + 32], None, method, margs))
                    cpu_s = time.perf_counter() - tc
                out['cpu_baseline'] = {'value': round(n_cpu / cpu_s, 2), 'unit': 'utterances/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': f'{n_cpu} of the {B} synthetic utterances (same waveforms, same '
                                       f'weights), oracle Fbank loop + oracle {cls} forward, torch CPU fp32, {cpu_s:.1f} s'}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
