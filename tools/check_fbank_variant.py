"""Parity + timing of a Fbank variant library (tools/fbank_waves_variant.py): python tools/check_fbank_variant.py tools/probe/libfbankw_w16.so"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
import torch
from mvector import _hip
from oracle import frontend
import layer_checks as lc
path = sys.argv[1]
cdll = _hip.bind_partial(ctypes.CDLL(path))
FB = dict(sample_frequency=16000, num_mel_bins=80)
res = {'lib': os.path.basename(path)}
wav = frontend.synth_waveforms(5, 48000, seed=3)
ratio = torch.tensor([1.0, 0.5, 0.75, 0.9, 0.31])
res['max_err_3s_masked'] = lc.fbank_case(cdll, 'cuda', wav, ratio, FB)
res['max_err_10s'] = lc.fbank_case(cdll, 'cuda', frontend.synth_waveforms(3, 160000, seed=4), None, FB)
res['max_err_short'] = lc.fbank_case(cdll, 'cuda', frontend.synth_waveforms(2, 5000, seed=5), None, FB)
fb = _hip.Fbank(FB, cdll=cdll)
big = frontend.synth_waveforms(256, 48000).cuda()
full = fb(big)
res['rows_equal_small_batch'] = bool(torch.equal(fb(big[:32]), full[:32]) and torch.equal(fb(big[:1]), full[:1]))
res['chunk_equals_single'] = bool(torch.equal(fb(big[:32]), fb(big[:32], workspace=False)))
ref64 = frontend.audio_featurizer_fbank_f64(big[:64].cpu(), None, FB)
res['max_err_vs_f64_64rows'] = (full[:64].cpu().double() - ref64).abs().max().item()
for _ in range(5):
    fb(big)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fb(big)
e1.record()
torch.cuda.synchronize()
res['us_256x3s'] = round(e0.elapsed_time(e1) / 50 * 1e3, 2)
print(json.dumps(res))
