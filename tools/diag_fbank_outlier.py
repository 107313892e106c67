"""Diagnosis of the fbank case the device fuzz r15bt failed on (7.07e-3 between the kernel and the torch-fp32 oracle on ONE value of 358 400): the same case on the SIMT
emulator (no GPU needed), kernel / fp32 oracle / fp64 arbiter side by side at the largest difference.  Result: profiles/r15bt/fbank_outlier_emulator_vs_oracle32_vs_f64.log
(the fp32 ORACLE is 6.5e-3 from the arbiter on the utterance's lowest log energy, the kernel 6.0e-4).  usage: python tools/diag_fbank_outlier.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(ROOT,'tests'),ROOT,os.path.join(ROOT,'voiceprintrecognition-pytorch_amd')]
import torch
import layer_checks as lc
from emu_lib import emu_cdll
from oracle import frontend
from mvector import _hip
kw=dict(B=8, L=112000, ragged=True, bins=80, seed=187)
extra={'frame_length': 26.0, 'snip_edges': False, 'frame_shift': 12.5, 'high_freq': 7600.0, 'use_power': False}
wav=frontend.synth_waveforms(kw['B'],kw['L'],seed=kw['seed'])
g=torch.Generator().manual_seed(kw['seed']); ratio=torch.rand(kw['B'],generator=g)*0.8+0.2; ratio[0]=1.0
args=dict(dict(sample_frequency=16000,num_mel_bins=80),**extra)
fb=_hip.Fbank(args,cdll=emu_cdll(),kernel='auto',subtract_time_mean=True)
print(fb.info())
out=fb(wav,ratio,None)
ref=frontend.audio_featurizer(wav,ratio,'Fbank',args)
ref64=frontend.audio_featurizer_fbank_f64(wav,ratio,args)
d=(out-ref).abs()
i=int(d.argmax()); b,t,m=i//(out.shape[1]*out.shape[2]), i//out.shape[2]%out.shape[1], i%out.shape[2]
print('argmax',b,t,m,'out',out[b,t,m].item(),'ref32',ref[b,t,m].item(),'ref64',ref64[b,t,m].item())
e_hip=(out.double()-ref64).abs(); e32=(ref.double()-ref64).abs()
print('HIP vs f64: max',e_hip.max().item(),'n>1e-3',int((e_hip>1e-3).sum()),' oracle32 vs f64: max',e32.max().item(),'n>1e-3',int((e32>1e-3).sum()))
print('at argmax: hip err',e_hip[b,t,m].item(),'oracle32 err',e32[b,t,m].item())
# raw (no CMN) log energy at that position: how close to the floor?
raw=frontend.kaldi_fbank_f64(wav[b:b+1],**args)
print('raw f64 value at (t,m):',raw.reshape(-1,80)[t,m].item(),' row min/median:',raw.min().item(),raw.median().item())
