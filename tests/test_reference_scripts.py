"""The drop-in claim, tested: the reference's own entry scripts (infer_contrast.py:19-23, eval.py:18-22,
infer_recognition.py:20-47) are copied BYTE FOR BYTE into a scratch directory and run with PYTHONPATH on this package.

Only what the scripts read from disk is prepared (a YAML config, a model.pth in the reference's checkpoint layout, wav
files, the reference's audio_db/ enrolment folders).  infer_recognition.py loops on input() for ever: its answers are
scripted and its microphone is replaced through RecordAudio.backend by the launcher below -- the script file itself is
untouched.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT, PKG
from helpers import GOLDEN, load_case

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout is not on this machine')

LAUNCHER = r'''
"""runs a reference script as __main__ with scripted input() answers and a file-backed microphone"""
import builtins, runpy, sys, numpy as np
import scipy.io.wavfile as wavfile
from mvector.utils.record import RecordAudio
script, answers, takes = sys.argv[1], sys.argv[2].split('|'), sys.argv[3].split('|')
state = dict(a=0, t=0)
def fake_input(prompt=''):
    if state['a'] >= len(answers):
        raise SystemExit(0)          # scripted session over: leave the script's endless loop
    v = answers[state['a']]; state['a'] += 1
    print(prompt + v)
    return v
def fake_mic(sample_rate, num_frames, channels):
    sr, pcm = wavfile.read(takes[state['t'] % len(takes)]); state['t'] += 1
    assert sr == sample_rate
    x = pcm.astype(np.float32) / 32768.0
    return x[:num_frames].reshape(-1, 1)
builtins.input = fake_input
RecordAudio.backend = fake_mic
sys.argv = [script] + sys.argv[4:]
runpy.run_path(script, run_name='__main__')
'''


def _workdir(tmp_path):
    import scipy.io.wavfile as wavfile
    man, sd, _, _, _ = load_case('tdnn')
    model_dir = tmp_path / 'models' / 'TDNN_Fbank' / 'best_model'
    model_dir.mkdir(parents=True)
    torch.save({'0.' + k: v for k, v in sd.items()}, str(model_dir / 'model.pth'))
    data = tmp_path / 'dataset'
    data.mkdir()
    z = np.load(os.path.join(GOLDEN, 'real_audio.npz'))
    names = ['a_1', 'a_2', 'b_1', 'b_2']
    for n, pcm in zip(names, z['pcm16']):
        wavfile.write(str(data / f'{n}.wav'), 16000, pcm)
    with open(str(data / 'enroll_list.txt'), 'w') as f:
        f.write(f'dataset/a_1.wav\t0\ndataset/b_1.wav\t1\n')
    with open(str(data / 'trials_list.txt'), 'w') as f:
        f.write(f'dataset/a_2.wav\t0\ndataset/b_2.wav\t1\n')
    cfg = dict(dataset_conf=dict(dataset=dict(min_duration=0.3, max_duration=3, sample_rate=16000, use_dB_normalization=True,
                                              target_dB=-20),
                                 eval_conf=dict(batch_size=2, max_duration=20), dataLoader=dict(num_workers=0),
                                 enroll_list='dataset/enroll_list.txt', trials_list='dataset/trials_list.txt'),
               preprocess_conf=dict(feature_method='Fbank', method_args=dict(sample_frequency=16000, num_mel_bins=80)),
               model_conf=dict(model='TDNN', model_args=dict(embd_dim=192)))
    (tmp_path / 'configs').mkdir()
    with open(str(tmp_path / 'configs' / 'tdnn.yml'), 'w') as f:
        yaml.safe_dump(cfg, f)
    for s in ('infer_contrast.py', 'eval.py', 'infer_recognition.py'):
        shutil.copyfile(os.path.join(REF, s), str(tmp_path / s))
        assert open(os.path.join(REF, s), 'rb').read() == open(str(tmp_path / s), 'rb').read()
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''), PYTHONIOENCODING='utf-8')
    common = ['--configs=configs/tdnn.yml', '--use_gpu=False']
    return env, common


def _run(cmd, env, cwd, stdin=None):
    p = subprocess.run(cmd, env=env, cwd=cwd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode('utf-8', 'replace')
    assert p.returncode == 0, out
    return out


def test_reference_infer_contrast_runs_unchanged(tmp_path):
    env, common = _workdir(tmp_path)
    out = _run([sys.executable, 'infer_contrast.py', *common, '--model_path=models/TDNN_Fbank/best_model/',
                '--audio_path1=dataset/a_1.wav', '--audio_path2=dataset/b_2.wav'], env, str(tmp_path))
    assert '相似度为' in out


def test_reference_eval_runs_unchanged(tmp_path):
    env, common = _workdir(tmp_path)
    out = _run([sys.executable, 'eval.py', *common, '--resume_model=models/TDNN_Fbank/best_model/',
                f'--save_image_path={tmp_path}/output/images/'], env, str(tmp_path))
    assert 'EER' in out and 'MinDCF' in out


def test_reference_infer_recognition_runs_unchanged(tmp_path):
    """register two users from the 'microphone', identify one, delete one, identify again; also loads the reference's shipped
    audio_db/ folders (44.1 kHz stereo) at start-up"""
    env, common = _workdir(tmp_path)
    shutil.copytree(os.path.join(REF, 'audio_db'), str(tmp_path / 'audio_db'))
    launcher = tmp_path / 'launch.py'
    launcher.write_text(LAUNCHER)
    answers = ['0', '', 'alice', '0', '', 'bob', '1', '', '2', 'bob', '1', '', '7']
    takes = [str(tmp_path / 'dataset' / n) for n in ('a_1.wav', 'b_1.wav', 'a_2.wav', 'b_2.wav')]
    out = _run([sys.executable, str(launcher), 'infer_recognition.py', '|'.join(answers), '|'.join(takes), *common,
                '--model_path=models/TDNN_Fbank/best_model/', '--audio_db_path=audio_db/', '--threshold=0.0'],
               env, str(tmp_path))
    assert out.count('开始录音') == 4 and '请正确选择功能' in out
    assert '识别说话的为' in out
    assert os.path.exists(str(tmp_path / 'audio_db' / 'audio_indexes.bin'))
    assert os.path.isdir(str(tmp_path / 'audio_db' / 'alice')) and not os.path.isdir(str(tmp_path / 'audio_db' / 'bob'))


def _public_api(path):
    """{name: argument names} of the top-level functions, classes and their public methods of a source file (no import: AST)"""
    import ast
    out = {}
    for n in ast.parse(open(path).read()).body:
        if isinstance(n, ast.ClassDef):
            out[n.name] = None
            for m in n.body:
                if isinstance(m, ast.FunctionDef) and (not m.name.startswith('_') or m.name == '__init__'):
                    out[f'{n.name}.{m.name}'] = [a.arg for a in m.args.args + m.args.kwonlyargs]
        elif isinstance(n, ast.FunctionDef) and not n.name.startswith('_'):
            out[n.name] = [a.arg for a in n.args.args]
    return out


def test_public_names_and_arguments_of_the_path_exist_here():
    """every top-level name, public method and argument name the reference defines in the files of the embedding path
    exists in this package's file of the same name (extra arguments with defaults are allowed; training-only names are listed apart)"""
    files = ['mvector/predict.py', 'mvector/data_utils/featurizer.py', 'mvector/metric/metrics.py', 'mvector/models/ecapa_tdnn.py', 'mvector/models/tdnn.py',
             'mvector/models/pooling.py', 'mvector/models/utils.py', 'mvector/utils/utils.py', 'mvector/utils/record.py', 'mvector/data_utils/collate_fn.py']
    missing = []
    for rel in files:
        ref, ours = _public_api(os.path.join(REF, rel)), _public_api(os.path.join(PKG, rel))
        for name, args in ref.items():
            if name not in ours:
                if name.endswith('.__init__') and name.rsplit('.', 1)[0] in ours:
                    continue   # (a constructor without arguments that the class here inherits)
                missing.append(f'{rel}: {name}')
            elif args and ours[name] is not None:
                lost = [a for a in args if a not in ours[name]]
                if lost:
                    missing.append(f'{rel}: {name} lacks {lost}')
    assert not missing, missing
    # the model files define the classes users construct (the reference's internal building blocks of CAM++ / ERes2Net are laid out differently here;
    # the state_dict contract is what test_state_dict_layout_equals_reference_manifest pins)
    for rel, names in (('mvector/models/campplus.py', ['CAMPPlus']), ('mvector/models/eres2net.py', ['ERes2Net', 'ERes2NetV2']),
                       ('mvector/trainer.py', ['MVectorTrainer', 'MVectorTrainer.evaluate', 'MVectorTrainer.train', 'MVectorTrainer.export', 'MVectorTrainer.extract_features'])):
        ref, ours = _public_api(os.path.join(REF, rel)), _public_api(os.path.join(PKG, rel))
        for name in names:
            assert name in ref and name in ours, (rel, name)
        for cls in [n for n in names if '.' not in n]:
            lost = [a for a in ref[f'{cls}.__init__'] if a not in ours[f'{cls}.__init__']]
            assert not lost, (rel, cls, lost)
