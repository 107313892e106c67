/* libmvector_hip.so -- C ABI of the MI355X-native (gfx950) embedding-extraction hot path.
 *
 * The reference (yeyupiaoling/VoiceprintRecognition-Pytorch, mvector 1.1.1) is pure Python and has
 * NO native / FFI / operator boundary for this path (setup.py:56 ``ext_modules=[]``), so there is no
 * reference FFI to mirror symbol-for-symbol.  The drop-in boundary is the Python surface
 * (AudioFeaturizer / build_model / model classes / MVectorPredictor); this header is the C ABI that a
 * maintainer binds UNDER that surface (ctypes stub shown in INTEGRATION.md).  Each entry point cites
 * the reference code whose device work it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer on the current HIP
 *     device unless its name ends in ``_host``;
 *   - the caller (PyTorch) owns inputs, outputs and workspaces; handles own only derived constants
 *     (window / filterbank tables, pre-packed fp16 weights) allocated at create time;
 *   - stream-ordered: work is enqueued on ``stream`` (a hipStream_t passed as void*), nothing in a
 *     ``*_forward`` call synchronises the device; ``*_create`` / ``*_destroy`` may synchronise;
 *   - every function returns MV_OK (0) or a negative MV_ERR_* code and records a message readable
 *     through mv_last_error() (thread-local).  Handles are immutable after create, so forwards on
 *     different streams with different workspaces may run concurrently.
 */
#ifndef MVECTOR_HIP_H
#define MVECTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared in this header are its ONLY dynamic exports besides the HIP
 * runtime's kernel handles (tests/test_native_library.py checks `nm -D` against this list). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define MV_OK 0
#define MV_ERR_INVALID_ARGUMENT (-1)
#define MV_ERR_HIP (-2)
#define MV_ERR_UNSUPPORTED (-3)
#define MV_ERR_WORKSPACE (-4)
#define MV_ERR_MISSING_TENSOR (-5)

#define MV_ABI_VERSION 5

typedef void* mv_stream_t; /* hipStream_t */

const char* mv_last_error(void);
int mv_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Front-end 1: Kaldi log-mel filterbank + per-utterance time-mean subtraction + length mask.
 * Replaces KaldiFbank.forward (mvector/data_utils/featurizer.py:119-132 ->
 * torchaudio.compliance.kaldi.fbank) fused with AudioFeaturizer.forward's normalisation and mask
 * (featurizer.py:77-90).  Arithmetic as restated in oracle/frontend.py::kaldi_fbank.
 * ------------------------------------------------------------------------------------------------ */
typedef struct MvFbankCfg {
    float sample_frequency;       /* 16000 */
    float frame_length_ms;        /* 25 */
    float frame_shift_ms;         /* 10 */
    int32_t num_mel_bins;         /* 80 (4 .. 128: torchaudio's get_mel_banks asserts num_bins > 3) */
    float low_freq;               /* 20 (0 <= low_freq < Nyquist, low_freq < high_freq: refused otherwise, as get_mel_banks asserts) */
    float high_freq;              /* 0 => Nyquist (+ high_freq if <= 0, as Kaldi); beyond Nyquist: refused (get_mel_banks asserts) */
    float preemphasis_coefficient;/* 0.97 */
    int32_t remove_dc_offset;     /* 1 */
    int32_t use_power;            /* 1 */
    int32_t use_log_fbank;        /* 1 */
    int32_t subtract_time_mean;   /* 1: featurizer.py:79 (mean over ALL frames, padded ones included) */
    /* ---- since ABI 4: the further kaldi.fbank keyword arguments featurizer.py:128 forwards (**kwargs) ---- */
    int32_t window_type;          /* MV_WINDOW_POVEY (0) | HAMMING | HANNING | RECTANGULAR | BLACKMAN */
    float blackman_coeff;         /* 0.42 (window_type BLACKMAN only) */
    int32_t snip_edges;           /* 1: frames lie inside the signal.  0: T = (L + shift / 2) / shift frames, the signal mirrored at both ends
                                   * (needs the caller workspace of mv_fbank_forward_ws: the mirrored rows are written there first) */
    int32_t subtract_mean;        /* 0.  kaldi.fbank's own subtract_mean: column means over the utterance's frames, BEFORE the wrapper's mean */
    float min_duration;           /* 0 s: shorter signals give no frames */
    float vtln_warp;              /* 1.0 = no vocal-tract-length warp of the filter edges; otherwise kaldi's 3-piece linear warp between */
    float vtln_low, vtln_high;    /* 100, -500 (negative: + Nyquist) */
    int32_t kernel;               /* MV_FBANK_KERNEL_AUTO (0) | MV_FBANK_KERNEL_GENERIC (fbank_kernel) | MV_FBANK_KERNEL_TILE (fbank_tile_kernel;
                                   * create fails when the mel geometry has no instantiation).  Both kernels implement the same contract; the
                                   * field exists so that tests and tools/bench_fbank.py can run either on any geometry. */
    /* ---- since ABI 5 ---- */
    int64_t min_samples;          /* > 0: the threshold of min_duration in SAMPLES, ceil(min_duration * sample_frequency) evaluated by the caller in
                                   * double as torchaudio compares (`len(waveform) < min_duration * sample_frequency`): the float32 field above
                                   * rounds 0.1 / 0.2 / 0.3 s upwards and a clip of exactly min_duration lost its frames.  0: derived
                                   * from min_duration. */
    int32_t use_energy;           /* 0.  1: one more column, the log energy of every frame -- out is [B, T, num_mel_bins + 1] (kaldi.fbank's use_energy;
                                   * AudioFeaturizer.feature_dim, featurizer.py:110-111, keeps reporting num_mel_bins, as the reference's does).  The
                                   * mel columns come from the same kernels through the caller workspace (mv_fbank_workspace_bytes; the forwards
                                   * without a workspace refuse), the energy column from fbank_energy_kernel; time mean and mask cover it too. */
    int32_t raw_energy;           /* 1: energy of the frame after the DC removal, before pre-emphasis and window; 0: of the windowed frame */
    float energy_floor;           /* 1.0: log energies below log(energy_floor) are raised to it; 0 = no floor (kaldi.fbank's rule) */
    int32_t htk_compat;           /* 0: the energy is column 0; 1: the last column */
} MvFbankCfg;

enum { MV_WINDOW_POVEY = 0, MV_WINDOW_HAMMING = 1, MV_WINDOW_HANNING = 2, MV_WINDOW_RECTANGULAR = 3, MV_WINDOW_BLACKMAN = 4 };
enum { MV_FBANK_KERNEL_AUTO = 0, MV_FBANK_KERNEL_GENERIC = 1, MV_FBANK_KERNEL_TILE = 2 };

typedef struct MvFbank MvFbank;

void mv_fbank_default_cfg(MvFbankCfg* cfg);
int mv_fbank_create(const MvFbankCfg* cfg, MvFbank** out);
int mv_fbank_destroy(MvFbank* h);
/* T = 1 + (L - window) / shift, or 0 when L < window (snip_edges = 1); (L + shift / 2) / shift (snip_edges = 0); 0 below min_duration */
int mv_fbank_num_frames(const MvFbank* h, int64_t num_samples, int64_t* num_frames);
/* which kernel this handle launches: *tile_kernel = 1 for fbank_tile_kernel (mel geometry of the reference configurations:
 * 80 bins / 16 kHz / 512-point FFT), 0 for the generic fbank_kernel; pass_steps[2] = FFT bins each mel MFMA pass walks */
int mv_fbank_info(const MvFbank* h, int32_t* tile_kernel, int32_t* pass_steps);
/* wav: [B, L] fp32 rows ``wav_stride`` elements apart.  lens_ratio: [B] fp32 or NULL
 * (featurizer.py:80-90: frames t >= round_half_even(ratio * T) are zeroed).  out: [B, T, num_mel_bins]
 * fp32, contiguous ([B, T, num_mel_bins + 1] with MvFbankCfg.use_energy, which needs mv_fbank_forward_ws). */
int mv_fbank_forward(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                     const float* lens_ratio, float* out, mv_stream_t stream);
/* The same forward with a caller workspace.  A batch of fewer utterances than the chip has CUs (predict()'s single utterance,
 * predict_batch()'s 32) is faster as several workgroups per utterance + a finish pass; that form needs per-CALL scratch for the
 * column sums of every 4-frame group, which the handle must not own (two forwards on two streams share the handle).
 * mv_fbank_workspace_bytes reports what this (B, L) needs on the current device (0 = none); `workspace` must be 16-byte aligned
 * and stay untouched until the forward has run on `stream`.  With a NULL or short workspace -- and in mv_fbank_forward /
 * mv_fbank_forward_varlen -- every utterance runs on one workgroup.  The two forms produce IDENTICAL bits: the time mean of an
 * utterance is summed in one fixed order (csrc/fbank.hip, FbankArgs) that depends on its own length only, never on the batch
 * size, the form or the stream. */
int mv_fbank_workspace_bytes(const MvFbank* h, int32_t B, int64_t L, size_t* bytes);
int mv_fbank_forward_ws(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const float* lens_ratio,
                        float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream);

/* Variable-length batch: row b holds num_samples[b] <= L valid samples (device array, int64).  Every utterance is
 * featurised on its own -- frame count, time mean -- and rows beyond its frame count are zero in the [B, T(L), F] output:
 * the result of the reference's evaluation path, which featurises per utterance (mvector/data_utils/reader.py:100-106)
 * and zero-pads the features (collate_fn.py:11-19). */
int mv_fbank_forward_varlen(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                            const int64_t* num_samples, float* out, mv_stream_t stream);
/* the same with the caller workspace of mv_fbank_workspace_bytes(h, B, L): what snip_edges = 0 handles need (the mirrored rows) */
int mv_fbank_forward_varlen_ws(const MvFbank* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride, const int64_t* num_samples,
                               float* out, void* workspace, size_t workspace_bytes, mv_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Front-end 2: MelSpectrogram (power STFT, centre/reflect padding, HTK mel filterbank, NO log) +
 * time-mean subtraction + mask.  Replaces torchaudio.transforms.MelSpectrogram(**method_args)
 * (featurizer.py:41-42) and featurizer.py:77-90.  Oracle: oracle/frontend.py::mel_spectrogram.
 * ------------------------------------------------------------------------------------------------ */
typedef struct MvMelSpecCfg {
    int32_t sample_rate; /* 16000 */
    int32_t n_fft;       /* 400 */
    int32_t win_length;  /* 400 (<= n_fft) */
    int32_t hop_length;  /* 200 */
    float f_min;         /* 0 */
    float f_max;         /* sample_rate / 2 */
    int32_t n_mels;      /* 128 */
    float power;         /* 2.0; any positive exponent of |X| (the FFT kernels take 2, the dense-DFT kernel every other one) */
    int32_t center;      /* 1 */
    int32_t subtract_time_mean;
    /* ---- since ABI 4: further MelSpectrogram(**method_args) keyword arguments (featurizer.py:41-42), all of them tables ---- */
    int32_t mel_scale;   /* MV_MEL_HTK (0) | MV_MEL_SLANEY */
    int32_t norm;        /* 0 = None | MV_MEL_NORM_SLANEY (1): filter j times 2 / (f[j + 2] - f[j]) */
    int32_t normalized;  /* 0 = False | MV_STFT_NORM_WINDOW (1; True): spectrum / sqrt(sum window^2) | MV_STFT_NORM_FRAME_LENGTH (2): / sqrt(n_fft) */
    const float* window; /* NULL = periodic Hann | HOST array [win_length]: what window_fn(win_length, **wkwargs) returned (copied at create) */
    /* ---- since ABI 5 ---- */
    int32_t pad;         /* 0.  Zeros in front of and behind the signal before the transform (torchaudio.functional.spectrogram's `pad`) */
    int32_t pad_mode;    /* MV_STFT_PAD_REFLECT (0) | CONSTANT | REPLICATE | CIRCULAR: how torch.stft(center=True) extends the signal by n_fft / 2.
                          * pad > 0 or a mode other than reflect writes the extended signal to the caller workspace first (one more pass over the
                          * waveform for a non-default argument); the transform kernels then run on it unchanged */
} MvMelSpecCfg;

enum { MV_STFT_PAD_REFLECT = 0, MV_STFT_PAD_CONSTANT = 1, MV_STFT_PAD_REPLICATE = 2, MV_STFT_PAD_CIRCULAR = 3 };

enum { MV_MEL_HTK = 0, MV_MEL_SLANEY = 1 };
enum { MV_MEL_NORM_NONE = 0, MV_MEL_NORM_SLANEY = 1 };
enum { MV_STFT_NORM_NONE = 0, MV_STFT_NORM_WINDOW = 1, MV_STFT_NORM_FRAME_LENGTH = 2 };

typedef struct MvMelSpec MvMelSpec;

void mv_melspec_default_cfg(MvMelSpecCfg* cfg);
int mv_melspec_create(const MvMelSpecCfg* cfg, MvMelSpec** out);
/* *tile_kernel = 1 when the handle launches melspec_tile_kernel (n_fft = 400 as a real FFT fused with mel + CMN + mask in one
 * launch), 0 for the dense-DFT kernels (any other geometry) */
int mv_melspec_info(const MvMelSpec* h, int32_t* tile_kernel);
int mv_melspec_destroy(MvMelSpec* h);
int mv_melspec_num_frames(const MvMelSpec* h, int64_t num_samples, int64_t* num_frames);
size_t mv_melspec_workspace_bytes(const MvMelSpec* h, int32_t B, int64_t L);
int mv_melspec_forward(const MvMelSpec* h, const float* wav, int32_t B, int64_t L, int64_t wav_stride,
                       const float* lens_ratio, float* out, void* workspace, size_t workspace_bytes,
                       mv_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backbones.  A model handle is built from the reference-layout fp32 ``state_dict`` (same key names
 * and shapes as the reference modules, so ``model.pth`` loads unchanged: mvector/utils/checkpoint.py:
 * 11-51).  create() folds eval-mode BatchNorm into per-channel scale/shift, packs conv weights to the
 * fp16 [Cout][tap][Cin] layout the MFMA kernels read, and keeps fp32 copies of the small dense layers.
 * ------------------------------------------------------------------------------------------------ */
typedef struct MvTensorRef {
    const char* name;   /* state_dict key relative to the backbone, e.g. "blocks.0.conv.conv.weight" */
    const float* data;  /* device pointer, fp32, contiguous */
    int64_t numel;
} MvTensorRef;

typedef struct MvModel MvModel;

/* EcapaTdnn.forward (mvector/models/ecapa_tdnn.py:253-283), pooling_type "ASP". */
typedef struct MvEcapaCfg {
    int32_t input_size;          /* F */
    int32_t embd_dim;            /* 192 */
    int32_t channels[5];         /* {512,512,512,512,1536} */
    int32_t kernel_sizes[5];     /* {5,3,3,3,1} */
    int32_t dilations[5];        /* {1,2,3,4,1} */
    int32_t attention_channels;  /* 128 */
    int32_t res2net_scale;       /* 8 */
    int32_t se_channels;         /* 128 */
    int32_t global_context;      /* 1 */
} MvEcapaCfg;
int mv_ecapa_create(const MvEcapaCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out);

/* CAMPPlus.forward (mvector/models/campplus.py:353-357). */
typedef struct MvCamppCfg {
    int32_t input_size;    /* F (80) */
    int32_t embd_dim;      /* 192 in configs/cam++.yml:58; class default 512 */
    int32_t growth_rate;   /* 32 */
    int32_t bn_size;       /* 4 */
    int32_t init_channels; /* 128 */
    int32_t head_precision; /* MV_CAMPP_HEAD_AUTO (0): decided at create from three probe utterances; _F16 / _F32 pin it -- e.g. the
                             * same value on every rank of a distributed run, so that enrol and verify embeddings share one numerics */
    int32_t xvector_probe;  /* since ABI 5.  MV_CAMPP_XVEC_PROBE_ON (0): create also measures what the fp16 operands of the x-vector part cost on the
                             * three probe utterances (MV_INFO_CAMPP_XVEC_*; ~330 small exact-fp32 launches, once); _OFF (1) skips it */
} MvCamppCfg;
#define MV_CAMPP_XVEC_PROBE_ON 0
#define MV_CAMPP_XVEC_PROBE_OFF 1
#define MV_CAMPP_HEAD_AUTO 0
#define MV_CAMPP_HEAD_F16 1
#define MV_CAMPP_HEAD_F32 2
int mv_campp_create(const MvCamppCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out);

/* TDNN.forward (mvector/models/tdnn.py:46-68), pooling_type "ASP". */
typedef struct MvTdnnCfg {
    int32_t input_size;
    int32_t channels; /* 512 */
    int32_t embd_dim; /* 192 */
} MvTdnnCfg;
int mv_tdnn_create(const MvTdnnCfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out);

/* ERes2Net.forward / ERes2NetV2.forward (mvector/models/eres2net.py:266-287 / 441-456): 2-D Res2Net blocks with AFF
 * fusion, temporal statistics pooling (pooling.py:130-148), seg_1 (+ optional ReLU -> seg_bn_1 -> seg_2). */
typedef struct MvEres2Cfg {
    int32_t version;       /* 1 = ERes2Net, 2 = ERes2NetV2 */
    int32_t input_size;    /* F (80); must be a multiple of 8 */
    int32_t embd_dim;      /* 192 */
    int32_t num_blocks[4]; /* {3,4,6,3} */
    int32_t m_channels;    /* 32 */
    int32_t mul_channel;   /* ERes2Net only: 1 */
    int32_t expansion;     /* 2 */
    int32_t base_width;    /* ERes2Net 32, ERes2NetV2 26 */
    int32_t scale;         /* 2 */
    int32_t two_emb_layer; /* 0 */
} MvEres2Cfg;
int mv_eres2net_create(const MvEres2Cfg* cfg, const MvTensorRef* tensors, int32_t num_tensors, MvModel** out);

int mv_model_destroy(MvModel* m);
int mv_model_embd_dim(const MvModel* m, int32_t* embd_dim);
/* Model-specific facts (tests, logs).  Keys:
 *   MV_INFO_CAMPP_HEAD_F32     1.0 when the CAM++ handle evaluates its FCM head in the exact form (since ABI 3: hi + lo fp16 operand pairs through the
 *                              conv2ds kernels; before: fp32 maps through the conv2d kernels), 0.0 for the fp16 head;
 *   MV_INFO_CAMPP_CALIBRATION  largest 1 - cos between the embeddings of the two heads over the handle's three probe utterances
 *                              (mv_campp_create; -1 when MvCamppCfg.head_precision pinned the head);
 *   MV_INFO_CAMPP_PROBE0 + p   the figure of probe p = 0..2 (white uniform / white bell-shaped / smooth voiced-like). */
#define MV_INFO_CAMPP_HEAD_F32 1
#define MV_INFO_CAMPP_CALIBRATION 2
#define MV_INFO_CAMPP_PROBE0 3
/* since ABI 4 -- the range of the exact head.  Its S16 maps hold 64 * gain * value in fp16 pairs (|.| <= 65504); gain = 2^k (k <= 0) is chosen at
 * create so that the probes' largest map value sits 16 x below that bound (the head is positively homogeneous, so the gain is exact).
 *   MV_INFO_CAMPP_HEAD_GAIN_LOG2  k
 *   MV_INFO_CAMPP_PROBE_PEAK      largest map value over the probes (real units)
 *   MV_INFO_CAMPP_HEAD_PEAK       largest map value the exact head has wanted to store on the caller's inputs since create (real units; waits for the device)
 *   MV_INFO_CAMPP_HEAD_SATURATED  1.0 when one of them exceeded the range and was clamped (the embedding of that call is not to be trusted) */
#define MV_INFO_CAMPP_HEAD_GAIN_LOG2 6
/* since ABI 5 -- what the head probes cannot see (campplus.py:295-357 behind the head): the x-vector part always runs on fp16 operands.  create
 * evaluates it once more in exact fp32 on the three probes (same head rows, the caller's fp32 weights, exact-fp32 GEMMs) and reports
 *   MV_INFO_CAMPP_XVEC_SENSITIVITY   largest 1 - cos between the shipped and the exact x-vector part over the probes (-1: probe switched off)
 *   MV_INFO_CAMPP_XVEC_PROBE0 + p    probe p's figure.
 * Above MV_CAMPP_XVEC_WARN the 1e-4 contract (1 - cos against the reference's fp32 forward) is at risk on this checkpoint whatever head runs: there
 * is no exact form of the x-vector part; the Python surface warns (mvector/models/campplus.py). */
#define MV_INFO_CAMPP_XVEC_SENSITIVITY 10
#define MV_INFO_CAMPP_XVEC_PROBE0 11
#define MV_CAMPP_XVEC_WARN 2.5e-5f
#define MV_INFO_CAMPP_PROBE_PEAK 7
#define MV_INFO_CAMPP_HEAD_PEAK 8
#define MV_INFO_CAMPP_HEAD_SATURATED 9
int mv_model_info(const MvModel* m, int32_t key, float* value);
int mv_model_workspace_bytes(const MvModel* m, int32_t B, int32_t T, size_t* bytes);
/* feats: [B, T, F] fp32 (the AudioFeaturizer output layout); emb: [B, embd_dim] fp32. */
int mv_model_forward(const MvModel* m, const float* feats, int32_t B, int32_t T, float* emb, void* workspace,
                     size_t workspace_bytes, mv_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Cosine scoring: S[i, j] = <a_i, b_j> / (|a_i| |b_j|).  Replaces sklearn cosine_similarity at
 * mvector/predict.py:174 and mvector/trainer.py:457-459, and predict.py:275-279 (contrast).
 * ------------------------------------------------------------------------------------------------ */
int mv_cosine_f32(const float* a, int32_t n, const float* b, int32_t m, int32_t dim, float* scores,
                  mv_stream_t stream);
/* in-place row L2 normalisation (predict.py:165-166) */
int mv_l2_normalize_f32(float* x, int32_t n, int32_t dim, mv_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layer-level entry points (same kernels the model forwards launch; exported so that the parity
 * tests can pin each kernel against the oracle layer by layer).
 * ------------------------------------------------------------------------------------------------ */
#define MV_PAD_ZERO 0
#define MV_PAD_REFLECT 1
#define MV_ACT_NONE 0
#define MV_ACT_RELU 1
#define MV_ACT_TANH 2
#define MV_ACT_SIGMOID 3
#define MV_DT_F32 0
#define MV_DT_F16 1

/* ------------------------------------------------------------------------------------------------
 * 2-D convolution layer of the ERes2Net family (eres2net.py:23-30: conv1x1 / conv3x3, zero padding k/2, no conv
 * bias) on channel-last maps [B, H = frequency, W = time, C], channels padded to multiples of 16 -- the semantics of
 * mv_conv2ds_forward below (split fp16 operands on S16 maps: the family is too deep for plain fp16 operands at the 1e-4 cosine
 * bar) and of the fp32-operand yard-stick tools/yardstick/conv2d_f32.hip:
 *   y = epi( sum_taps W . in + bias ),  in = x | x + x2 (eres2net.py:92) | cat(x, x2) (AFF, eres2net.py:49)
 *   epi 0: clamp(v [+ res], lo, hi)   -- BatchNorm folded into W / bias, ReLU(0..20) eres2net.py:12-15, residual :103-105
 *   epi 1: SiLU (AFF local_att, eres2net.py:41)
 *   epi 2: res * (1 + tanh v) + res2 * (1 - tanh v)   (AFF output, eres2net.py:50-52)
 * ------------------------------------------------------------------------------------------------ */
#define MV_EPI_CLAMP 0
#define MV_EPI_SILU 1
#define MV_EPI_AFF 2
/* (The fp32-operand form of these layers -- MvConv2dDesc, mv_conv2d_pack_weight / _packed_elems / _forward, ABI 1-3 -- left the library with ABI 4: no
 * model handle had run it since round 4; it is the exact-fp32 yard-stick under tools/yardstick/.) */
/* first ERes2Net conv: features fp32 [B, T, F] -> fp32 [B, F, T, C] = relu(conv3x3(1 -> C) + bias), w fp32 [C][9] */
int mv_conv2d_first(const float* feats, float* out, const float* w, const float* bias, int32_t B, int32_t T, int32_t F,
                    int32_t C, mv_stream_t stream);
/* temporal statistics pooling of fp32 [B, H, W, ld] (C real channels) -> fp32 [B, 2*C*H]: mean | sqrt(unbiased var + 1e-8),
 * index c*H + h as the reference flattens [B, C, H] */
int mv_tstp_f32(const float* x, int64_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* stats, mv_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The same layers on the fp16 matrix pipe with split operands (csrc/conv2ds.hip; what mv_eres2net_create's handles run since ABI 3).
 * Maps are "S16": channel-last with 4 bytes per channel like the fp32 maps above (same pointers, leading dimensions and
 * 16-channel slices), but a unit of 16 channels holds [16 x hi | 16 x lo] fp16 of 64 * value (hi = fp16(V), lo = fp16(V - hi)):
 * 22 significant bits, |value| saturates at 1023.5.  mv_map_split_f32 / mv_map_merge_f32 convert n = pixels * ld fp32
 * elements (whole units) to and from that form.  Weights: mv_conv2ds_pack_weight scales a layer by a power of two, splits it the
 * same way into [cout16][k*k][round_up(cin16, 32) / 16 units][32] (mv_conv2ds_packed_elems 4-byte elements) and returns the factor
 * `oscale` the accumulator is multiplied with (synchronises the stream: weights are read back to the host).
 *   y = epi( oscale * sum_taps W . in + bias ),  in = x | cat(x, x2)  (x2 != NULL: cin1 channels from x, the rest from x2)
 *   epi as above; optionally y2 = y + add (the input of the next 3x3 conv of a Res2Net block, eres2net.py:92)
 * ------------------------------------------------------------------------------------------------ */
int64_t mv_conv2ds_packed_elems(int32_t cout, int32_t cin, int32_t ks);
int mv_conv2ds_pack_weight(const float* w, const float* out_scale, int32_t cout, int32_t cin, int32_t ks, void* packed, float* oscale,
                           mv_stream_t stream);
int mv_map_split_f32(const float* x, void* y, int64_t n, mv_stream_t stream);
int mv_map_merge_f32(const void* x, float* y, int64_t n, mv_stream_t stream);
typedef struct MvConv2dsDesc {
    const void* x;       /* S16 [B, H, W, ldx] */
    const void* x2;      /* optional: channels cin1 .. cin16 of the input come from here (AFF concatenation, eres2net.py:49) */
    int32_t cin1;
    int64_t ldx, ldx2;
    const void* w;       /* from mv_conv2ds_pack_weight */
    const float* bias;   /* [cout16] */
    float oscale;        /* from mv_conv2ds_pack_weight */
    const void* res;     /* epi 0: optional residual; epi 2: first AFF operand; S16 [B, Ho, Wo, ldres] */
    const void* res2;    /* epi 2: second AFF operand */
    int64_t ldres, ldres2;
    const void* add;     /* optional with y2: S16 [B, Ho, Wo, ldadd] */
    int64_t ldadd;
    void* y;             /* S16 [B, Ho, Wo, ldy], Ho = (H + 2*(ks/2) - ks)/stride + 1, Wo likewise */
    int64_t ldy;
    void* y2;            /* optional second output y + add */
    int64_t ldy2;
    int32_t B, H, W, cin16, cout16, ks, stride, epi;
    float lo, hi;
    int32_t cin_alg, cout_alg; /* channel counts of the layer before padding (0: same as cin16 / cout16): profile accounting only */
    int32_t stride_w;          /* stride along W when it differs from `stride` (then the stride along H); 0 = same (the CAM++ head strides the
                                * frequency axis only, campplus.py:221-292) */
    uint32_t* peak;            /* optional device word (since ABI 4): the kernel max-es in the bits of the largest |64 * value| it wanted to store BEFORE the
                                * clamp to the fp16 range (positive floats order like unsigned integers); >= 65504.0f means the S16 split saturated */
    int32_t nbw_hint, ct_hint, rows_hint, ring_hint, wgs_hint, spw_hint, nprod_hint; /* 0 = the launcher's choice; otherwise blocks of 16 output channels per wave
                                           * (1..3), blocks per workgroup, rows per 3x3 tile (1..8), LDS ring stages (>= 2), workgroups per CU
                                           * (1 | 2), segments per consumer wave (8 | 4 | 2 | 1), producer waves: launch shapes for tests and tools/bench_conv2d.py (never the bits of a result) */
} MvConv2dsDesc;
int mv_conv2ds_forward(const MvConv2dsDesc* d, mv_stream_t stream);
/* the first conv and the pooling with S16 maps on the map side */
int mv_conv2d_first_s16(const float* feats, void* out, const float* w, const float* bias, int32_t B, int32_t T, int32_t F,
                        int32_t C, mv_stream_t stream);
int mv_tstp_s16(const void* x, int64_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* stats, mv_stream_t stream);

/* pack [Cout][Cin][k] fp32 (nn.Conv1d layout) -> fp16 [Cout_pad][k][Cin_pad]; returns element count */
int64_t mv_conv1d_packed_elems(int32_t cout, int32_t cin, int32_t k);
int mv_conv1d_pack_weight(const float* w, int32_t cout, int32_t cin, int32_t k, void* packed_f16, mv_stream_t stream);

typedef struct MvConv1dDesc {
    /* y[b, t, co] = epilogue( sum_{j, ci} W[co, j, ci] * in(b, t*stride - pad + j*dilation, ci) )
     * in(.) = x (+ x2) optionally passed through relu(x*in_scale+in_shift); channel-last tensors. */
    const void* x;        /* [B, T_in, ldx] */
    const void* x2;       /* optional second input added to x (Res2Net: ecapa_tdnn.py:47), same ld/dtype */
    int32_t x_dtype;      /* MV_DT_F32 | MV_DT_F16 */
    int64_t ldx;          /* elements between consecutive time steps of x (>= cin) */
    int64_t ldx2;
    const float* in_scale; /* optional [cin]: pre-activation BatchNorm folded (campplus.py:139-141) */
    const float* in_shift;
    const void* w_packed; /* from mv_conv1d_pack_weight */
    const float* bias;    /* optional [cout] */
    const float* row_bias;/* optional [B, cout]: per-utterance bias (hoisted ASP context, pooling.py:110-117) */
    int32_t pre_act;      /* activation applied to (acc + bias) BEFORE the affine (TDNNBlock: ReLU, models/utils.py:138) */
    const float* scale;   /* optional [cout] folded BatchNorm scale */
    const float* shift;
    int32_t post_act;     /* activation after the affine */
    const float* gate;    /* optional [B, n_seg, cout] multiplicative gate (CAM mask, campplus.py:94-99) */
    int32_t gate_seg_len; /* frames per gate segment */
    void* y;              /* [B, T_out, ldy] */
    int32_t y_dtype;
    int64_t ldy;
    const void* add_src;  /* optional fp16 [B, T_out, ld_add]: second output sum_dst = y + add_src, i.e. the input */
    void* sum_dst;        /* x_{j+1} + y_j of the next Res2Net step (ecapa_tdnn.py:47) produced by this step's epilogue */
    int64_t ld_add, ld_sum;
    int32_t B, T_in, T_out, cin, cout, k, dilation, stride, pad, pad_mode;
    int32_t tile;         /* workgroup tile: 0 = choose (256x256 for wide layers that fill the chip, 64x64 for small problems, else 128x128 / 128x160), 64, 128, 160, 256 */
    /* optional fused time statistics of y (SE squeeze ecapa_tdnn.py:79, ASP global mean / std pooling.py:104-109): fp32 partial
     * buffers of mv_conv1d_stats_elems(B, T_out, cout) floats each; stat_sq may be NULL (mean only).  Only on the persistent
     * 1x1 path: fp16 in/out, cout % 256 == 0, cin % 64 == 0, no row_bias / gate, T_out >= 64.  Finish with
     * mv_conv1d_stats_finish. */
    float* stat_sum;
    float* stat_sq;
    /* optional fused time statistics of the INPUT x (the ASP global mean / std, pooling.py:104-109, taken from the x tiles the hidden
     * 1x1 conv of the attention streams through LDS anyway): fp32 partial buffers of mv_conv1d_in_stats_elems(B, T_in, cin) floats
     * each (sums and sums of squares per (utterance, 160-frame tile, channel)).  Only on the direct fp16 path with the 128 x 160 tile:
     * k = 1, stride 1, no padding, T_out == T_in, cout <= 128, no x2 / in_scale.  Finish with mv_conv1d_in_stats_finish. */
    float* in_stat_sum;
    float* in_stat_sq;
    int32_t persist_blocks_hint;  /* 0 = one resident workgroup per CU; otherwise that many (rounded up to 8) workgroups walk the tiles of the
                                   * persistent kernels -- tests make small problems walk several tiles per workgroup; never the bits of a result
                                   * (since ABI 4; replaces the MV_CONV_PERSIST_BLOCKS environment hook: no getenv is left in the library) */
    uint64_t* clock_probe;        /* since ABI 5, optional: device array [resident workgroups (<= CUs rounded up to 8)][4].  The dense 1x1 persistent
                                   * (ring) kernel leaves {shader-clock cycles at entry, at exit, 100 MHz reference at entry, at exit} of every
                                   * workgroup there: (exit - entry cycles) / (exit - entry reference ticks) x 100 MHz = the shader clock the launch
                                   * SUSTAINED (bench.py's `box` block prices the 2.5 PFLOP/s peak's 2.4 GHz against it).  Other kernels ignore it. */
} MvConv1dDesc;
int mv_conv1d_forward(const MvConv1dDesc* d, mv_stream_t stream);
/* floats in one partial-statistics buffer, and the reduction of the partial rows to per-utterance mean[b, c] (and
 * std[b, c] = sqrt(max(E[(y - mean)^2], clamp_eps)) when stat_sq / std are given).  `shift` = the BatchNorm shift passed to the
 * conv (the moments are taken about it), NULL if the conv had none. */
int64_t mv_conv1d_stats_elems(int32_t B, int32_t T_out, int32_t cout);
/* per-utterance mean[b, c] and std[b, c] = sqrt(max(E[x^2] - mean^2, clamp_eps)) of the conv INPUT from the in_stat_* partials */
int64_t mv_conv1d_in_stats_elems(int32_t B, int32_t T_in, int32_t cin);
int mv_conv1d_in_stats_finish(const float* in_stat_sum, const float* in_stat_sq, int32_t B, int32_t T_in, int32_t cin, float* mean,
                              float* std, int64_t ld_out, float clamp_eps, mv_stream_t stream);
int mv_conv1d_stats_finish(const float* stat_sum, const float* stat_sq, const float* shift, int32_t B, int32_t T_out, int32_t cout,
                           float* mean, float* std, int64_t ld_out, float clamp_eps, mv_stream_t stream);

/* Fused Res2Net chain (mvector/models/ecapa_tdnn.py:39-51): x, y fp16 [B, T, C]; `groups` channel groups of width C/groups;
 * y[..., 0:w] = x[..., 0:w];  y_j = BN(ReLU(conv_j(x_j + y_{j-1}))) for j = 1..groups-1 with reflect "same" padding.
 * Arrays of groups-1 pointers (host arrays of device pointers): packed weights, bias, folded BN scale / shift.
 * One workgroup per utterance up to 320 frames; longer utterances are cut into chunks of equal length that carry the chain's receptive field
 * ((groups-1) * dilation * (k-1)/2 frames) as halo on either side, one workgroup each.  Returns MV_ERR_UNSUPPORTED unless width is 64 or 128. */
int mv_res2net_chain_f16(const void* x, void* y, const void* const* w_packed, const float* const* bias,
                         const float* const* scale, const float* const* shift, int32_t B, int32_t T, int32_t C,
                         int32_t groups, int32_t k, int32_t dilation, mv_stream_t stream);

/* y[b, o] = act( sum_k x[b, k] * w[o, k] + bias[o] ) in exact fp32 (f32 MFMA).
 * ONE summation order per ENTRY POINT, not across them: this call sums K in one sweep (the direct kernel); mv_linear_f32_ws with a sufficient
 * workspace sums layers of K >= 2048 as slices of 384 added in slice order (what the model handles run) -- the two differ in the last bits
 * (~1e-7 relative) on such layers, and mv_linear_f32_ws with a NULL / short workspace IS this call.  "A row's bits depend on the row only"
 * holds within either form for every batch size; a caller that compares bits across the two forms must pick one. */
int mv_linear_f32(const float* x, int64_t ldx, const float* w, const float* bias, int32_t act, float* y, int64_t ldy,
                  int32_t B, int32_t K, int32_t O, mv_stream_t stream);
/* The same layer with a caller workspace (ABI 4): long reductions (K >= 2048, any number of rows: the [256, 6144] x [192, 6144] final layer of
 * EcapaTdnn-1024) run as K slices of 384 on 32 x 32 output tiles + a second launch that adds the slices' partial sums in slice order -- a fixed order,
 * so a row's bits depend on the row only.  mv_linear_f32_workspace_floats = floats that form needs (0: the direct kernel runs either way); with a
 * NULL or short workspace the call is mv_linear_f32. */
size_t mv_linear_f32_workspace_floats(int32_t B, int32_t K, int32_t O);
int mv_linear_f32_ws(const float* x, int64_t ldx, const float* w, const float* bias, int32_t act, float* y, int64_t ldy, int32_t B, int32_t K, int32_t O,
                     float* workspace, size_t workspace_floats, mv_stream_t stream);

/* mean (and optionally std = sqrt(clamp(E[(x-mean)^2], eps)), or unbiased std) over time of a
 * channel-last fp16 tensor [B, T, ld] -> fp32 [B, C].  std may be NULL. */
int mv_time_stats_f16(const void* x, int64_t ld, int32_t B, int32_t T, int32_t C, float* mean, float* std,
                      int32_t unbiased, float clamp_eps, mv_stream_t stream);

/* Pre-activation written out once: y[n, c] = fp16(relu(x[n, c] * scale[c] + shift[c])) over channel-last fp16 rows (eval BatchNorm
 * folded to scale / shift + ReLU in front of a 1x1 conv: the CAM++ transit layers, mvector/models/campplus.py:186-189), so that the
 * conv behind it takes the direct global -> LDS path instead of transforming on load.  C % 8 == 0, 16-byte aligned rows / parameters. */
int mv_bn_relu_rows_f16(const void* x, int64_t ldx, const float* scale, const float* shift, void* y, int64_t ldy, int64_t n_rows,
                        int32_t C, mv_stream_t stream);

/* Per-kernel-class timing with HIP events recorded on the launch stream (used by bench.py for its roofline legs; off by
 * default, not thread-safe).  work = algorithmic FLOPs (MV_PROF_CONV1D: 2*B*T_out*cin*cout*k per launch) or algorithmic
 * bytes (MV_PROF_FBANK: B*(4*L + 4*T*num_mel_bins) per launch).  mv_profile_read waits for the recorded launches. */
#define MV_PROF_CONV1D 0
#define MV_PROF_FBANK 1
#define MV_PROF_CONV2D 2 /* work = 2*B*Ho*Wo*cin*cout*ks*ks FLOPs on the layer's own (unpadded) channel counts */
#define MV_PROF_CONV1D_RING 3 /* since ABI 5: the launches of the dense 1x1 persistent (ring) GEMM alone -- a subset of MV_PROF_CONV1D, which keeps counting them */
int mv_profile_enable(int32_t on);
int mv_profile_read(int32_t kernel_class, int32_t* calls, double* total_ms, double* total_work, int32_t reset);

/* int16 PCM rows [B, L] -> float32 waveforms in [-1, 1) (sample / 32768) on the device, zero beyond num_samples[b] (device
 * int64 array, NULL = L everywhere).  normalize != 0: per-row dB normalisation over the true length, gain =
 * 10^((target_db - 10 log10(mean x^2)) / 20) -- AudioSegment.normalize as called at mvector/predict.py:210-211; rows whose gain
 * would exceed max_gain_db (digital silence; the reference raises ValueError) are left unscaled and flagged in too_quiet[b]
 * (device int32 array, may be NULL). */
int mv_wave_prepare_i16(const int16_t* pcm, int64_t pcm_stride, const int64_t* num_samples, int32_t B, int64_t L, int32_t normalize,
                        float target_db, float max_gain_db, float* wav, int64_t wav_stride, int32_t* too_quiet, mv_stream_t stream);

/* Attentive-statistics pooling tail (mvector/models/pooling.py:117-125): attention logits = W2 . h (the bias of the
 * projection is constant over time and cancels in the softmax over time), softmax over time, weighted mean / std.
 *   h  fp16 [B, T, A] (tanh output, |h| <= 1),  w2_packed = mv_conv1d_pack_weight of  W2[C, A, 1] * log2(e),
 *   x  fp16 [B, T, ldx],  gmean fp32 [B, gmean_ld] or NULL (centre of the second moment),  out fp32 [B, 2C] = mean | std.
 * logit_bound_log2 = max_c sum_k |W2[c,k]| * log2(e) selects the form without max subtraction when it is in [0, 60];
 * pass a negative value for the online-softmax form (any weights). */
int mv_asp_pool_f16(const void* h, const void* w2_packed, const void* x, int64_t ldx, const float* gmean, int64_t gmean_ld,
                    float* out, int32_t B, int32_t T, int32_t C, int32_t A, float logit_bound_log2, mv_stream_t stream);

/* One 3x3 Conv2d of the CAM++ front-end (FCM, mvector/models/campplus.py:221-292: BasicResBlock.conv1 / conv2 (+ shortcut) and
 * FCM.conv2) on channel-last fp16 maps with 32 feature maps, eval BatchNorm already folded into w / bias, ReLU at the end:
 *   x  fp16 [B, Fin, T, 32];  stride sf (1 | 2) on the frequency axis, zero padding 1;  y fp16, element (b, fo, t, co) at
 *   y + b * y_sB + fo * y_sF + t * y_sT + co (strides in elements);  Fout = (Fin - 1) / sf + 1;
 *   w  fp16 [9 or 10][32 co][32 ci], tap = 3 * df + dt;  bias fp32 [32];
 *   mode2 = 0: none;  1: tenth tap = the block's strided 1x1 shortcut conv (+BN) on x2 [B, F2, T, 32] at row fo * sf2;
 *   2: identity residual, x2 added before the ReLU.  Layer-level entry point of the parity tests. */
int mv_fcm_conv3x3_f16(const void* x, int32_t Fin, int32_t sf, const void* x2, int32_t F2, int32_t sf2, int32_t mode2, const void* w,
                       const float* bias, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, int32_t Fout,
                       mv_stream_t stream);

/* One BasicResBlock of the CAM++ front-end (mvector/models/campplus.py:221-254) as ONE launch:
 *   mid = ReLU(BN1(conv3x3, stride (sf, 1))(x)),  y = ReLU(BN2(conv3x3)(mid) + shortcut(x));  the intermediate map stays on the chip.
 *   x fp16 [B, Fin, T, 32];  Fout = (Fin - 1) / sf + 1;  y element (b, fo, t, co) at y + b * y_sB + fo * y_sF + t * y_sT + co;
 *   w1 fp16 [9][32 co][32 ci] (tap = 3 * df + dt, BN1 folded), b1 fp32 [32];  w2 fp16 [9 | 10][32][32] (BN2 folded), b2 fp32 [32];
 *   shortcut != 0: tap 9 of w2 is the block's strided 1x1 shortcut conv with its BatchNorm folded (its shift is part of b2);
 *   shortcut == 0: identity residual (sf must be 1). */
int mv_fcm_block_f16(const void* x, int32_t Fin, int32_t sf, const void* w1, const float* b1, const void* w2, const float* b2,
                     int32_t shortcut, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, mv_stream_t stream);

/* The FIRST BasicResBlock of the CAM++ front-end together with the conv in front of it (campplus.py:262-264,283-285: head.conv1 + bn1 +
 * ReLU, then head.layer1[0], stride (2, 1) with its shortcut conv) as ONE launch: the 32-map image of the features is evaluated inside the
 * kernel and never written.  feats fp32 [B, T, F];  c1a = mv_fcm_c1_pack of the folded conv weights, c1b fp32 [32] = the folded BN shift;
 * the other arguments as mv_fcm_block_f16 with sf = 2, shortcut = 1.  The conv weights act as fp16 (like every other conv of the fp16
 * head), the features exactly (as x_hi + x_lo). */
int mv_fcm_block_c1_f16(const float* feats, int32_t F, const void* c1a, const float* c1b, const void* w1, const float* b1, const void* w2,
                        const float* b2, void* y, int64_t y_sB, int64_t y_sF, int64_t y_sT, int32_t B, int32_t T, mv_stream_t stream);
/* host -> host: w fp32 [32 maps][3 mel taps][3 time taps] (BatchNorm scale folded) -> out fp16 [2][64][8], the MFMA operand order of the kernel */
int mv_fcm_c1_pack(const float* w, void* out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MVECTOR_HIP_H */
