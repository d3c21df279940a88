"""TEST INFRASTRUCTURE ONLY — CPU oracle for the host-input pipeline:
the arithmetic of `get_self_inputs` (llm_trainer.py:306-381).  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this; macaw_llm_amd/ never does.

The reference delegates this arithmetic to packages that are not vendored under
/root/reference and (partly) not installed here:

  * torchvision (requirements.txt, NOT installed): Compose([Resize(224, BICUBIC),
    CenterCrop(224), convert RGB, ToTensor(), Normalize(mean, std)]) llm_trainer.py:150-157.
    Resize on a PIL image calls `PIL.Image.resize` -> Pillow's libImaging/Resample.c.
    Pillow 12.2.0 IS installed, so `pil_transform` below runs the real resampler and restates
    only torchvision's size / crop arithmetic (functional.resize / center_crop) and
    ToTensor / Normalize (float32 div 255, sub mean, div std).
  * `resample_restated` additionally restates Resample.c itself (precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc) in numpy; it is PINNED
    bit-for-bit against Pillow in tests/test_oracle_preprocess.py so the algorithm the HIP
    kernel implements is the one Pillow runs.
  * openai-whisper (requirements.txt, NOT installed): `log_mel_spectrogram`, `pad_or_trim`
    (whisper/audio.py).  `log_mel_whisper_fp32` restates it line by line on torch CPU fp32
    (torch.stft 400/160 hann, |.|^2 minus last frame, mel_filters @, log10 clamp, max - 8,
    (x + 4) / 4); `log_mel_f64` is the same in exact-ish float64 numpy.  The mel filterbank
    whisper ships as an .npz is `librosa.filters.mel(sr=16000, n_fft=400, n_mels=80)`;
    PINNED against transformers' WhisperFeatureExtractor (installed; its numpy implementation
    of the same published algorithm, mel_filter_bank(norm="slaney", mel_scale="slaney")).
"""
from __future__ import annotations

import math

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PB = 22


# ------------------------------------------------------------------ image --
def tv_resized_size(w, h, size):
    """torchvision.transforms.functional._compute_resized_output_size for an int size"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def pil_transform(img_u8: np.ndarray, n_px: int = 224) -> torch.Tensor:
    """_transform(n_px)(PIL.Image.fromarray(img_u8)) -> float32 [3, n_px, n_px]"""
    from PIL import Image
    im = Image.fromarray(img_u8)
    w, h = im.size
    nw, nh = tv_resized_size(w, h, n_px)
    im = im.resize((nw, nh), Image.BICUBIC)
    top = int(round((nh - n_px) / 2.0))
    left = int(round((nw - n_px) / 2.0))
    im = im.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous()
    t = t.to(torch.float32).div(255)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(-1, 1, 1)
    return t.sub_(mean).div_(std)


def pil_crop_u8(img_u8: np.ndarray, n_px: int = 224) -> np.ndarray:
    """the uint8 pixels after Resize + CenterCrop (before ToTensor), HWC"""
    from PIL import Image
    im = Image.fromarray(img_u8)
    w, h = im.size
    nw, nh = tv_resized_size(w, h, n_px)
    im = im.resize((nw, nh), Image.BICUBIC)
    top = int(round((nh - n_px) / 2.0))
    left = int(round((nw - n_px) / 2.0))
    return np.asarray(im.crop((left, top, left + n_px, top + n_px))).copy()


def _bicubic(x, a=-0.5):
    if x < 0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size, out_size):
    """Resample.c precompute_coeffs (bicubic, full box) + normalize_coeffs_8bpc"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PB)) if v < 0 else int(0.5 + v * (1 << PB))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_axis0(img, out_size):
    kk, b = _coeffs(img.shape[0], out_size)
    out = np.zeros((out_size,) + img.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = b[xx]
        ss = (img[xmin:xmin + n].astype(np.int64) * kk[xx, :n, None, None]).sum(0) + (1 << (PB - 1))
        out[xx] = np.clip(ss >> PB, 0, 255)
    return out


def resample_restated(img_u8: np.ndarray, w: int, h: int) -> np.ndarray:
    """ImagingResample(BICUBIC): horizontal pass, 8-bit intermediate, vertical pass; a pass
    whose size does not change is skipped"""
    H, W, _ = img_u8.shape
    img = img_u8
    if w != W:
        img = _resample_axis0(img.transpose(1, 0, 2), w).transpose(1, 0, 2)
    if h != H:
        img = _resample_axis0(img, h)
    return img


def synthetic_image(seed: int, H: int, W: int) -> np.ndarray:
    """deterministic test picture: smooth structure + edges + noise (exercises clipping)"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    base = np.stack([127 + 120 * np.sin(x / 9.0 + seed) * np.cos(y / 13.0),
                     127 + 120 * np.sin((x + y) / 17.0),
                     255.0 * ((x // 16 + y // 16) % 2)], -1)
    img = base + rng.normal(0, 25, (H, W, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ audio --
N_FFT, HOP, N_SAMPLES = 400, 160, 480000


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)),
                    3.0 * f / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


def mel_filters(n_mels=80, sr=16000, n_fft=N_FFT):
    """librosa.filters.mel(sr, n_fft, n_mels) (htk=False, norm='slaney'), float32"""
    fft_freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).astype(np.float32)


def pad_or_trim(x: np.ndarray, length=N_SAMPLES):
    if x.shape[-1] > length:
        return x[..., :length]
    if x.shape[-1] < length:
        return np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, length - x.shape[-1])])
    return x


def log_mel_whisper_fp32(x: np.ndarray, n_mels=80) -> torch.Tensor:
    """whisper.log_mel_spectrogram, line by line, torch CPU float32 (the reference's precision)"""
    audio = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(mel_filters(n_mels))
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def log_mel_f64(x: np.ndarray, n_mels=80) -> np.ndarray:
    """the same algorithm in float64 (window kept at its float32 values)"""
    n = np.arange(N_FFT)
    win = torch.hann_window(N_FFT).numpy().astype(np.float64)
    xp = np.pad(x.astype(np.float64), (N_FFT // 2, N_FFT // 2), mode="reflect")
    nfr = 1 + (len(xp) - N_FFT) // HOP
    fr = xp[np.arange(nfr)[:, None] * HOP + n[None, :]] * win
    sp = np.fft.rfft(fr, axis=1)[:-1]
    mel = (sp.real ** 2 + sp.imag ** 2) @ mel_filters(n_mels).astype(np.float64).T
    ls = np.log10(np.maximum(mel, 1e-10))
    ls = np.maximum(ls, ls.max() - 8.0)
    return ((ls + 4.0) / 4.0).T


def synthetic_audio(seed: int, seconds: float = 30.0) -> np.ndarray:
    """deterministic PCM: decaying chirp + tone bursts + noise floor, float32 in [-1, 1]"""
    rng = np.random.default_rng(seed)
    n = int(seconds * 16000)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * (200 + 150 * t) * t) * np.exp(-t / 12.0)
    x += 0.2 * np.sin(2 * np.pi * 3100 * t) * (np.sin(2 * np.pi * 0.7 * t) > 0.3)
    x += 0.005 * rng.standard_normal(n)
    return np.clip(x, -1, 1).astype(np.float32)
