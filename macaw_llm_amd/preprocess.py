"""GPU input pipeline: the per-step host work of the reference's `get_self_inputs`
(llm_trainer.py:306-381) -- CLIP `_transform(224)` on 1 image + 6 video frames per sample and
`whisper.log_mel_spectrogram` on 30 s of audio, all synchronous on the trainer's main thread --
done by two HIP kernels (csrc/preprocess.hip) on decoded pixels / PCM already on the device.
File decoding (JPEG via PIL, audio via ffmpeg) stays on the host: it is not arithmetic we own.

Same call surface as the reference pieces:

    preprocess = ImageTransform(224, device)      # llm_trainer.py:150-157,168  `_transform`
    frames = preprocess([img0, img1, ...])        # uint8 HWC arrays / PIL RGB -> [N,3,224,224]
    mel = log_mel_spectrogram(pad_or_trim(audio)) # whisper.audio API -> [.., 80, 3000]

Numerics: the image path is integer work and bit-exact with PIL + torchvision; the mel path
accumulates the DFT in fp64 and agrees with the fp32 reference to ~3e-5 absolute (the
reference's own FFT rounding), tests/test_preprocess_*.py.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import lib as _L
from . import ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # llm_trainer.py:156
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2                           # Pillow libImaging/Resample.c

# ------------------------------------------------------------------ images --


def _bicubic(x: float, a: float = -0.5) -> float:
    """Pillow's bicubic kernel (Keys, a = -0.5), support 2"""
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=512)
def resample_coeffs(in_size: int, out_size: int, first: int, count: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow `precompute_coeffs` + `normalize_coeffs_8bpc` for output positions
    [first, first + count) of an in_size -> out_size bicubic resize.  Evaluated in Python
    floats (IEEE double, the type and operation order Pillow uses) so the integer taps are
    identical.  Returns (kk int32 [count, ksize], bounds int32 [count, 2] = first tap, n taps).
    in_size == out_size is the pass Pillow skips: identity taps."""
    if in_size == out_size:
        kk = np.full((count, 1), 1 << PRECISION_BITS, np.int32)
        bounds = np.stack([np.arange(first, first + count), np.ones(count, np.int64)], 1).astype(np.int32)
        return kk, bounds
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((count, ksize), np.int32)
    bounds = np.zeros((count, 2), np.int32)
    ss = 1.0 / filterscale
    for i in range(count):
        center = (first + i + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x, v in enumerate(w):
            if ww != 0.0:
                v = v / ww
            kk[i, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[i] = (xmin, xmax)
    return kk, bounds


def resized_size(w: int, h: int, size: int) -> Tuple[int, int]:
    """torchvision Resize(int): shorter side -> size, longer = int(size * long / short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def _as_rgb_u8(img) -> np.ndarray:
    if hasattr(img, "mode") and hasattr(img, "size"):      # PIL image
        if img.mode != "RGB":
            # the reference resizes in the source mode and converts afterwards; only RGB inputs
            # are pixel-identical on this path
            raise ValueError(f"ImageTransform takes RGB images (got PIL mode {img.mode!r})")
        img = np.asarray(img)
    if torch.is_tensor(img):
        img = img.cpu().numpy()
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected uint8 [H, W, 3] RGB, got {a.dtype} {a.shape}")
    return a


class ImageTransform:
    """`_transform(n_px)` of llm_trainer.py:150-157 for a batch, on the GPU."""

    def __init__(self, n_px: int = 224, device="cuda", dtype=torch.float32,
                 mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD,
                 via_half: bool = False):
        self.n_px, self.device, self.dtype = n_px, torch.device(device), dtype
        # ToTensor: byte.to(float32).div(255); Normalize: sub_(mean).div_(std) -- evaluated
        # once for all 256 byte values with the reference's own float32 ops
        v = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
        lut = torch.stack([(v - torch.tensor(m, dtype=torch.float32)) / torch.tensor(s, dtype=torch.float32)
                           for m, s in zip(mean, std)])
        if via_half:   # the trainer's `.half()` (llm_trainer.py:366-368) before the model's own cast
            lut = lut.half().float()
        self.lut_cpu = lut.contiguous()
        self._lut_dev = None

    def plan(self, sizes: Sequence[Tuple[int, int]]):
        """host-side geometry for images of (H, W): descriptors + coefficient table"""
        n = self.n_px
        descs = np.zeros((len(sizes), 12), np.int64)
        coef: List[np.ndarray] = []
        coff = 0
        src_off = tmp_off = 0
        max_rows = 1
        for i, (H, W) in enumerate(sizes):
            new_w, new_h = resized_size(W, H, n)
            top = int(round((new_h - n) / 2.0))       # torchvision center_crop
            left = int(round((new_w - n) / 2.0))
            hk, hb = resample_coeffs(W, new_w, left, n)
            vk, vb = resample_coeffs(H, new_h, top, n)
            row0 = int(vb[:, 0].min())
            nrows = int((vb[:, 0] + vb[:, 1]).max()) - row0
            offs = []
            for arr in (hk, hb, vk, vb):
                offs.append(coff)
                coef.append(arr.reshape(-1))
                coff += arr.size
            descs[i] = (src_off, H, W, tmp_off, row0, nrows, offs[0], offs[1], hk.shape[1], offs[2],
                        offs[3], vk.shape[1])
            src_off += H * W * 3
            tmp_off += nrows * n * 3
            max_rows = max(max_rows, nrows)
        return descs, np.concatenate(coef).astype(np.int32), src_off, tmp_off, max_rows

    def __call__(self, images) -> torch.Tensor:
        if not isinstance(images, (list, tuple)):
            images = [images]
        arrs = [_as_rgb_u8(im) for im in images]
        if self.device.type != "cuda":
            raise ops.MacawHipError("ImageTransform runs on the HIP device only (no CPU fallback)")
        descs, coef, src_bytes, tmp_bytes, max_rows = self.plan([a.shape[:2] for a in arrs])
        host = torch.empty(src_bytes, dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        o = 0
        for a in arrs:
            hv[o:o + a.size] = a.reshape(-1)
            o += a.size
        dev = self.device
        src = host.to(dev, non_blocking=True)
        d_desc = torch.from_numpy(descs).to(dev, non_blocking=True)
        d_coef = torch.from_numpy(coef).to(dev, non_blocking=True)
        if self._lut_dev is None:
            self._lut_dev = self.lut_cpu.to(dev)
        tmp = torch.empty(max(tmp_bytes, 1), dtype=torch.uint8, device=dev)
        out = torch.empty((len(arrs), 3, self.n_px, self.n_px), dtype=self.dtype, device=dev)
        lib = _L.load()
        _L.check(lib.mk_image_transform(src.data_ptr(), tmp.data_ptr(), d_desc.data_ptr(),
                                        d_coef.data_ptr(), self._lut_dev.data_ptr(), out.data_ptr(),
                                        len(arrs), self.n_px, max_rows, ops._DT[self.dtype],
                                        torch.cuda.current_stream(dev).cuda_stream),
                 "mk_image_transform")
        return out


# ------------------------------------------------------------------- audio --
SAMPLE_RATE, N_FFT, HOP_LENGTH, CHUNK_LENGTH = 16000, 400, 160, 30      # whisper/audio.py
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)),
                    3.0 * f / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


@lru_cache(maxsize=4)
def mel_filters(n_mels: int = 80, sr: int = SAMPLE_RATE, n_fft: int = N_FFT) -> np.ndarray:
    """The Slaney-scale, area-normalised triangular filterbank whisper ships as
    assets/mel_filters.npz (= librosa.filters.mel(sr=16000, n_fft=400, n_mels=n_mels)),
    float32 [n_mels, n_fft // 2 + 1]."""
    fft_freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


_AUDIO_CONST = {}


def _audio_constants(device, n_mels):
    key = (str(device), n_mels)
    c = _AUDIO_CONST.get(key)
    if c is None:
        mf = mel_filters(n_mels)
        nz = mf != 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), mf.shape[1] - nz[:, ::-1].argmax(1), 0).astype(np.int32)
        j = np.arange(N_FFT, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * j / N_FFT), np.sin(2 * np.pi * j / N_FFT)], 1)
        c = _AUDIO_CONST[key] = dict(
            window=torch.hann_window(N_FFT, dtype=torch.float32).to(device),   # periodic, as whisper
            twiddle=torch.from_numpy(tw).to(device),
            mel=torch.from_numpy(mf).to(device), lo=torch.from_numpy(lo).to(device),
            hi=torch.from_numpy(hi).to(device))
    return c


def pad_or_trim(audio: torch.Tensor, length: int = N_SAMPLES) -> torch.Tensor:
    """whisper.pad_or_trim on the last axis (zero pad / cut to 30 s)"""
    n = audio.shape[-1]
    if n > length:
        return audio[..., :length]
    if n < length:
        out = torch.empty(audio.shape[:-1] + (length,), dtype=audio.dtype, device=audio.device)
        flat_in = audio.reshape(-1, n)
        flat_out = out.view(-1, length)
        ops.fill_(flat_out, 0.0)
        ops.copy2d(flat_in, flat_out, flat_in.shape[0], n, flat_in.stride(0), length)
        return out
    return audio


def log_mel_spectrogram(audio: torch.Tensor, n_mels: int = 80, dtype=torch.float32) -> torch.Tensor:
    """whisper.log_mel_spectrogram for float32 PCM on the device: [N] -> [n_mels, N/160] or
    [B, N] -> [B, n_mels, N/160].  (Per-clip dynamic-range clamp, as the reference calls it once
    per clip, llm_trainer.py:336-345.)"""
    if not audio.is_cuda:
        raise ops.MacawHipError("log_mel_spectrogram runs on the HIP device only (no CPU fallback)")
    if audio.dtype != torch.float32:
        raise ops.MacawHipError(f"expected float32 PCM, got {audio.dtype}")
    single = audio.dim() == 1
    a = audio.reshape(-1, audio.shape[-1])
    if a.stride(1) != 1:
        a = a.contiguous()
    B, N = a.shape
    if N % HOP_LENGTH or N < N_FFT:
        raise ValueError(f"clip length {N} must be a multiple of {HOP_LENGTH} (use pad_or_trim)")
    F = N // HOP_LENGTH
    dev = a.device
    c = _audio_constants(dev, n_mels)
    ws = torch.empty((B, n_mels, F), dtype=torch.float32, device=dev)
    wmax = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty((B, n_mels, F), dtype=dtype, device=dev)
    lib = _L.load()
    _L.check(lib.mk_log_mel(a.data_ptr(), a.stride(0), B, N, c["window"].data_ptr(),
                            c["twiddle"].data_ptr(), c["mel"].data_ptr(), c["lo"].data_ptr(),
                            c["hi"].data_ptr(), n_mels, ws.data_ptr(), wmax.data_ptr(), out.data_ptr(),
                            ops._DT[dtype], torch.cuda.current_stream(dev).cuda_stream), "mk_log_mel")
    return out[0] if single else out.view(audio.shape[:-1] + (n_mels, F))


# -------------------------------------------------------- batch assembly --
TAG_IDS = {"image_starts": 32000, "image_ends": 32001, "audio_starts": 32002, "audio_ends": 32003,
           "video_starts": 32004, "video_ends": 32005}   # llm_trainer.py:126-133


class InputBuilder:
    """`get_self_inputs` (llm_trainer.py:306-381) with the arithmetic on the GPU: decoded
    images / frames / PCM in, the model's `inputs` dict out.  Absent modalities (index -1 in the
    reference) are given as None and become zero tensors exactly as the reference feeds them
    (llm_trainer.py:315,332,352), unless drop_absent=True (then the key is None: BASELINE cfg 2/3)."""

    def __init__(self, device="cuda", n_px: int = 224, n_frames: int = 6, dtype=torch.float16,
                 tag_ids=None):
        self.device = torch.device(device)
        self.n_px, self.n_frames, self.dtype = n_px, n_frames, dtype
        # `.half()` of fp32 values, then whatever the model casts to: same rounding chain
        self.transform = ImageTransform(n_px, device, dtype=dtype, via_half=(dtype != torch.float16))
        self.tag_ids = dict(TAG_IDS if tag_ids is None else tag_ids)

    def _images(self, items, per):
        """items: list (len B) of None | image (per == 1) | list of `per` images"""
        B, n = len(items), self.n_px
        out = torch.empty((B * per, 3, n, n), dtype=self.dtype, device=self.device)
        ops.fill_(out.view(B * per, -1), 0.0)
        flat, where = [], []
        for b, it in enumerate(items):
            if it is None:
                continue
            frames = [it] if per == 1 else list(it)
            if len(frames) != per:
                raise ValueError(f"sample {b}: expected {per} frames, got {len(frames)}")
            flat.extend(frames)
            where.extend(range(b * per, (b + 1) * per))
        if flat:
            px = self.transform(flat)
            idx = torch.tensor(where, dtype=torch.int64, device=self.device)
            out.index_copy_(0, idx, px)
        return out.view(B, 3, n, n) if per == 1 else out.view(B, per, 3, n, n)

    def __call__(self, input_ids, attention_mask, labels=None, images=None, videos=None, audios=None,
                 drop_absent: bool = False):
        B = input_ids.shape[0]
        dev = self.device
        inputs = {"input_ids": input_ids.to(dev), "attention_mask": attention_mask.to(dev),
                  "labels": labels.to(dev) if labels is not None else None}
        for key, items, per in (("images", images, 1), ("videos", videos, self.n_frames)):
            if items is None:
                items = [None] * B
            if drop_absent and all(i is None for i in items):
                inputs[key] = None
            else:
                inputs[key] = self._images(items, per)
        if audios is None:
            audios = [None] * B
        if drop_absent and all(a is None for a in audios):
            inputs["audios"] = None
        else:
            mel = torch.empty((B, 80, N_FRAMES), dtype=self.dtype, device=dev)
            ops.fill_(mel.view(B, -1), 0.0)
            present = [b for b, a in enumerate(audios) if a is not None]
            if present:
                pcm = torch.empty((len(present), N_SAMPLES), dtype=torch.float32, device=dev)
                ops.fill_(pcm, 0.0)
                for i, b in enumerate(present):
                    a = torch.as_tensor(audios[b], dtype=torch.float32).reshape(-1)[:N_SAMPLES]
                    pcm[i, :a.numel()].copy_(a, non_blocking=True)       # H2D (or D2D) upload
                # fp32 -> `.half()` -> model dtype, the reference's rounding chain
                m = log_mel_spectrogram(pcm, 80, dtype=torch.float16)
                m = m if self.dtype == torch.float16 else ops.cast(m, self.dtype)
                mel.index_copy_(0, torch.tensor(present, dtype=torch.int64, device=dev), m)
            inputs["audios"] = mel
        for k, v in self.tag_ids.items():
            inputs[k] = torch.full((B,), v, dtype=torch.int32, device=dev)
        return {"inputs": inputs}
