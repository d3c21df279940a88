"""Writes tests/golden/preprocess.npz: outputs of the REAL third-party code the reference
calls for its input pipeline, on deterministic synthetic inputs (oracle/preprocess_ref.py
synthetic_image / synthetic_audio), so the pins survive a Pillow / transformers upgrade:

  crop_<H>x<W>   uint8 [224,224,3]  Pillow 12.2.0 Image.resize(BICUBIC) + torchvision crop rule
  mel_fe         float32 [80, 3000][:, ::8]  transformers WhisperFeatureExtractor (numpy path)
  mel_torch      float32 same stride, whisper.log_mel_spectrogram restated on torch.stft fp32

Run from the repo root:  python -m oracle.make_golden_preprocess
"""
import os

import numpy as np

from oracle import preprocess_ref as R

SIZES = [(300, 400), (224, 224), (231, 500), (57, 41), (640, 427)]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "preprocess.npz")


def main():
    import PIL
    import transformers
    from transformers import WhisperFeatureExtractor
    d = {"pillow_version": np.array(PIL.__version__), "transformers_version": np.array(transformers.__version__)}
    for i, (H, W) in enumerate(SIZES):
        d[f"crop_{H}x{W}"] = R.pil_crop_u8(R.synthetic_image(i, H, W))
    x = R.synthetic_audio(7)
    fe = WhisperFeatureExtractor()
    d["mel_fe"] = fe(x, sampling_rate=16000, return_tensors="np")["input_features"][0][:, ::8].astype(np.float32)
    d["mel_torch"] = R.log_mel_whisper_fp32(x).numpy()[:, ::8]
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
