"""Times the GPU input pipeline (macaw_llm_amd.preprocess) for one BASELINE cfg-3 batch per GPU
(32 images 640x480 + 32 clips of 30 s) beside the reference's CPU path (PIL + torch.stft,
oracle/preprocess_ref.py) on the host cores.  Run on the GPU box: python scripts/bench_preprocess.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import preprocess as P  # noqa: E402
from oracle import preprocess_ref as R  # noqa: E402

B = 32
dev = torch.device("cuda:0")
imgs = [R.synthetic_image(i, 480, 640) for i in range(B)]
pcm = torch.from_numpy(np.stack([R.synthetic_audio(i) for i in range(4)])).repeat(B // 4, 1).to(dev)
tr = P.ImageTransform(224, dev, dtype=torch.bfloat16, via_half=True)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_img = timed(lambda: tr(imgs))
# kernels only (pixels already resident): events around the launches
descs, coef, sb, tb, mr = tr.plan([a.shape[:2] for a in imgs])
t_plan = timed(lambda: tr.plan([a.shape[:2] for a in imgs]), 20)
t_mel = timed(lambda: P.log_mel_spectrogram(pcm, dtype=torch.float16))
t0 = time.perf_counter(); [R.pil_transform(a) for a in imgs]; c_img = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); [R.log_mel_whisper_fp32(pcm[i].cpu().numpy()) for i in range(4)]
c_mel = (time.perf_counter() - t0) * 1e3 * B / 4
print(f"images x{B} (640x480 -> 224): GPU incl. pack+H2D {t_img:.2f} ms (host plan {t_plan:.2f} ms, cached) | "
      f"CPU PIL+ToTensor+Normalize {c_img:.1f} ms")
print(f"log-mel x{B} (30 s): GPU {t_mel:.2f} ms | CPU torch.stft path {c_mel:.1f} ms "
      f"({torch.get_num_threads()} threads)")
