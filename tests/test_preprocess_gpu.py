"""GPU parity of the input-pipeline kernels (csrc/preprocess.hip through the C ABI) against the
oracle (oracle/preprocess_ref.py: real Pillow + restated torchvision / whisper arithmetic).
  images : integer resampling -> BIT-EXACT float32 output (and exactly the cast of it for
           fp16 / bf16 outputs)
  log-mel: the kernel accumulates the DFT in fp64; |ours - float64 oracle| <= 2e-5 and
           |ours - fp32 whisper restatement| <= 1e-4 (the reference's own FFT rounding) on
           values in [-1.5, 2]
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import preprocess_ref as R  # noqa: E402

SIZES = [(300, 400), (224, 224), (231, 500), (57, 41), (640, 427), (225, 1000), (224, 301), (1200, 900),
         (3, 5)]


def test_image_transform_bit_exact_mixed_batch(dev):
    from macaw_llm_amd import preprocess as P
    imgs = [R.synthetic_image(20 + i, H, W) for i, (H, W) in enumerate(SIZES)]
    imgs.append(np.full((100, 100, 3), 255, np.uint8))     # saturated: clip8 upper edge
    imgs.append(np.zeros((333, 224, 3), np.uint8))
    ref = torch.stack([R.pil_transform(a) for a in imgs])
    out = P.ImageTransform(224, dev)(imgs)
    assert out.shape == (len(imgs), 3, 224, 224) and out.dtype == torch.float32
    assert torch.equal(out.cpu(), ref)
    # PIL image objects and a single image are accepted like the reference's preprocess(img)
    from PIL import Image
    one = P.ImageTransform(224, dev)(Image.fromarray(imgs[0]))
    assert torch.equal(one.cpu()[0], ref[0])
    for dt in (torch.float16, torch.bfloat16):
        o = P.ImageTransform(224, dev, dtype=dt)(imgs[:3])
        assert torch.equal(o.cpu(), ref[:3].to(dt))
    # the trainer's .half() followed by a bf16 model cast (llm_trainer.py:366-368)
    o = P.ImageTransform(224, dev, dtype=torch.bfloat16, via_half=True)(imgs[:3])
    assert torch.equal(o.cpu(), ref[:3].half().to(torch.bfloat16))


def test_image_transform_rejects_bad_input(dev):
    from macaw_llm_amd import preprocess as P
    tr = P.ImageTransform(224, dev)
    with pytest.raises(ValueError):
        tr(np.zeros((10, 10), np.uint8))
    with pytest.raises(ValueError):
        tr(np.zeros((10, 10, 3), np.float32))


def test_log_mel_matches_oracle(dev):
    from macaw_llm_amd import preprocess as P
    clips = [R.synthetic_audio(7), 0.01 * R.synthetic_audio(8), R.pad_or_trim(R.synthetic_audio(9, 4.0)),
             np.zeros(R.N_SAMPLES, np.float32)]
    x = torch.from_numpy(np.stack(clips)).to(dev)
    out = P.log_mel_spectrogram(x)
    assert out.shape == (4, 80, 3000) and out.dtype == torch.float32
    o = out.cpu().numpy()
    for i, c in enumerate(clips):                       # per-clip max: clips do not interact
        e64 = np.abs(o[i] - R.log_mel_f64(c)).max()
        e32 = np.abs(o[i] - R.log_mel_whisper_fp32(c).numpy()).max()
        assert e64 <= 2e-5, (i, e64)
        assert e32 <= 1e-4, (i, e32)
    assert np.all(o[3] == -1.5)
    # 1-D input, whisper API shape; fp16 output = the trainer's .half()
    single = P.log_mel_spectrogram(x[0])
    assert single.shape == (80, 3000) and torch.equal(single, out[0])
    h = P.log_mel_spectrogram(x[:2], dtype=torch.float16)
    assert torch.equal(h, out[:2].half())
    # pad_or_trim on the device
    short = torch.from_numpy(R.synthetic_audio(9, 4.0)).to(dev)
    assert torch.equal(P.pad_or_trim(short).cpu(), torch.from_numpy(clips[2]))
    assert P.pad_or_trim(torch.cat([x[0], x[0]])).shape[-1] == R.N_SAMPLES
    with pytest.raises(ValueError):
        P.log_mel_spectrogram(x[0, :1001])


def test_log_mel_128_bins_and_short_clip(dev):
    from macaw_llm_amd import preprocess as P
    c = R.synthetic_audio(11, 2.0)                      # 32000 samples = 200 frames, 7 tiles
    out = P.log_mel_spectrogram(torch.from_numpy(c).to(dev), n_mels=128).cpu().numpy()
    assert out.shape == (128, 200)
    assert np.abs(out - R.log_mel_f64(c, 128)).max() <= 2e-5


def test_input_builder_matches_get_self_inputs(dev):
    """llm_trainer.py:306-381: dict keys, dtypes (.half()), zero tensors for absent modalities"""
    from macaw_llm_amd import preprocess as P
    B = 3
    ids = torch.randint(3, 1000, (B, 16))
    am = torch.ones_like(ids)
    img = [R.synthetic_image(1, 300, 400), None, R.synthetic_image(2, 250, 224)]
    vid = [None, [R.synthetic_image(30 + j, 240, 320) for j in range(6)], None]
    aud = [R.synthetic_audio(3, 5.0), None, R.synthetic_audio(4)]
    d = P.InputBuilder(dev)(ids, am, labels=ids, images=img, videos=vid, audios=aud)["inputs"]
    assert d["images"].shape == (B, 3, 224, 224) and d["images"].dtype == torch.float16
    assert d["videos"].shape == (B, 6, 3, 224, 224) and d["audios"].shape == (B, 80, 3000)
    assert torch.equal(d["images"][0].cpu(), R.pil_transform(img[0]).half())
    assert torch.equal(d["images"][2].cpu(), R.pil_transform(img[2]).half())
    assert not d["images"][1].any() and not d["videos"][0].any() and not d["audios"][1].any()
    for j in range(6):
        assert torch.equal(d["videos"][1, j].cpu(), R.pil_transform(vid[1][j]).half())
    ref_a = R.log_mel_whisper_fp32(R.pad_or_trim(aud[0])).half()
    assert (d["audios"][0].cpu().float() - ref_a.float()).abs().max() <= 2e-3   # 1 fp16 ulp at |x|<=2
    for k, v in P.TAG_IDS.items():
        assert d[k].dtype == torch.int32 and d[k].tolist() == [v] * B
    assert d["input_ids"].is_cuda and d["labels"].is_cuda
    # BASELINE cfg 2/3 style: absent modality -> None
    d2 = P.InputBuilder(dev, dtype=torch.bfloat16)(ids, am, images=img[:1] * B, drop_absent=True)["inputs"]
    assert d2["videos"] is None and d2["audios"] is None and d2["images"].dtype == torch.bfloat16
    assert torch.equal(d2["images"][0].cpu(), R.pil_transform(img[0]).half().to(torch.bfloat16))
